# BASELINE configs[4] on the reference's own schedule (crowd_nav/configs/train.config: 3000 IL episodes, 50 IL epochs,
# 10 000 training episodes x (1 sampled episode + 100 SGD batches), target update every 50, evaluation every 1000)
mkdir -p gpurun_out/r02_train && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02_train; cd $REPO
timeout 2300 python examples/train_sarl.py --gpu --seed 0 --output-dir $OUT/model --timing-json $OUT/config5_reference_schedule.json > $OUT/config5_reference_schedule.log 2>&1 < /dev/null; echo "train rc=$?"
tail -n 4 $OUT/config5_reference_schedule.log | cut -c 1-600
rm -f $OUT/model/*.pth
