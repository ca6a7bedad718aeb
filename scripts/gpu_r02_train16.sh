# regression check of the training pipeline on the round's final kernels: the 16-episodes-per-iteration schedule (~90 s)
mkdir -p gpurun_out/r02_train16 && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02_train16; cd $REPO
timeout 400 python examples/train_sarl.py --gpu --seed 0 --train-episodes 625 --sample-episodes 16 --epsilon-decay 250 --target-update-interval 3 --evaluation-interval 125 --output-dir $OUT/model --timing-json $OUT/config5_train16.json > $OUT/config5_train16.log 2>&1 < /dev/null; echo "train rc=$?"
tail -n 3 $OUT/config5_train16.log | cut -c 1-400
rm -f $OUT/model/*.pth
