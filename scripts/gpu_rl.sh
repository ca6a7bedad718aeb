# RL-phase pipeline on the GPU: parity tests, then BASELINE configs[4] (train.py schedule) with per-phase wall-clock:
#  (a) reference-like pacing (16 episodes per RL iteration, 625 iterations = the reference's 10 000 RL episodes)
#  (b) throughput pacing (4096 episodes per RL iteration)
mkdir -p gpurun_out/rl && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/rl; cd $REPO
timeout 500 python -m pytest tests/test_rl_pipeline.py tests/test_compat.py -m gpu -q > $OUT/pytest_rl.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_rl.log
timeout 420 python examples/train_sarl.py --gpu --il-episodes 3000 --il-epochs 50 --train-episodes 625 --sample-episodes 16 \
  --train-batches 100 --epsilon-decay 250 --target-update-interval 3 --evaluation-interval 125 \
  --timing-json $OUT/config5_paced.json > $OUT/config5_paced.log 2>&1 < /dev/null; echo "train rc=$?" >> $OUT/config5_paced.log
tail -n 8 $OUT/pytest_rl.log; grep -v "TRAIN in" $OUT/config5_paced.log | cut -c1-400 | tail -n 30
