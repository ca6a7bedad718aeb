# RL-phase pipeline on the GPU: new parity tests, BASELINE configs[4] schedule with per-phase wall-clock, OM-SARL bench
mkdir -p gpurun_out/rl && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/rl; cd $REPO
timeout 500 python -m pytest tests/test_rl_pipeline.py tests/test_sarl.py -m gpu -q > $OUT/pytest_rl.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_rl.log
timeout 420 python examples/train_sarl.py --gpu --il-episodes 3000 --il-epochs 50 --train-episodes 20 --sample-episodes 4096 \
  --train-batches 100 --evaluation-interval 10 --target-update-interval 5 --timing-json $OUT/config5.json > $OUT/config5.log 2>&1 < /dev/null; echo "train rc=$?" >> $OUT/config5.log
timeout 200 python bench.py --workload om-sarl --no-cpu-baseline > $OUT/bench_om_sarl.log 2>&1 < /dev/null
tail -n 25 $OUT/pytest_rl.log; tail -n 12 $OUT/config5.log; timeout 20 python scripts/bench_line.py $OUT/bench_om_sarl.log
