# final sanity of the round: whole GPU suite, smoke, the driver-shape line and the SARL line
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02y; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv.log 2>&1; echo -n "driver: "; python scripts/bench_line.py $OUT/drv.log
timeout 100 python bench.py --no-cpu-baseline --workload sarl > $OUT/sarl.log 2>&1; echo -n "sarl: "; python scripts/bench_line.py $OUT/sarl.log
for om in 0 1; do timeout 100 python scripts/sarl_bench.py --om $om 2>&1 | grep with_om; done
