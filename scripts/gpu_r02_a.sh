# round 2, call A: GPU tests + the bench line under the driver's flags and the default flags + kernel trace of both
mkdir -p gpurun_out/r02a && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02a; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 15 $OUT/pytest_gpu.log
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_$i.log 2>&1 < /dev/null; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_full.log 2>&1 < /dev/null
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.log 2>&1 < /dev/null
timeout 300 python bench.py --no-cpu-baseline --chunk 20 --steps 2000 --warmup 200 > $OUT/bench_chunk20.log 2>&1 < /dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_100.log 2>&1 < /dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o trace -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_driver.log 2>&1 < /dev/null; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o trace -- python $REPO/bench.py --no-cpu-baseline > $OUT/trace_default.log 2>&1 < /dev/null; echo "trace rc=$?"
cd $REPO
for f in $OUT/bench*.log; do echo $f; tail -n 1 $f | cut -c 1-1500; done
python scripts/prof_summary.py $OUT/trace_driver | head -30
python scripts/prof_summary.py $OUT/trace_default | head -20
