# one-workgroup summary kernel for small block sets: boundary tests, driver-shape line
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02z; mkdir -p $OUT; cd $REPO
timeout 300 python -m pytest tests/test_boundary.py tests/test_distributed.py -m gpu -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log | grep -v "version\|Hostname\|Librccl"
for i in 1 2 3; do timeout 100 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 > $OUT/drv$i.log 2>&1; echo -n "driver: "; python scripts/bench_line.py $OUT/drv$i.log; done
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o trace -- python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/trace.log 2>&1; cd $REPO
python scripts/prof_summary.py $OUT/trace_driver | grep -i "summary\|pack\|finish"
