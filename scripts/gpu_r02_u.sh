# per-env transition counts + rollout_finish_kernel instead of one atomic per env: full GPU suite, launch probe, bench lines
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02u; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe.txt
for i in 1 2 3; do timeout 100 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/drv$i.log 2>&1; echo -n "drv$i: "; python scripts/bench_line.py $OUT/drv$i.log; done
timeout 100 python bench.py --no-cpu-baseline > $OUT/h5.log 2>&1; echo -n "h5: "; python scripts/bench_line.py $OUT/h5.log
timeout 100 python bench.py --no-cpu-baseline --workload sarl > $OUT/sarl.log 2>&1; echo -n "sarl: "; python scripts/bench_line.py $OUT/sarl.log
