mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
for E in 1 2 3 4 5; do CROWDNAV_AMD_ENVS_PER_WAVE=$E timeout 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 400 > gpurun_out/bench_E$E.log 2>&1 < /dev/null; done
tail -n 12 gpurun_out/pytest_gpu.log; for f in gpurun_out/bench_E*.log; do echo $f; timeout 20 python scripts/bench_line.py "$f"; done
