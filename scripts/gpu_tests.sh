mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1 < /dev/null
tail -n 15 gpurun_out/pytest_gpu.log; tail -n 2 gpurun_out/bench_quick.log | cut -c1-300
