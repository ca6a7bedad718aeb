mkdir -p gpurun_out && nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/host.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -5 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log
