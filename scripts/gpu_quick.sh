cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_mixed.py tests/test_compat.py -m gpu -q -x 2>&1 | tail -n 3
for a in "" "--envs 32768 --steps 4000" "--humans 20 --circle-radius 12 --steps 1000 --warmup 500 --chunk 500"; do timeout 100 python bench.py --no-cpu-baseline $a > gpurun_out/b.log 2>&1; python scripts/bench_line.py gpurun_out/b.log; done
