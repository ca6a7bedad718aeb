mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 < /dev/null | tail -n 3
for cfg in "2 1" "3 2" "4 2" "5 3" "6 3" "8 4" "10 5" "10 3" "10 8"; do set -- $cfg; echo "E $1 W $2"; CROWDNAV_AMD_ENVS_PER_WAVE=$1 CROWDNAV_AMD_WAVES_PER_BLOCK=$2 timeout 300 python bench.py --no-cpu-baseline --steps 4000 --warmup 1000 --chunk 1000 2>&1 < /dev/null | timeout 20 python scripts/bench_line.py /dev/stdin; done
