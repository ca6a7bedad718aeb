mkdir -p gpurun_out/r02k && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02k; cd $REPO
timeout 600 python -m pytest tests/test_sarl.py tests/test_noquery.py tests/test_rl_pipeline.py tests/test_mixed.py tests/test_big_crowds.py -m gpu -q -x 2>&1 | tail -n 3
echo "== pipe"; timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om; timeout 120 python scripts/sarl_bench.py --om 1 2>&1 | grep with_om
echo "== no pipe"; CROWDNAV_AMD_SARL_PIPE=0 timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS="--humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500"
run h20_r12 X=1
BARGS="--workload sarl"
run sarl X=1
