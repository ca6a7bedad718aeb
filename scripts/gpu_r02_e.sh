# SARL value network: persistent kernel + cross-barrier B prefetch vs the round-1 kernel
mkdir -p gpurun_out/r02e && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02e; cd $REPO
timeout 600 python -m pytest tests/test_sarl.py tests/test_noquery.py tests/test_rl_pipeline.py -m gpu -q -x 2>&1 | tail -n 4
echo "== new"; timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om; timeout 120 python scripts/sarl_bench.py --om 1 2>&1 | grep with_om
echo "== v1";  CROWDNAV_AMD_SARL_V1=1 timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om; CROWDNAV_AMD_SARL_V1=1 timeout 120 python scripts/sarl_bench.py --om 1 2>&1 | grep with_om
timeout 200 python bench.py --workload sarl --no-cpu-baseline 2>&1 | tail -n 1 | cut -c 1-400
# H = 20 after the LDS fix, 32768 / 16384 envs with the new default E
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS="--humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500"
run h20_r12 X=1
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run b32k X=1
BARGS="--steps 4000 --warmup 1000 --envs 16384"
run b16k X=1
BARGS="--steps 4000 --warmup 1000 --envs 65536"
run b64k X=1
