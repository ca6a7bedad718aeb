# fused kernel hygiene (LDS requests first, no data-dependent branches, state pointers re-read): parity + bench + probes
mkdir -p gpurun_out/r02j && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02j; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 6 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS=""
run h5 X=1
run h5_again X=1
BARGS="--steps 20 --warmup 5"
run drv1 X=1
run drv2 X=1
run drv3 X=1
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run b32k X=1
echo "== probe"; timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 14 | tee $OUT/phase_probe.txt
