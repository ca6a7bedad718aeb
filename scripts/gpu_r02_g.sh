# four-barrier fused rollout: parity (whole GPU suite), A/B against the general kernel (CROWDNAV_AMD_FUSED=0), launch + phase probes
mkdir -p gpurun_out/r02g && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02g; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 8 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS=""
run h5_fused X=1
run h5_general CROWDNAV_AMD_FUSED=0
BARGS="--steps 20 --warmup 5"
run drv_fused1 X=1
run drv_fused2 X=1
run drv_general CROWDNAV_AMD_FUSED=0
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run b32k_fused X=1
run b32k_general CROWDNAV_AMD_FUSED=0
echo "== probe fused"; timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_fused.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 14 | tee $OUT/phase_probe.txt
