# candidate-form LP: parity, then A/B against the previous solve (lib_oldlp.so), launch probe, phase probe
mkdir -p gpurun_out/r02c && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02c; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 8 $OUT/pytest_gpu.log
run() { name=$1; shift; ( export "$@"; timeout 200 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
OLD=$REPO/crowdnav_amd/lib/exp/lib_oldlp.so
BARGS="--steps 8000 --warmup 1000"
run h5_new X=1
run h5_old CROWDNAV_AMD_LIB=$OLD
run h5_new_e1 CROWDNAV_AMD_ENVS_PER_WAVE=1
run h5_new_e3 CROWDNAV_AMD_ENVS_PER_WAVE=3
run h5_new_e4 CROWDNAV_AMD_ENVS_PER_WAVE=4
BARGS="--steps 20 --warmup 5"
run drv_new X=1
run drv_new2 X=1
run drv_old CROWDNAV_AMD_LIB=$OLD
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run b32k_new X=1
run b32k_old CROWDNAV_AMD_LIB=$OLD
run b32k_new_e5 CROWDNAV_AMD_ENVS_PER_WAVE=5
echo "== probe new"; timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_new.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 14 | tee $OUT/phase_probe.txt
