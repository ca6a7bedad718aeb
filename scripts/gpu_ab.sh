# kernel A/B on the GPU: parity tests with the in-tree library, then bench lines for alternative builds under
# crowdnav_amd/lib/exp/*.so (CROWDNAV_AMD_LIB) and geometry knobs.  Usage: bash scripts/gpu_ab.sh
mkdir -p gpurun_out/ab && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/ab; cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_compat.py tests/test_mixed.py -m gpu -q -x > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
run() { # name, env..., -- bench args
  name=$1; shift
  ( export "$@"; timeout 120 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null )
  echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log
}
ALT=$REPO/crowdnav_amd/lib/exp/lib_lp3_serial.so
BARGS="--steps 4000 --warmup 1000"
run h5_default X=1
run h5_alt CROWDNAV_AMD_LIB=$ALT
run h5_default_e3 CROWDNAV_AMD_ENVS_PER_WAVE=3
run h5_default_e4 CROWDNAV_AMD_ENVS_PER_WAVE=4
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run h5_b32k_default X=1
run h5_b32k_alt CROWDNAV_AMD_LIB=$ALT
BARGS="--steps 1000 --warmup 500 --chunk 500 --humans 20 --circle-radius 12"
run h20_default X=1
run h20_alt CROWDNAV_AMD_LIB=$ALT
run h20_default_e2 CROWDNAV_AMD_ENVS_PER_WAVE=2
run h20_default_e3 CROWDNAV_AMD_ENVS_PER_WAVE=3
run h20_default_w2 CROWDNAV_AMD_WAVES_PER_BLOCK=2
run h20_default_e3w4 CROWDNAV_AMD_ENVS_PER_WAVE=3 CROWDNAV_AMD_WAVES_PER_BLOCK=4
L=$REPO/crowdnav_amd/lib/exp/lib_timing.so
( CROWDNAV_AMD_LIB=$L timeout 120 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 11
( CROWDNAV_AMD_LIB=$L timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 11
