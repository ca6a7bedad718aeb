# kernel A/B on the GPU: parity tests with the in-tree library, then bench lines for alternative builds under
# crowdnav_amd/lib/exp/*.so (CROWDNAV_AMD_LIB) and geometry knobs.  Usage: bash scripts/gpu_ab.sh
mkdir -p gpurun_out/ab && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/ab; cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_compat.py -m gpu -q -x > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
run() { # name, env..., -- bench args
  name=$1; shift
  ( export "$@"; timeout 120 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null )
  echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log
}
BARGS="--steps 4000 --warmup 1000"
run h5_default X=1
run h5_base CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_base.so
run h5_coop_e1 CROWDNAV_AMD_ENVS_PER_WAVE=1
run h5_coop_e3 CROWDNAV_AMD_ENVS_PER_WAVE=3 CROWDNAV_AMD_WAVES_PER_BLOCK=1
run h5_coop_e4 CROWDNAV_AMD_ENVS_PER_WAVE=4 CROWDNAV_AMD_WAVES_PER_BLOCK=1
run h5_coop_e4w2 CROWDNAV_AMD_ENVS_PER_WAVE=4 CROWDNAV_AMD_WAVES_PER_BLOCK=2
BARGS="--steps 4000 --warmup 1000 --envs 32768"
run h5_b32k_default X=1
run h5_b32k_base CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_base.so
BARGS="--steps 1000 --warmup 500 --chunk 500 --humans 20 --circle-radius 12"
run h20_base CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_base.so
run h20_coop CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_coop_both.so
run h20_coop_w2 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_coop_both.so CROWDNAV_AMD_WAVES_PER_BLOCK=2
run h20_coop_w4 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_coop_both.so CROWDNAV_AMD_WAVES_PER_BLOCK=4
