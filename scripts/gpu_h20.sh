# H = 20 A/B: occupancy-constrained builds of the 10-half-plane rollout kernel (lib/exp/lib_h20_w{3,4}.so)
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; cd $REPO; mkdir -p gpurun_out/ab
run() { name=$1; shift; ( export "$@"; timeout 120 python bench.py --no-cpu-baseline --steps 1000 --warmup 500 --chunk 500 --humans 20 --circle-radius 12 > gpurun_out/ab/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py gpurun_out/ab/$name.log; }
run h20_default X=1
run h20_waves3 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_h20_w3.so
run h20_waves4 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_h20_w4.so
run h20_waves3_e2 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_h20_w3.so CROWDNAV_AMD_ENVS_PER_WAVE=2
run h20_waves4_e2 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_h20_w4.so CROWDNAV_AMD_ENVS_PER_WAVE=2
