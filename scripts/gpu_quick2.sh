cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; ( export "$@"; timeout 120 python bench.py --no-cpu-baseline --steps 1000 --warmup 500 --chunk 500 --humans 20 --circle-radius 12 > gpurun_out/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py gpurun_out/$name.log; }
run h20_e1 X=1
run h20_e2 CROWDNAV_AMD_ENVS_PER_WAVE=2
run h20_e3 CROWDNAV_AMD_ENVS_PER_WAVE=3
L=$GRAFT_REPO_ROOT/crowdnav_amd/lib/exp/lib_timing.so
CROWDNAV_AMD_LIB=$L timeout 100 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 2>&1 | tail -n 12
