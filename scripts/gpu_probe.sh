mkdir -p gpurun_out/probe && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/probe; cd $REPO
L=$REPO/crowdnav_amd/lib/exp
( CROWDNAV_AMD_LIB=$L/lib_timing.so timeout 120 python scripts/phase_probe.py ) > $OUT/h5_e2.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing.so CROWDNAV_AMD_ENVS_PER_WAVE=1 timeout 120 python scripts/phase_probe.py ) > $OUT/h5_e1.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing.so CROWDNAV_AMD_ENVS_PER_WAVE=4 timeout 120 python scripts/phase_probe.py ) > $OUT/h5_e4.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing.so timeout 120 python scripts/phase_probe.py --envs 32768 ) > $OUT/h5_b32k.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing_coop.so timeout 120 python scripts/phase_probe.py ) > $OUT/h5_e2_coop.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing.so timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) > $OUT/h20.log 2>&1 < /dev/null
( CROWDNAV_AMD_LIB=$L/lib_timing_coop.so timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) > $OUT/h20_coop.log 2>&1 < /dev/null
for f in $OUT/*.log; do echo "== $f"; grep -v amdgpu.ids $f | tail -n 11; done
