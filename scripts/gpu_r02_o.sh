cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; cd $REPO; mkdir -p gpurun_out/r02o
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 14 | tee gpurun_out/r02o/phase_probe_h20.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/phase_probe.py --humans 20 --circle-radius 4 --steps 300 ) 2>&1 | grep -v amdgpu | tail -n 14 | tee gpurun_out/r02o/phase_probe_h20_r4.txt
