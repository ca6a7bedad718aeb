cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/crowdnav_amd/lib/exp/lib_timing.so
for e in 1 2 3; do echo "== H20 E=$e"; CROWDNAV_AMD_LIB=$L CROWDNAV_AMD_ENVS_PER_WAVE=$e timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 2>&1 | grep -v amdgpu.ids | tail -n 10; done
