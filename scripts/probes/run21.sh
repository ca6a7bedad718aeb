cd $GRAFT_REPO_ROOT
for b in 4 8 16; do for n in 2 0; do echo "envs $b narrow $n"; CROWDNAV_AMD_SARL_NARROW=$n timeout 120 python scripts/probes/narrow_probe.py $b 2>&1 | grep -v amdgpu.ids | grep "per call\|per step" | tail -2; done; done
