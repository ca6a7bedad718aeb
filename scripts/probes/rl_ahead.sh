cd $GRAFT_REPO_ROOT
for a in 2 4 8 16; do echo -n "ahead=$a: "; CROWDNAV_AMD_RL_AHEAD=$a python scripts/probes/rl_parts.py 2>&1 | tail -1; done
