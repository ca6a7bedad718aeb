cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/build/exp
for l in lib_timing lib_timing_warm; do echo "== $l"; CROWDNAV_AMD_LIB=$E/$l.so timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | tail -20 | head -17; done
for l in "" $E/lib_ab_warm.so; do echo "== ${l:-intree}"; CROWDNAV_AMD_LIB=$l timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per call\|per step" | tail -3; done
CROWDNAV_AMD_LIB=$E/lib_ab_warm.so timeout 600 python -m pytest tests/test_rl_pipeline.py tests/test_sarl.py -m gpu -q -x -k "narrow or sample_step or single_episode" 2>&1 | grep -E "passed|failed"
