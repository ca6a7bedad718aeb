cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_rl_pipeline.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per step"
