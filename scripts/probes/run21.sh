cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rl_pipeline.py tests/test_sarl.py -m gpu -q -x -k "narrow or sample_step or reproduces" 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -4
timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per call\|per step"
timeout 300 python scripts/probes/rl_parts.py 2>&1 | grep "ms per call"
