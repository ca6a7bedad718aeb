cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_sarl.py tests/test_rl_pipeline.py tests/test_noquery.py tests/test_mixed.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -12
timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per step" | tail -1
