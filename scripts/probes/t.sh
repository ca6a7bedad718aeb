cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests/test_mixed.py tests/test_sarl.py tests/test_rl_pipeline.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -6
