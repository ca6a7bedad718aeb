# the narrow-tile value network + cn_sarl_sample_step: parity, then what a sampled step costs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_sarl.py tests/test_rl_pipeline.py tests/test_noquery.py tests/test_mixed.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -25
timeout 300 python scripts/probes/rl_parts.py 2>&1 | grep -v amdgpu.ids | tail -15
