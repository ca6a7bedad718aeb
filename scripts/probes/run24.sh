cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_rl_pipeline.py tests/test_sarl.py tests/test_compat.py tests/test_dropin_surface.py tests/test_noquery.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -12
timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per call\|per step"
CROWDNAV_AMD_SARL_FUSED_STEP=0 timeout 120 python scripts/probes/narrow_probe.py 1 2>&1 | grep -v amdgpu.ids | grep "per step"
timeout 300 python scripts/probes/rl_parts.py 2>&1 | grep "ms per call\|capture"
