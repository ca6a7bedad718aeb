# swept distances inside the fallback's first exchange: parity of the fused kernel, then A/B against the previous build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_bench_size_parity.py tests/test_gpu_parity.py tests/test_boundary.py tests/test_ring_wrap.py tests/test_dropin_surface.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
CROWDNAV_AMD_LIB=$GRAFT_REPO_ROOT/build/exp/lib_ab_prev.so timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
bash scripts/gpu.sh ab
