cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_ring_wrap.py tests/test_gpu_parity.py tests/test_boundary.py tests/test_bench_size_parity.py tests/test_edge_sizes.py tests/test_compat.py -m gpu -q -x 2>&1 | tail -4
for cf in 1 0; do
  for i in 1 2; do CROWDNAV_AMD_COMPACT_FILL=$cf python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-r3-definition > gpurun_out/r05/cf${cf}_drv_$i.log 2>&1; echo -n "compact_fill=$cf driver: "; python - gpurun_out/r05/cf${cf}_drv_$i.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print({k:(round(d[k]/1e6,1) if 'value' in k and d[k] else d[k]) for k in ('value','value_incl_boundary','value_amortised_fill','boundary_ms','fill_ms')})
PY
  done
  CROWDNAV_AMD_COMPACT_FILL=$cf python bench.py --no-secondary --no-cpu-baseline --no-r3-definition > gpurun_out/r05/cf${cf}_default.log 2>&1; echo -n "compact_fill=$cf default: "; python scripts/bench_line.py gpurun_out/r05/cf${cf}_default.log
done
( CROWDNAV_AMD_LIB=$GRAFT_REPO_ROOT/build/exp/lib_timing.so timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 14 | tee gpurun_out/r05/phase_probe_h20.txt
