#!/bin/bash
# the value-network translation unit built with other flags (build/exp/lib_ab_sarl*.so) against the in-tree library:
# cn_sarl_select of the five networks at 4096 x 81 x 5, the streamed sampled step
mkdir -p gpurun_out/r06
{
for rep in 1 2; do
  for lib in "" $(ls build/exp/lib_ab_sarl*.so 2>/dev/null); do
    echo "== ${lib:-in-tree}"
    export CROWDNAV_AMD_LIB=$lib
    timeout 300 python scripts/sarl_bench.py --iters 20 2>&1 | tail -1
    for p in cadrl lstm_rl lstm_rl2; do timeout 300 python scripts/policy_bench.py --policy $p --humans 5 --iters 30; done
    timeout 300 python scripts/probes/sample_step_rows.py 2>&1 | tail -4
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/ab_sarl_tu.txt
