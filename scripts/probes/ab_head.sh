#!/bin/bash
# one-output layers on the vector ALUs (in-tree) against the previous library (build/exp/lib_ab_headbase.so): parity tests of the
# networks it touches, then cn_sarl_select at 4096 x 81 x 5
mkdir -p gpurun_out/r06
{
timeout 900 python -m pytest tests/test_sarl.py tests/test_big_crowds.py tests/test_mixed.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in "" build/exp/lib_ab_headbase.so; do
    echo "== ${lib:-in-tree}"
    export CROWDNAV_AMD_LIB=$lib
    for p in ${CN_HEAD_POLICIES:-cadrl}; do timeout 300 python scripts/policy_bench.py --policy $p --humans 5 --iters 30; done
    [ -n "$CN_HEAD_SARL" ] && { timeout 300 python scripts/sarl_bench.py --iters 20 2>&1 | tail -1; timeout 300 python scripts/sarl_bench.py --iters 20 --om 1 2>&1 | tail -1; }
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/ab_head.txt
