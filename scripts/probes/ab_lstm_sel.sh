#!/bin/bash
# LSTM kernels with the branch-free present-select (in-tree) against the previous library (build/exp/lib_ab_lstmbase.so)
mkdir -p gpurun_out/r06
{
timeout 900 python -m pytest tests/test_sarl.py tests/test_big_crowds.py tests/test_mixed.py -q -m gpu -k "lstm" 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in "" build/exp/lib_ab_lstmbase.so; do
    echo "== ${lib:-in-tree}"
    for p in lstm_rl lstm_rl2; do CROWDNAV_AMD_LIB=$lib timeout 300 python scripts/policy_bench.py --policy $p --humans 5 --iters 30; done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/ab_lstm_sel.txt
