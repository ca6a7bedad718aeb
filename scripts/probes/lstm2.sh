#!/bin/bash
# lstm2_reg_kernel (lstm_rl.ValueNetwork2 in registers): parity tests, then cn_sarl_select at 4096 x 81 x 5 against the LDS kernel
mkdir -p gpurun_out/r06
{
timeout 900 python -m pytest tests/test_sarl.py tests/test_big_crowds.py -q -m gpu -k "pairwise" -x 2>&1 | tail -5
for reg in 1 0; do
  echo "== CROWDNAV_AMD_SARL_REG=$reg"
  CROWDNAV_AMD_SARL_REG=$reg timeout 300 python scripts/policy_bench.py --policy lstm_rl2 --humans 5 --iters 20
done
for lib in $(ls build/exp/lib_ab_lstm2*.so 2>/dev/null); do
  echo "== $lib"
  CROWDNAV_AMD_LIB=$lib timeout 300 python scripts/policy_bench.py --policy lstm_rl2 --humans 5 --iters 20
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/lstm2.txt
