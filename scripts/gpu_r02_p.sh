# register-resident value network (sarl_reg_kernel): parity tests, then timing against the LDS kernel
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02p; mkdir -p $OUT; cd $REPO
timeout 240 python -m pytest tests/test_sarl.py tests/test_mixed.py tests/test_noquery.py tests/test_rl_pipeline.py -m gpu -x -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 5 $OUT/pytest.log
for reg in 1 0; do
  echo -n "reg $reg: "; CROWDNAV_AMD_SARL_REG=$reg timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om
done
CROWDNAV_AMD_SARL_REG=1 timeout 120 python bench.py --workload sarl --no-cpu-baseline > $OUT/bench_sarl.log 2>&1; tail -n 1 $OUT/bench_sarl.log | cut -c 1-400
