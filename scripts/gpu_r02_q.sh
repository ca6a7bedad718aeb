# quick A/B of the value-network kernel: SARL parity tests, then cn_sarl_select timing (plain and with occupancy maps)
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02q; mkdir -p $OUT; cd $REPO
timeout 200 python -m pytest tests/test_sarl.py tests/test_mixed.py -m gpu -x -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
for om in 0 1; do timeout 100 python scripts/sarl_bench.py --om $om 2>&1 | grep with_om; done
