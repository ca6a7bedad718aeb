# sarl_reg_kernel iteration: parity tests, timing, per-layer clock probe
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02s; mkdir -p $OUT; cd $REPO
timeout 200 python -m pytest tests/test_sarl.py tests/test_mixed.py -m gpu -x -q > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
for om in 0 1; do timeout 100 python scripts/sarl_bench.py --om $om 2>&1 | grep with_om; done
CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/sarl_reg_probe.py 2>&1 | grep -v amdgpu | tee $OUT/reg_probe.txt
