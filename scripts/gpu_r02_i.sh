mkdir -p gpurun_out/r02i && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02i; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 12 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
echo "== new"; timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om; timeout 120 python scripts/sarl_bench.py --om 1 2>&1 | grep with_om
echo "== probe new"; CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/sarl_phase_probe.py 2>&1 | grep -v amdgpu | tee $OUT/sarl_probe_new.txt
