mkdir -p gpurun_out/r02f && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02f; cd $REPO
timeout 600 python -m pytest tests/test_sarl.py tests/test_noquery.py -m gpu -q -x 2>&1 | tail -n 3
echo "== new"; timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om; timeout 120 python scripts/sarl_bench.py --om 1 2>&1 | grep with_om
echo "== v1";  CROWDNAV_AMD_SARL_V1=1 timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om
echo "== probe new"; CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/sarl_phase_probe.py 2>&1 | grep -v amdgpu | tee $OUT/sarl_probe_new.txt
echo "== probe v1"; CROWDNAV_AMD_SARL_V1=1 CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/sarl_phase_probe.py 2>&1 | grep -v amdgpu | tee $OUT/sarl_probe_v1.txt
