# the wave-per-scenario generator keeps the env's numpy stream: explore test at 12 / 14 humans + the suites around it
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02x; mkdir -p $OUT; cd $REPO
timeout 400 python -m pytest tests/test_rl_pipeline.py tests/test_big_crowds.py tests/test_gpu_parity.py tests/test_sarl.py -m gpu -q -x > $OUT/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 12 $OUT/pytest.log | grep -v "version\|Hostname\|Librccl"
