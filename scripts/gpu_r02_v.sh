# 20-human kernel: two-sweep pair phase (float4 candidate rows, half-planes on kept slots only): full GPU suite, bench, phase probe
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02v; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500 > $OUT/h20_$i.log 2>&1; echo -n "h20 r12: "; python scripts/bench_line.py $OUT/h20_$i.log; done
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 14 | tee $OUT/phase_probe_h20.txt
