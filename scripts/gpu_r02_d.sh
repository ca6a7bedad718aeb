# query_env=false + reference-geometry tests, driver-shape bench after the host-side changes, E sweep at 32768 envs,
# configs[3] (H = 20) at the reference geometry (circle radius 4) and at radius 12 with a kernel trace (resets vs steps)
mkdir -p gpurun_out/r02d && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02d; cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 8 $OUT/pytest_gpu.log
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS="--steps 20 --warmup 5"
run drv1 X=1
run drv2 X=1
run drv3 X=1
BARGS=""
run h5_default X=1
BARGS="--steps 4000 --warmup 1000 --envs 32768"
for e in 3 4 5 6 8; do run b32k_e$e CROWDNAV_AMD_ENVS_PER_WAVE=$e; done
BARGS="--steps 2000 --warmup 500 --envs 16384"
for e in 2 3 4 5; do run b16k_e$e CROWDNAV_AMD_ENVS_PER_WAVE=$e; done
BARGS="--steps 4000 --warmup 1000 --envs 8192"
for e in 2 3 4; do run b8k_e$e CROWDNAV_AMD_ENVS_PER_WAVE=$e; done
BARGS="--humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500"
run h20_r12 X=1
BARGS="--humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100"
run h20_r4 X=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_h20_r4 -o trace -- python $REPO/bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100 > $OUT/trace_h20_r4.log 2>&1 < /dev/null; echo "trace rc=$?"
cd $REPO
python scripts/prof_summary.py $OUT/trace_h20_r4 | head -12
