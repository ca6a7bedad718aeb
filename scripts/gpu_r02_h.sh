# scenario generators with the sqrt-free rejection test: parity, then configs[3] (H = 20) at the reference geometry
mkdir -p gpurun_out/r02h && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02h; cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mixed.py tests/test_boundary.py tests/test_compat.py -m gpu -q -x 2>&1 | tail -n 3
run() { name=$1; shift; ( export "$@"; timeout 300 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; }
BARGS=""
run h5_default X=1
BARGS="--humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100"
run h20_r4_train_seeds X=1
BARGS="--humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100 --seed-base 1000 --seed-mod 1024"
run h20_r4_first1024 X=1
BARGS="--humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 40 --preroll 100 --seed-base 1000 --seed-mod 1024"
run h20_r4_first1024_c40 X=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_h20_r4 -o trace -- python $REPO/bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100 --seed-base 1000 --seed-mod 1024 > $OUT/trace_h20_r4.log 2>&1 < /dev/null; echo "trace rc=$?"
cd $REPO
python scripts/prof_summary.py $OUT/trace_h20_r4 | head -8
timeout 100 python scripts/reset_probe.py 2>&1 | tail -n 5
