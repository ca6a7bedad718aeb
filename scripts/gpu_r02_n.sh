# (tight timeouts: an asynchronous-fill bug must not eat the GPU budget)
mkdir -p gpurun_out/r02n && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02n; cd $REPO
timeout 120 python -m pytest tests/test_boundary.py -m gpu -q -x > $OUT/pytest_boundary.log 2>&1 < /dev/null; echo "boundary rc=$?"; tail -n 3 $OUT/pytest_boundary.log | grep -v "version\|Hostname\|Librccl"
timeout 240 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 6 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
run() { name=$1; shift; ( export "$@"; timeout 90 python bench.py --no-cpu-baseline $BARGS > $OUT/$name.log 2>&1 < /dev/null ); echo -n "$name: "; timeout 20 python scripts/bench_line.py $OUT/$name.log; grep -o '"paused_env_steps": [0-9]*, "episodes_finished": [0-9]*' $OUT/$name.log; }
S="--seed-base 1000 --seed-mod 1024"
BARGS="--humans 20 --circle-radius 4 --steps 4000 --warmup 200 --chunk 400 --preroll 100 $S --async-fill"
run h20_r4_async_c400 X=1
BARGS="--humans 20 --circle-radius 4 --steps 8000 --warmup 400 --chunk 1000 --preroll 100 $S --async-fill"
run h20_r4_async_c1000 X=1
BARGS="--humans 20 --circle-radius 4 --steps 4000 --warmup 200 --chunk 400 --preroll 100 --async-fill"
run h20_r4_async_c400_train_seeds X=1
