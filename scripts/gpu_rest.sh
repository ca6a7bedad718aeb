cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/final; mkdir -p $OUT; cd $REPO
timeout 420 python examples/train_sarl.py --gpu --il-episodes 3000 --il-epochs 50 --train-episodes 625 --sample-episodes 16 --train-batches 100 --epsilon-decay 250 --target-update-interval 3 --evaluation-interval 125 --timing-json $OUT/config5_paced.json > $OUT/config5_paced.log 2>&1 < /dev/null; echo "train rc=$?"
timeout 200 python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500 > $OUT/bench_h20.log 2>&1 < /dev/null
timeout 300 python bench.py > $OUT/bench2.log 2>&1 < /dev/null
for f in $OUT/bench_h20.log $OUT/bench2.log; do timeout 20 python scripts/bench_line.py $f; done; grep -v "TRAIN in" $OUT/config5_paced.log | tail -n 4 | cut -c1-300
