# round-end evidence: GPU tests, smoke, bench (headline + SARL), rocprofv3 kernel trace + PMC passes -> gpurun_out/
mkdir -p gpurun_out/final && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/final; cd $REPO
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.log 2>&1 < /dev/null; echo "bench rc=$?" >> $OUT/bench.log
timeout 300 python bench.py --workload sarl > $OUT/bench_sarl.log 2>&1 < /dev/null
timeout 300 python bench.py --workload om-sarl > $OUT/bench_om_sarl.log 2>&1 < /dev/null
cd /tmp
ARGS="--no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 < /dev/null; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 < /dev/null; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 < /dev/null; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1 < /dev/null; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sarl -o trace -- python $REPO/scripts/sarl_bench.py > $OUT/trace_sarl.log 2>&1 < /dev/null; echo "trace_sarl rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sarl -o sarl -- python $REPO/scripts/sarl_bench.py --iters 3 > $OUT/pmc_sarl.log 2>&1 < /dev/null; echo "pmc_sarl rc=$?"
cd $REPO
timeout 420 python examples/train_sarl.py --gpu --il-episodes 3000 --il-epochs 50 --train-episodes 625 --sample-episodes 16 --train-batches 100 --epsilon-decay 250 --target-update-interval 3 --evaluation-interval 125 --timing-json $OUT/config5_paced.json > $OUT/config5_paced.log 2>&1 < /dev/null; echo "train rc=$?"
timeout 200 python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500 > $OUT/bench_h20.log 2>&1 < /dev/null
cd $REPO; tail -n 3 $OUT/pytest_gpu.log $OUT/smoke.log; for f in $OUT/bench*.log; do echo $f; timeout 20 python scripts/bench_line.py $f; done
