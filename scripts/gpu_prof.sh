# rocprofv3 passes over a short bench run; every step bounded by `timeout`, nothing reads stdin.
mkdir -p gpurun_out/prof && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof
ARGS="--steps 800 --warmup 200 --no-cpu-baseline $BENCH_EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 < /dev/null; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 < /dev/null; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 < /dev/null; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1 < /dev/null; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem -o mem -- python $REPO/bench.py $ARGS > $OUT/pmc_mem.log 2>&1 < /dev/null; echo "mem rc=$?"
cd $REPO; find gpurun_out/prof -name "*.csv" | head -30; du -sh gpurun_out/prof
