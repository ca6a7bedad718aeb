mkdir -p gpurun_out/prof_sarl && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/prof_sarl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/scripts/sarl_bench.py > $OUT/trace.log 2>&1 < /dev/null; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python $REPO/scripts/sarl_bench.py --iters 3 > $OUT/pmc_sq.log 2>&1 < /dev/null; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem -o mem -- python $REPO/scripts/sarl_bench.py --iters 3 > $OUT/pmc_mem.log 2>&1 < /dev/null; echo "mem rc=$?"
cd $REPO; tail -n 2 $OUT/trace.log
