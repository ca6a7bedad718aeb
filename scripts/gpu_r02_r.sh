# where does a tile of sarl_reg_kernel go: per-layer clock probe (timing build) + instruction-cache / MFMA / wait counters
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02r; mkdir -p $OUT; cd $REPO
CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/sarl_reg_probe.py 2>&1 | grep -v amdgpu | tee $OUT/reg_probe.txt
cd /tmp
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_icache -o p -- python $REPO/scripts/sarl_bench.py --iters 3 > $OUT/pmc_icache.log 2>&1; echo "pmc1 rc=$?"
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python $REPO/scripts/sarl_bench.py --iters 3 > $OUT/pmc_mfma.log 2>&1; echo "pmc2 rc=$?"
cd $REPO
python scripts/prof_summary.py $OUT/pmc_icache | grep -i "reg_kernel"
python scripts/prof_summary.py $OUT/pmc_mfma | grep -i "reg_kernel"
