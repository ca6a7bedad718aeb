# the final tree: whole GPU suite; PMC pass (counters only) over the single-env sampling loop
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
bash scripts/gpu.sh tests
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_np -o p -- python $GRAFT_REPO_ROOT/scripts/probes/narrow_probe.py 1 > /tmp/pmc_np.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/prof_summary.py /tmp/pmc_np | grep -E "narrow|decide_step|orca_kernel" | tee gpurun_out/r05/pmc_sample_step_summary.txt
tail -2 /tmp/pmc_np.log
