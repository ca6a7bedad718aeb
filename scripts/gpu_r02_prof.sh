# round-2 evidence: GPU tests, smoke, bench lines, rocprofv3 kernel traces + separate PMC passes for both launch shapes
mkdir -p gpurun_out/r02prof && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02prof; cd $REPO
timeout 300 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -n 4 $OUT/pytest_gpu.log | grep -v "version\|Hostname\|Librccl"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log
timeout 200 python bench.py > $OUT/bench_default.log 2>&1 < /dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1 < /dev/null
timeout 300 python bench.py --workload sarl --no-cpu-baseline > $OUT/bench_sarl.log 2>&1 < /dev/null
timeout 300 python bench.py --workload om-sarl --no-cpu-baseline > $OUT/bench_om_sarl.log 2>&1 < /dev/null
timeout 300 python bench.py --no-cpu-baseline --envs 32768 --steps 4000 > $OUT/bench_32k.log 2>&1 < /dev/null
timeout 300 python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 2000 --warmup 500 --chunk 500 > $OUT/bench_h20_r12.log 2>&1 < /dev/null
timeout 120 python bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 400 --warmup 100 --chunk 100 --preroll 100 --seed-base 1000 --seed-mod 1024 > $OUT/bench_h20_r4.log 2>&1 < /dev/null
timeout 120 python bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 8000 --warmup 400 --chunk 1000 --preroll 100 --seed-base 1000 --seed-mod 1024 --async-fill > $OUT/bench_h20_r4_async.log 2>&1 < /dev/null
rocprofv3 -L > $OUT/counters_available.txt 2>&1
cd /tmp
A="--no-cpu-baseline"; D="--no-cpu-baseline --steps 20 --warmup 5"
prof() { # name, rocprof args..., -- bench args
  name=$1; shift; timeout 300 rocprofv3 "$@" > $OUT/$name.log 2>&1 < /dev/null; echo "$name rc=$?"; }
prof trace_default --kernel-trace --stats --output-format csv -d $OUT/trace_default -o trace -- python $REPO/bench.py $A
prof trace_driver --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o trace -- python $REPO/bench.py $D
for shape in default driver; do
  if [ $shape = default ]; then B="$A"; else B="$D"; fi
  prof pmc_${shape}_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${shape}_fetch -o p -- python $REPO/bench.py $B
  prof pmc_${shape}_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${shape}_write -o p -- python $REPO/bench.py $B
  prof pmc_${shape}_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${shape}_sq1 -o p -- python $REPO/bench.py $B
  prof pmc_${shape}_sq2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_${shape}_sq2 -o p -- python $REPO/bench.py $B
done
prof trace_sarl --kernel-trace --stats --output-format csv -d $OUT/trace_sarl -o trace -- python $REPO/scripts/sarl_bench.py
prof pmc_sarl --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sarl -o p -- python $REPO/scripts/sarl_bench.py --iters 3
cd $REPO
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/sarl_reg_probe.py ) 2>&1 | grep -v amdgpu > $OUT/sarl_reg_probe.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 14 > $OUT/phase_probe_h5.txt
( CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_timing.so timeout 100 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 14 > $OUT/phase_probe_h20.txt
timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/launch_probe.txt
rm -f $OUT/r02_traffic.json
python scripts/pmc_to_traffic.py $OUT/r02_traffic.json 4096 5 1000 rollout_fused_kernel tail $OUT/pmc_default_fetch $OUT/pmc_default_write $OUT/pmc_default_sq1 $OUT/pmc_default_sq2 > /dev/null
python scripts/pmc_to_traffic.py $OUT/r02_traffic.json 4096 5 20 rollout_fused_kernel 2 $OUT/pmc_driver_fetch $OUT/pmc_driver_write $OUT/pmc_driver_sq1 $OUT/pmc_driver_sq2 > /dev/null
cat $OUT/r02_traffic.json | head -80
python scripts/prof_summary.py $OUT/trace_default | head -8
python scripts/prof_summary.py $OUT/trace_driver | head -8
python scripts/prof_summary.py $OUT/trace_sarl | head -10
python scripts/prof_summary.py $OUT/pmc_sarl | tail -n 8
for f in $OUT/bench*.log; do echo $f; timeout 20 python scripts/bench_line.py $f; done
