# gpurun_out/<tag>/ -> profiles/<tag>_* : the committed record of a round's GPU evidence (bench lines, kernel stats, traces,
# the GPU suite's log, the traffic / counter summaries).      bash scripts/stage_profiles.sh r06
TAG=${1:-r06}; cd "$(dirname "$0")/.."; S=gpurun_out/$TAG; P=profiles
python scripts/bench_to_profiles.py $TAG
for t in default driver sarl om_sarl h20 cadrl lstm_rl lstm_rl2 step; do
  [ -f $S/trace_$t/trace_kernel_stats.csv ] && cp $S/trace_$t/trace_kernel_stats.csv $P/${TAG}_kernel_stats_$t.csv
done
[ -f $S/trace_driver/trace_kernel_trace.csv ] && cp $S/trace_driver/trace_kernel_trace.csv $P/${TAG}_kernel_trace_driver.csv
[ -f $S/pytest_gpu.log ] && grep -vE "amdgpu.ids|Hostname|Librccl|version" $S/pytest_gpu.log > $P/${TAG}_pytest_gpu.log
[ -f $S/smoke.log ] && tail -n 2 $S/smoke.log >> $P/${TAG}_pytest_gpu.log
[ -f $S/${TAG}_traffic.json ] && cp $S/${TAG}_traffic.json $P/${TAG}_traffic.json
for f in pmc_networks_summary valu_rate scratch_launch boundary_probe2 phase_probe_h5 phase_probe_h20 launch_probe reset_probe trig_error; do
  [ -f $S/$f.txt ] && cp $S/$f.txt $P/${TAG}_$f.txt
done
for v in sarl om_sarl; do [ -d $S/pmc_${v}_mfma ] && python scripts/prof_summary.py $S/pmc_${v}_mfma $S/pmc_${v}_fetch $S/pmc_${v}_write > $P/${TAG}_pmc_${v}_summary.txt 2>/dev/null; done
ls $P | grep "^${TAG}_" | tr '\n' ' '
