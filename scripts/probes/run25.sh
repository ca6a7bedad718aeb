# final bench lines + rocprofv3 kernel stats of the single-env sampling loop (narrow_probe.py: 220 cn_sarl_select, 300 cn_sarl_sample_step)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
CN_TAIL=4 bash scripts/gpu.sh bench
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np -o np -- python $GRAFT_REPO_ROOT/scripts/probes/narrow_probe.py 1 > /tmp/np.log 2>&1
grep "per call\|per step" /tmp/np.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, shutil
for f in glob.glob('/tmp/prof_np/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, 'gpurun_out/r05/kernel_stats_sample_step.csv')
    for row in csv.DictReader(open(f)):
        if 'cn::' in row['Name']:
            print(row['Name'].split('(')[0][:70], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
PY
