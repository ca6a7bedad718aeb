cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_rl_pipeline.py tests/test_sarl.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rl -o rl -- python $GRAFT_REPO_ROOT/scripts/probes/rl_parts.py > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_rl.log 2>&1
cd $GRAFT_REPO_ROOT
grep "ms per call" gpurun_out/r05/prof_rl.log
python - <<'PY'
import csv, glob, shutil
for f in glob.glob('/tmp/prof_rl/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, 'gpurun_out/r05/rl_kernel_stats.csv')
    for row in csv.DictReader(open(f)):
        n = row['Name']
        if 'cn::' in n:
            print(n.split('(')[0][:70], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
PY
timeout 300 python scripts/probes/rl_parts.py 2>&1 | grep "ms per call"
