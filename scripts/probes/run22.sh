cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np -o np -- python $GRAFT_REPO_ROOT/scripts/probes/narrow_probe.py 1 > /tmp/np.log 2>&1
grep "per call\|per step" /tmp/np.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prof_np/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        n = row['Name']
        if 'cn::' in n:
            print(n.split('(')[0][:70], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
# gaps between consecutive kernels in the streamed loop
for f in glob.glob('/tmp/prof_np/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if 'cn::' in r['Kernel_Name']]
    tail = rows[-300:]
    import collections
    gaps = collections.defaultdict(list)
    for a, b in zip(tail, tail[1:]):
        gaps[(a['Kernel_Name'].split('(')[0][-30:], b['Kernel_Name'].split('(')[0][-30:])].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
    for k, v in gaps.items():
        print(k, len(v), 'median gap ns', sorted(v)[len(v) // 2])
PY
