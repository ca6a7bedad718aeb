#!/bin/bash
# Kernel durations of the single-episode sampling calls with and without the schedule's 100 SGD batches between them
# (rocprofv3 --kernel-trace --stats on probes/rl_parts.py): what runs slower "in situ", and by how much
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
for b in 1 100; do
  rm -rf /tmp/rlt_$b
  CN_RL_PARTS_BATCHES=$b CN_RL_PARTS_EPISODES=80 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rlt_$b -o t -- python $R/scripts/probes/rl_parts.py > /tmp/rlt_$b.log 2>&1
  f=$(find /tmp/rlt_$b -name '*kernel_stats.csv' | head -1)
  echo "== $b SGD batch(es) per episode: $(tail -1 /tmp/rlt_$b.log | cut -c1-200)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'cn::' in n or 'rocclr' in n:
        print('%-60s calls %6s avg %9.1f us  total %8.2f ms' % (n.split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done 2>&1 | tee $R/gpurun_out/r06/rl_insitu_trace.txt
