"""Summarise rocprofv3 csv output (kernel stats + per-kernel PMC averages) into one small text file."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof'
for f in sorted(glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True)):
    print('##', f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 6:
            print(','.join(c[:60] for c in row))
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:50]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        n[k][row['Counter_Name']] += 1
    print('##', f, '(per-dispatch averages)')
    for k in acc:
        if 'cn::' in k:
            print(k, {c: round(v / n[k][c], 1) for c, v in acc[k].items()}, 'dispatches', max(n[k].values()))
