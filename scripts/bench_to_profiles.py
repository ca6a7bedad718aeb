"""Copies the JSON line of gpurun_out/<tag>/bench_*.log into profiles/<tag>_bench_*.json (the committed record of a bench run)."""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag)
for f in sorted(os.listdir(src)):
    if f.startswith('bench_') and f.endswith('.log'):
        lines = [l for l in open(os.path.join(src, f)) if l.startswith('{')]
        if lines:
            out = os.path.join(root, 'profiles', '%s_%s.json' % (tag, f[:-4]))
            json.dump(json.loads(lines[-1]), open(out, 'w'), indent=1)
            print(out)
