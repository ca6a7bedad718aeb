import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(round(d['value'] / 1e6, 1), 'M env-steps/s', 'ms/step', round(d['ms_per_step'], 5), 'launch_ms',
              round(d['roofline'].get('avg_launch_ms', 0), 4), 'frac', round(d['roofline']['frac'], 4),
              d.get('cpu_baseline', {}).get('value'))
    elif 'Error' in l or 'error' in l:
        print(l.strip()[:200])
