"""rocprofv3 PMC csv files -> one record of profiles/r04_traffic.json (what bench.py prints as roofline.traffic / issue_roofline).

    python scripts/pmc_to_traffic.py <out.json> <envs> <humans> <steps_per_launch> <kernel substring> <dispatch index, or 'tail'> <dir> [<dir> ...]

Each <dir> holds one rocprofv3 --pmc pass (counter_collection.csv).  The rollout kernel's dispatches are taken in launch
order; `dispatch index` picks the one whose counters describe the timed launch shape (e.g. 2 = the third cn_rollout call of
`bench.py --steps 20 --warmup 5`: pre-roll, warm-up, timed), 'tail' averages all but the first two (pre-roll + warm-up)."""
import csv
import glob
import json
import os
import sys

out, envs, humans, spl, kernel, which = sys.argv[1:7]
rec = dict(envs=int(envs), humans=int(humans), steps_per_launch=int(spl), circle_radius=float(os.environ.get('CN_PMC_RADIUS', 4.0)),
           kernel=kernel, dispatches=which,
           source='rocprofv3 --pmc, separate passes per counter group (scripts/gpu.sh pmc); per-dispatch values of the '
                  'rollout kernel')
vals = {}
for d in sys.argv[7:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if kernel not in row['Kernel_Name']:
                continue
            per.setdefault(row['Counter_Name'], []).append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
        for name, lst in per.items():
            lst.sort()
            v = [x for _, x in lst]
            pick = v[2:] if which == 'tail' else [v[int(which)]]
            vals[name] = sum(pick) / len(pick)
for k, v in sorted(vals.items()):
    rec[k.lower()] = v
if 'fetch_size' in rec:
    rec['fetch_size_kb'] = rec.pop('fetch_size')
if 'write_size' in rec:
    rec['write_size_kb'] = rec.pop('write_size')
n = rec['envs'] * rec['steps_per_launch']
if 'sq_insts_valu' in rec:
    rec['valu_per_env_step'] = rec['sq_insts_valu'] / n
    f64 = sum(rec.get(k, 0.0) for k in ('sq_insts_valu_add_f64', 'sq_insts_valu_mul_f64', 'sq_insts_valu_fma_f64',
                                        'sq_insts_valu_trans_f64'))
    if f64:
        rec['f64_share'] = f64 / rec['sq_insts_valu']
    tr = rec.get('sq_insts_valu_trans_f32', 0.0) + rec.get('sq_insts_valu_trans_f64', 0.0)
    if tr:
        rec['trans_share'] = tr / rec['sq_insts_valu']
if 'sq_insts_salu' in rec and rec.get('sq_insts_valu'):
    rec['salu_per_valu'] = rec['sq_insts_salu'] / rec['sq_insts_valu']
if 'sq_insts' in rec:
    # round 6 (profiles/r06_valu_rate.txt): ONE wave issues an instruction of any class every ~5 cycles, whatever the SIMD's
    # pipes could take — the per-wave issue slots are what a latency-bound kernel at 2-3 waves per SIMD runs out of first.
    # issue_slot_share = instructions x 5 cycles over the waves' resident cycles (SQ_WAVE_CYCLES counts quad-cycles)
    rec['insts_per_env_step'] = rec['sq_insts'] / n
    if rec.get('sq_wave_cycles'):
        rec['issue_slot_share'] = 5.0 * rec['sq_insts'] / (4.0 * rec['sq_wave_cycles'])
if 'sq_thread_cycles_valu' in rec and 'sq_active_inst_valu' in rec and rec['sq_active_inst_valu']:
    # lanes active per VALU issue cycle / 64 (both counters in quad-cycles of the same unit)
    rec['lane_occupancy'] = rec['sq_thread_cycles_valu'] / (64.0 * rec['sq_active_inst_valu'])
doc = json.load(open(out)) if os.path.exists(out) else {'profiles': []}
key = lambda p: (p['envs'], p['humans'], p['steps_per_launch'], p.get('circle_radius', 4.0))  # noqa: E731
doc['profiles'] = [p for p in doc['profiles'] if key(p) != key(rec)]
doc['profiles'].append(rec)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
doc['csrc_sha'] = bench.csrc_sha()  # the rollout kernels these counters were collected on (bench.py: pmc_provenance)
json.dump(doc, open(out, 'w'), indent=1)
print(json.dumps(rec, indent=1))
