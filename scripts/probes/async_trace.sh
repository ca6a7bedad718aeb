# kernel trace of the 20-human shard on the 4 m circle under the asynchronous fill: how long the fill kernel and the rollout kernel
# of a 999-step call run, and how they overlap (scenario cache on / off)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
for c in 0; do
  CROWDNAV_AMD_SCENARIO_CACHE=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_async_c$c -o t -- python $REPO/bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 3996 --warmup 402 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill --no-r3-definition > $OUT/trace_async_c$c.log 2>&1
  echo "cache $c: $(grep -o '"value": [0-9.e+]*' $OUT/trace_async_c$c.log | head -1)"
  python - <<PY
import csv, glob
f = glob.glob('$OUT/trace_async_c$c/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in [r for r in rows if "cn::" in r["Kernel_Name"]][-16:]:
    print('%-60s start %10.3f ms  dur %9.3f ms  grid %s' % (r['Kernel_Name'][:60], (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Grid_Size')))
PY
done
