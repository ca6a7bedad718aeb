cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_shard20.py tests/test_bench_size_parity.py tests/test_ring_wrap.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -4
run() { # label, env...
  label=$1; shift
  ( export "$@"; timeout 120 python bench.py --no-cpu-baseline --humans 20 --circle-radius 4 --steps 7992 --warmup 402 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill --no-r3-definition 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', round(d['value']/1e6,1), 'M; paused share', round(d['paused_env_steps']/(4096*d['steps']),4), 'ms/step', round(d['ms_per_step'],4))
" )
}
run "nocache list1024       " CROWDNAV_AMD_SCENARIO_CACHE=0
run "nocache list512        " CROWDNAV_AMD_SCENARIO_CACHE=0 CROWDNAV_AMD_FILL_QUEUE_WGS=512
run "nocache list256        " CROWDNAV_AMD_SCENARIO_CACHE=0 CROWDNAV_AMD_FILL_QUEUE_WGS=256
run "nocache list2048       " CROWDNAV_AMD_SCENARIO_CACHE=0 CROWDNAV_AMD_FILL_QUEUE_WGS=2048
run "nocache per-slot grid  " CROWDNAV_AMD_SCENARIO_CACHE=0 CROWDNAV_AMD_FILL_QUEUE_WGS=0
run "nocache list1024 d96   " CROWDNAV_AMD_SCENARIO_CACHE=0 CROWDNAV_AMD_RING_DEPTH=96
run "cache   list1024       " CROWDNAV_AMD_SCENARIO_CACHE=1
run "cache   per-slot grid  " CROWDNAV_AMD_SCENARIO_CACHE=1 CROWDNAV_AMD_FILL_QUEUE_WGS=0
bash scripts/probes/async_trace.sh
