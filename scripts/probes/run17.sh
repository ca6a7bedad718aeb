# the scenario cache of the wave generators: parity, then the 4 m circle (bounded 'test' seeds) with the cache on / off, synchronous and
# asynchronous fill
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_shard20.py tests/test_bench_size_parity.py tests/test_ring_wrap.py tests/test_boundary.py tests/test_big_crowds.py tests/test_gpu_parity.py tests/test_kd_ties.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
r4() { name=$1; shift; ( export CROWDNAV_AMD_SCENARIO_CACHE=$1; shift; python bench.py --no-cpu-baseline --no-r3-definition --no-fill-probe --humans 20 --circle-radius 4 --steps 5994 --warmup 999 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 "$@" > gpurun_out/r05/cache_r4_$name.log 2>&1 ); echo -n "r4 $name: "; line gpurun_out/r05/cache_r4_$name.log; }
r4 async_cache 1 --async-fill
r4 async_nocache 0 --async-fill
r4 sync_cache 1
r4 async_cache_b 1 --async-fill
r4 sync_cache_b 1
