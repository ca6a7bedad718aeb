# R = 4 with the asynchronous fill: how fast is ONE round of the chip with head-room for generator workgroups?  (a shard of 2560 /
# 2816 / 3072 envs = 10 / 11 / 12 step workgroups per CU in one plain launch; 4096 with and without the 3-of-4 schedule)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 4 --steps 5994 --warmup 501 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill "$@" > gpurun_out/r05/r4_$name.log 2>&1; echo -n "$name: "; python - gpurun_out/r05/r4_$name.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
for e in 2560 2816 3072; do run envs$e --envs $e; done
run envs4096_sched
CROWDNAV_AMD_SCHED_MIN_STEPS=1000000000 run envs4096_plain
run envs4096_sched_b
