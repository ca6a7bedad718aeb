# R = 4, asynchronous fill: does the fill CADENCE (one fill launch per cn_rollout call) bound the run?  Same total steps in calls of
# 150 / 300 / 501 / 999 / 1998 steps.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 4 --warmup 501 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill "$@" > gpurun_out/r05/r4c_$name.log 2>&1; echo -n "$name: "; python - gpurun_out/r05/r4c_$name.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
run chunk150 --steps 6000 --chunk 150
run chunk300 --steps 6000 --chunk 300
run chunk501 --steps 6012 --chunk 501
run chunk999 --steps 5994 --chunk 999
run chunk1998 --steps 5994 --chunk 1998
run chunk300_b --steps 6000 --chunk 300
