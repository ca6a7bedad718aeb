# the dynamic schedule of the shard kernel: parity, then R = 12 (dynamic vs static) and R = 4 with the asynchronous fill
# (slots reserved for generator workgroups x call length)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_shard20.py tests/test_bench_size_parity.py tests/test_kd_ties.py tests/test_ring_wrap.py tests/test_boundary.py tests/test_big_crowds.py -m gpu -q -x 2>&1 | tail -6
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
r12() { name=$1; shift; ( export "$@"; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 12 --steps 3996 --warmup 999 --chunk 999 > gpurun_out/r05/dyn_r12_$name.log 2>&1 ); echo -n "r12 $name: "; line gpurun_out/r05/dyn_r12_$name.log; }
r4() { name=$1; chunk=$2; steps=$3; shift 3; ( export "$@"; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 4 --steps $steps --warmup 501 --chunk $chunk --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill > gpurun_out/r05/dyn_r4_$name.log 2>&1 ); echo -n "r4 $name: "; line gpurun_out/r05/dyn_r4_$name.log; }
r12 dynamic CROWDNAV_AMD_SCHED_DYNAMIC=1
r12 static CROWDNAV_AMD_SCHED_DYNAMIC=0
r12 dynamic_b CROWDNAV_AMD_SCHED_DYNAMIC=1
r4 static_c999 999 5994 CROWDNAV_AMD_SCHED_DYNAMIC=0
for res in 0 256 512 768 1024; do
  r4 dyn_res${res}_c501 501 6012 CROWDNAV_AMD_DYN_RESERVE=$res
  r4 dyn_res${res}_c999 999 5994 CROWDNAV_AMD_DYN_RESERVE=$res
done
