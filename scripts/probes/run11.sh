cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_shard20.py tests/test_bench_size_parity.py tests/test_kd_ties.py tests/test_ring_wrap.py tests/test_boundary.py tests/test_big_crowds.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
r12() { name=$1; chunk=$2; steps=$3; shift 3; ( export "$@"; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 12 --steps $steps --warmup $chunk --chunk $chunk > gpurun_out/r05/dyn_r12_$name.log 2>&1 ); echo -n "r12 $name: "; line gpurun_out/r05/dyn_r12_$name.log; }
r4() { name=$1; chunk=$2; steps=$3; shift 3; ( export "$@"; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 4 --steps $steps --warmup 501 --chunk $chunk --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill > gpurun_out/r05/dyn_r4_$name.log 2>&1 ); echo -n "r4 $name: "; line gpurun_out/r05/dyn_r4_$name.log; }
for v in 9 12 18 27 54 108; do r12 c999_v$v 999 3996 CROWDNAV_AMD_DYN_VISITS=$v; done
for v in 9 18 36; do r12 c500_v$v 500 3000 CROWDNAV_AMD_DYN_VISITS=$v; done
for v in 9 27; do r4 c999_v$v 999 5994 CROWDNAV_AMD_DYN_VISITS=$v; r4 c501_v$v 501 6012 CROWDNAV_AMD_DYN_VISITS=$v; done
