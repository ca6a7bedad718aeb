cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_shard20.py tests/test_bench_size_parity.py tests/test_kd_ties.py tests/test_ring_wrap.py tests/test_boundary.py tests/test_big_crowds.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), 'M env-steps/s, paused share', round(d['paused_env_steps']/(d['config']['envs_per_gpu']*d['steps']),3), 'ms/step', round(d['ms_per_step'],4))
PY
}
r12() { name=$1; shift; ( export "$@"; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 12 --steps 3996 --warmup 999 --chunk 999 > gpurun_out/r05/dyn_r12_$name.log 2>&1 ); echo -n "r12 $name: "; line gpurun_out/r05/dyn_r12_$name.log; }
for v in 3 6 9 3 6; do r12 visits$v CROWDNAV_AMD_DYN_VISITS=$v; done
( export CROWDNAV_AMD_DYN_VISITS=6; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 12 --steps 3000 --warmup 500 --chunk 500 > gpurun_out/r05/dyn_r12_c500_v6.log 2>&1 ); echo -n "r12 chunk500 visits6: "; line gpurun_out/r05/dyn_r12_c500_v6.log
( export CROWDNAV_AMD_DYN_VISITS=3; python bench.py --no-cpu-baseline --no-r3-definition --humans 20 --circle-radius 12 --steps 3000 --warmup 500 --chunk 500 > gpurun_out/r05/dyn_r12_c500_v3.log 2>&1 ); echo -n "r12 chunk500 visits3: "; line gpurun_out/r05/dyn_r12_c500_v3.log
