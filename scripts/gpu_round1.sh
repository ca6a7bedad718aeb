mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
for E in 1 2 3 5 10; do CROWDNAV_AMD_ENVS_PER_WAVE=$E timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_E$E.log 2>&1; done
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -n 4 gpurun_out/pytest_gpu.log gpurun_out/smoke.log; for f in gpurun_out/bench*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,1),'M env-steps/s', 'ms/step',round(d['ms_per_step'],5),'launch_ms',round(d['roofline']['avg_launch_ms'],4),'frac',round(d['roofline']['frac'],4), d.get('cpu_baseline',{}).get('value'))
PY
done
