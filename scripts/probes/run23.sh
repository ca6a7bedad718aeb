# full GPU suite, then BASELINE configs[4] on the reference's train.config schedule with cn_sarl_sample_step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
bash scripts/gpu.sh tests
timeout 1500 python examples/train_sarl.py --gpu --seed 0 --timing-json gpurun_out/r05/config5.json > gpurun_out/r05/config5.log 2>&1; echo "train rc=$?"
grep -E "TEST|VAL" gpurun_out/r05/config5.log | tail -4
python -c "
import json; d=json.load(open('gpurun_out/r05/config5.json')); print(d['timing']); print(d['stats'] if not isinstance(d['stats'], dict) else {k: d['stats'][k] for k in list(d['stats'])[:6]})"
CN_TAIL=4 bash scripts/gpu.sh smoke bench
