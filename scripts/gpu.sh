# The ONE GPU-box runner: `gpurun -- 'bash scripts/gpu.sh <stage> [<stage> ...]'`.  Stages write under gpurun_out/<tag>/
# (tag = $CN_TAG, default r06); summaries that matter are copied into profiles/ by hand.  Experiment builds live under
# build/exp/ (make -C crowdnav_amd/csrc exp NAME=.. DEFS=..) and are selected with CROWDNAV_AMD_LIB.
#   tests [pytest args]   pytest -m gpu (whole suite, or the files given in $CN_TESTS)
#   smoke                 __graft_entry__.smoke()
#   bench                 bench.py default shape (1000-step launches) + the driver's shape (--steps 20 --warmup 5), with secondary
#   bench32k, benchh20    other headline-kernel sizes
#   ab                    bench lines of every build/exp/lib_ab_*.so against the in-tree library, both shapes
#   h20ab                 the same for the 20-human shard (12 m circle, 4 m circle with the asynchronous fill), reset probe, LDS granule
#   probe                 shader-clock phase probes of build/exp/lib_timing.so (5 and 20 humans), launch probe
#   sarl                  cn_sarl_select timing (scripts/sarl_bench.py) for the in-tree library and build/exp/lib_ab_sarl*.so
#   trace                 rocprofv3 --kernel-trace --stats of both bench shapes and the SARL decision
#   pmc                   separate rocprofv3 --pmc passes: FETCH / WRITE / SQ for the fused kernel (both shapes), rollout_kernel<10>,
#                         sarl_reg_kernel<4> and <16> (MFMA busy) -> ${TAG}_traffic.json
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
shopt -s nullglob
REPO=$GRAFT_REPO_ROOT; TAG=${CN_TAG:-r06}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; cd $REPO
line() { timeout 20 python scripts/bench_line.py "$1"; }
bench() { # name, [VAR=val ...] -- args
  name=$1; shift; envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
  ( [ ${#envs[@]} -gt 0 ] && export "${envs[@]}"; timeout 400 python bench.py "$@" > $OUT/$name.log 2>&1 < /dev/null )
  echo -n "$name: "; line $OUT/$name.log; }
prof() { name=$1; shift; ( cd /tmp; timeout 400 rocprofv3 "$@" > $OUT/$name.log 2>&1 < /dev/null ); echo "$name rc=$?"; }

for stage in "$@"; do case $stage in
tests)
  timeout 900 python -m pytest ${CN_TESTS:-tests} -m gpu -q -x > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  grep -v "version\|Hostname\|Librccl\|amdgpu.ids" $OUT/pytest_gpu.log | tail -n ${CN_TAIL:-25} ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 < /dev/null; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log ;;
bench)
  bench bench_driver -- --gpus 1 --steps 20 --warmup 5
  bench bench_default -- --no-cpu-baseline --no-secondary
  python - <<PY
import json
for l in open('$OUT/bench_driver.log'):
    if l.startswith('{'):
        d = json.loads(l); s = d.get('secondary') or {}
        for k in ('sarl', 'om_sarl'):
            if k in s: print(k, round(s[k]['value'] / 1e6, 3), 'M env-steps/s, select ms', round(s[k]['roofline']['select_ms'], 3), 'mfma frac', round(s[k]['roofline']['frac'], 3))
        for k in ('cadrl', 'lstm_rl'):
            if k in s: print(k, 'select ms', round(s[k]['roofline']['select_ms'], 3), 'mfma frac', round(s[k]['roofline']['frac'], 3))
        if 'sample_step' in s:
            print('sample_step', round(s['sample_step']['us_per_step'], 1), 'us per sampled step of one env,', s['sample_step']['launches_per_step'], 'launches;',
                  ', '.join('%s %.1f us (%s launches)' % (k, v['us_per_step'], v['launches_per_step']) for k, v in s['sample_step'].items() if isinstance(v, dict) and 'us_per_step' in v))
        for k, v in (s.get('h20') or {}).items():
            if isinstance(v, dict) and 'value' in v: print('h20', k, round(v['value'] / 1e6, 2), 'M env-steps/s, paused', v.get('paused_env_steps'))
        print('cpu', {k: (round(v) if isinstance(v, float) else v) for k, v in (d.get('cpu_baseline') or {}).items() if k in ('value', 'cores', 'single_core_value')})
PY
  ;;
bench32k) bench bench_32k -- --no-cpu-baseline --no-secondary --envs 32768 --steps 4000 ;;
benchh20)
  bench bench_h20_r12 -- --no-cpu-baseline --humans 20 --circle-radius 12 --steps 3996 --warmup 999 --chunk 999
  bench bench_h20_r4 -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 7992 --warmup 999 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021
  bench bench_h20_r4_async -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 7992 --warmup 402 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill
  bench bench_h20_r4_async_nocache CROWDNAV_AMD_SCENARIO_CACHE=0 -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 7992 --warmup 402 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill ;;
ab)
  for lib in "" $REPO/build/exp/lib_ab_*.so; do
    n=$(basename "${lib:-intree}" .so)
    bench ab_${n}_default CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --no-secondary --steps 4000
    bench ab_${n}_driver CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --no-secondary --steps 20 --warmup 5
    bench ab_${n}_driver2 CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --no-secondary --steps 20 --warmup 5
    [ -n "$CN_AB_32K" ] && bench ab_${n}_32k CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --no-secondary --envs 32768 --steps 4000
    [ -n "$CN_AB_H20" ] && bench ab_${n}_h20 CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --humans 20 --circle-radius 12 --steps 1500 --warmup 500 --chunk 500
  done ;;
probe)
  L=$REPO/build/exp/lib_timing.so
  ( CROWDNAV_AMD_LIB=$L timeout 120 python scripts/phase_probe.py ) 2>&1 | grep -v amdgpu | tail -n 12 | tee $OUT/phase_probe_h5.txt
  ( CROWDNAV_AMD_LIB=$L timeout 120 python scripts/phase_probe.py --humans 20 --circle-radius 12 --steps 1000 ) 2>&1 | grep -v amdgpu | tail -n 12 | tee $OUT/phase_probe_h20.txt
  timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/launch_probe.txt ;;
sarl)
  for lib in "" $REPO/build/exp/lib_ab_sarl*.so; do
    n=$(basename "${lib:-intree}" .so)
    ( export CROWDNAV_AMD_LIB=$lib; timeout 200 python scripts/sarl_bench.py ${CN_SARL_ARGS} ) 2>&1 | grep -v amdgpu.ids | tail -n 6 | sed "s/^/$n: /" | tee -a $OUT/sarl_bench.txt
  done ;;
trace)
  prof trace_default --kernel-trace --stats --output-format csv -d $OUT/trace_default -o trace -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-r3-definition --no-fill-probe
  prof trace_driver --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o trace -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-r3-definition --no-fill-probe --steps 20 --warmup 5
  prof trace_sarl --kernel-trace --stats --output-format csv -d $OUT/trace_sarl -o trace -- python $REPO/scripts/sarl_bench.py
  prof trace_om_sarl --kernel-trace --stats --output-format csv -d $OUT/trace_om_sarl -o trace -- python $REPO/scripts/sarl_bench.py --om 1
  prof trace_h20 --kernel-trace --stats --output-format csv -d $OUT/trace_h20 -o trace -- python $REPO/bench.py --no-cpu-baseline --no-r3-definition --no-fill-probe --humans 20 --circle-radius 12 --steps 2997 --warmup 999 --chunk 999
  for t in default driver sarl om_sarl h20; do python scripts/prof_summary.py $OUT/trace_$t | head -8; done ;;
pmc)
  SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
  SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"
  SQ3="SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  MF="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  declare -A CMD
  CMD[default]="$REPO/bench.py --no-cpu-baseline --no-secondary --no-r3-definition --no-fill-probe"
  CMD[driver]="$REPO/bench.py --no-cpu-baseline --no-secondary --no-r3-definition --no-fill-probe --steps 20 --warmup 5"
  # (h20 counters: the dynamic schedule is ONE dispatch per call: 4096 envs x 999 steps, the shape of stage benchh20)
  CMD[h20]="$REPO/bench.py --no-cpu-baseline --no-r3-definition --no-fill-probe --humans 20 --circle-radius 12 --steps 2997 --warmup 999 --chunk 999"
  for shape in ${CN_PMC_SHAPES:-default driver h20}; do
    prof pmc_${shape}_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${shape}_fetch -o p -- python ${CMD[$shape]}
    prof pmc_${shape}_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${shape}_write -o p -- python ${CMD[$shape]}
    prof pmc_${shape}_sq1 --pmc $SQ1 --output-format csv -d $OUT/pmc_${shape}_sq1 -o p -- python ${CMD[$shape]}
    prof pmc_${shape}_sq2 --pmc $SQ2 --output-format csv -d $OUT/pmc_${shape}_sq2 -o p -- python ${CMD[$shape]}
    prof pmc_${shape}_sq3 --pmc $SQ3 --output-format csv -d $OUT/pmc_${shape}_sq3 -o p -- python ${CMD[$shape]}
  done
  for v in sarl om_sarl; do
    a=""; [ $v = om_sarl ] && a="--om 1"
    prof pmc_${v}_mfma --pmc $MF --output-format csv -d $OUT/pmc_${v}_mfma -o p -- python $REPO/scripts/sarl_bench.py --iters 3 $a
    prof pmc_${v}_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${v}_fetch -o p -- python $REPO/scripts/sarl_bench.py --iters 3 $a
    prof pmc_${v}_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${v}_write -o p -- python $REPO/scripts/sarl_bench.py --iters 3 $a
  done
  rm -f $OUT/${TAG}_traffic.json
  P="python scripts/pmc_to_traffic.py $OUT/${TAG}_traffic.json"
  $P 4096 5 1000 rollout_fused_kernel tail $OUT/pmc_default_fetch $OUT/pmc_default_write $OUT/pmc_default_sq1 $OUT/pmc_default_sq2 $OUT/pmc_default_sq3 > /dev/null
  $P 4096 5 20 rollout_fused_kernel 2 $OUT/pmc_driver_fetch $OUT/pmc_driver_write $OUT/pmc_driver_sq1 $OUT/pmc_driver_sq2 $OUT/pmc_driver_sq3 > /dev/null
  CN_PMC_RADIUS=12 $P 4096 20 999 rollout_kernel tail $OUT/pmc_h20_fetch $OUT/pmc_h20_write $OUT/pmc_h20_sq1 $OUT/pmc_h20_sq2 $OUT/pmc_h20_sq3 > /dev/null
  python scripts/prof_summary.py $OUT/pmc_sarl_mfma | tail -n 4; python scripts/prof_summary.py $OUT/pmc_om_sarl_mfma | tail -n 4
  head -c 1500 $OUT/${TAG}_traffic.json ;;
h20ab)
  # the 20-human shard's kernel and the scenario generator: every build/exp/lib_ab_*.so against the in-tree library
  [ -x $REPO/build/exp/lds_granule ] && $REPO/build/exp/lds_granule | tee $OUT/lds_granule.txt
  for lib in "" $REPO/build/exp/lib_ab_*.so; do
    n=$(basename "${lib:-intree}" .so)
    bench h20_${n}_r12 CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --humans 20 --circle-radius 12 --steps 1500 --warmup 500 --chunk 500
    bench h20_${n}_r4_async CROWDNAV_AMD_LIB=$lib -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 2000 --warmup 500 --chunk 1000 --preroll 100 --seed-base 1000 --seed-mod 1021 --async-fill
    ( export CROWDNAV_AMD_LIB=$lib; timeout 120 python scripts/reset_probe.py 22 2>&1 | grep "reset ms" | sed "s/^/$n: /" | tee -a $OUT/reset_probe.txt )
  done
  for envs in ${CN_H20_ENVS:-}; do bench h20_intree_r12_envs$envs -- --no-cpu-baseline --humans 20 --circle-radius 12 --envs $envs --steps 1500 --warmup 500 --chunk 500; done ;;
gen)
  # the scenario generators: trig bound, parity of everything that resets, and the 20-human shard with every scenario generated afresh
  mkdir -p build/exp; hipcc --offload-arch=gfx950 -O2 scripts/probes/trig_error.hip -o build/exp/trig_error 2>/dev/null && build/exp/trig_error | tee $OUT/trig_error.txt
  timeout 900 python -m pytest tests/test_generator_trig.py tests/test_gpu_parity.py tests/test_shard20.py tests/test_ring_wrap.py tests/test_mixed.py tests/test_bench_size_parity.py tests/test_big_crowds.py -m gpu -q -x 2>&1 | grep -vE "version|Hostname|Librccl|amdgpu.ids" | tail -8
  bench bench_h20_r4_async_nocache CROWDNAV_AMD_SCENARIO_CACHE=0 -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 7992 --warmup 402 --chunk 999 --preroll 99 --seed-base 1000 --seed-mod 1021 --async-fill
  bench bench_h20_r4_async_train CROWDNAV_AMD_SCENARIO_CACHE=0 -- --no-cpu-baseline --humans 20 --circle-radius 4 --steps 5994 --warmup 402 --chunk 999 --preroll 99 --async-fill
  ( timeout 120 python scripts/reset_probe.py 22 2>&1 | grep "reset ms" | tee $OUT/reset_probe.txt ) ;;
pmcnet)
  # MFMA-pipe counters of the value networks (one --pmc pass each) + their kernel-trace durations: SQ_VALU_MFMA_BUSY_CYCLES
  # over the SIMD-cycles of the kernel = how busy the matrix pipe is in EXECUTED terms (padding included)
  MF="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  declare -A NET
  NET[sarl]="$REPO/scripts/sarl_bench.py --iters 3"; NET[om_sarl]="$REPO/scripts/sarl_bench.py --iters 3 --om 1"
  NET[cadrl]="$REPO/scripts/policy_bench.py --policy cadrl --iters 3"; NET[lstm_rl]="$REPO/scripts/policy_bench.py --policy lstm_rl --iters 3"
  NET[lstm_rl2]="$REPO/scripts/policy_bench.py --policy lstm_rl2 --iters 3"  # with_interaction_module = true (ValueNetwork2)
  for v in sarl om_sarl cadrl lstm_rl lstm_rl2; do
    prof pmc_${v}_mfma --pmc $MF --output-format csv -d $OUT/pmc_${v}_mfma -o p -- python ${NET[$v]}
    prof trace_${v} --kernel-trace --stats --output-format csv -d $OUT/trace_${v} -o trace -- python ${NET[$v]}
  done
  python scripts/mfma_busy.py $OUT | tee $OUT/pmc_networks_summary.txt ;;
flips)
  rm -f $OUT/argmax_flips.jsonl
  CROWDNAV_AMD_ARGMAX_REPORT=$OUT/argmax_flips.jsonl timeout 600 python -m pytest tests/test_sarl.py tests/test_big_crowds.py tests/test_noquery.py tests/test_mixed.py -m gpu -q 2>&1 | tail -n 2
  python - <<PY
import json
rows = [json.loads(l) for l in open('$OUT/argmax_flips.jsonl')]
print('decisions', sum(r['decisions'] for r in rows), 'arg-max flips vs the reference', sum(r['argmax_flips'] for r in rows))
for r in rows: print(' %-110s %4d decisions %3d flips  largest reference gap of a flip %.2e' % (r['fixture'][:110], r['decisions'], r['argmax_flips'], r['largest_reference_gap_of_a_flip']))
PY
  ;;
config5)
  # BASELINE configs[4] on the reference's own schedule (train.config: 3000 IL episodes + 50 epochs, 10 000 single-episode RL
  # sampling calls with 100 SGD batches each, evaluations): ~14 minutes, of which the in-scope sampling is ~35 s
  ( timeout 1500 python examples/train_sarl.py --gpu --seed 0 --timing-json $OUT/config5.json > $OUT/config5.log 2>&1 < /dev/null ); echo "config5 rc=$?"
  grep -E "TEST|VAL" $OUT/config5.log | tail -n 4
  python -c "import json; d = json.load(open('$OUT/config5.json')); print({k: round(v, 2) if isinstance(v, float) else v for k, v in d['timing'].items()})" ;;
*) echo "unknown stage $stage" ;;
esac; done
