"""bench.py's JSON line must be true under ANY --steps / --chunk: PMC-derived fields (roofline.traffic, issue_roofline) are
only attached when a committed rocprofv3 record describes exactly the launch shape that was timed, and the vector-issue
peak is the MI355X figure (SIMD-32: one wave64 VALU instruction per 2 cycles)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_fields_only_for_the_profiled_launch_shape(tmp_path, monkeypatch):
    import bench
    prof = tmp_path / 'traffic.json'
    prof.write_text(json.dumps({'profiles': [
        dict(envs=4096, humans=5, steps_per_launch=1000, fetch_size_kb=1000.0, write_size_kb=500.0, sq_insts_valu=2.0e9),
        dict(envs=4096, humans=5, steps_per_launch=20, fetch_size_kb=100.0, write_size_kb=50.0, sq_insts_valu=4.0e7)]}))
    monkeypatch.setattr(bench, 'PMC_PROFILE', str(prof))
    assert bench.pmc_profile(4096, 5, 100) is None          # a 100-step launch has no profile: no traffic, no issue roofline
    assert bench.pmc_traffic_bytes(None) is None and bench.pmc_issue(None, 4096, 100, 1e-3) is None
    short = bench.pmc_profile(4096, 5, 20)
    assert bench.pmc_traffic_bytes(short) == (2 * 100.0 + 50.0) * 1024   # FETCH_SIZE doubled (gfx950 correction)
    issue = bench.pmc_issue(short, 4096, 20, 100e-6)
    assert issue['peak'] == 1024 * 2.4 / 2                   # G wave-instructions/s: 2 cycles per wave64 VALU op
    assert abs(issue['achieved'] - 4.0e7 / 100e-6 / 1e9) < 1e-9 and 0.0 < issue['frac'] < 1.0
    assert abs(issue['valu_per_env_step'] - 4.0e7 / (4096 * 20)) < 1e-9
    long = bench.pmc_profile(4096, 5, 1000)
    assert long['sq_insts_valu'] == 2.0e9 and bench.pmc_profile(4096, 20, 1000) is None


def test_committed_pmc_profile_is_well_formed():
    import bench
    if not os.path.exists(bench.PMC_PROFILE):
        return
    doc = json.load(open(bench.PMC_PROFILE))
    shapes = [(p['envs'], p['humans'], p['steps_per_launch'], p.get('circle_radius', 4.0)) for p in doc['profiles']]
    assert len(shapes) == len(set(shapes))
    for p in doc['profiles']:
        assert p['fetch_size_kb'] > 0 and p['write_size_kb'] > 0 and p['sq_insts_valu'] > 0


def test_algorithmic_bytes_follow_the_survey():
    import bench
    assert bench.algorithmic_bytes_per_env_step(5) == 458 and bench.algorithmic_bytes_per_env_step(20) == 1538


def test_issue_roofline_names_its_counter_source_and_is_dropped_when_stale(tmp_path, monkeypatch):
    """issue_roofline divides COMMITTED SQ_INSTS_VALU by this run's launch time.  The line says where the
    counters come from, and a kernel edit without a re-profile drops the field instead of printing a stale fraction."""
    import bench
    prof = tmp_path / 'traffic.json'
    rec = dict(envs=4096, humans=5, steps_per_launch=20, fetch_size_kb=100.0, write_size_kb=50.0, sq_insts_valu=4.0e7)
    # (1) stamped profile (scripts/pmc_to_traffic.py writes csrc_sha): fresh when the stamp equals the sources' hash
    prof.write_text(json.dumps({'csrc_sha': bench.csrc_sha(), 'profiles': [rec]}))
    monkeypatch.setattr(bench, 'PMC_PROFILE', str(prof))
    src = bench.pmc_provenance()
    assert src['stale'] is False and src['csrc_sha_profiled'] == bench.csrc_sha()
    issue = bench.pmc_issue(bench.pmc_profile(4096, 5, 20), 4096, 20, 100e-6)
    assert issue['counters_from'] == src and 'NOT measured in this run' in issue['note']
    # (2) the kernels changed since: no issue_roofline
    prof.write_text(json.dumps({'csrc_sha': '0' * 16, 'profiles': [rec]}))
    assert bench.pmc_provenance()['stale'] is True
    assert bench.pmc_issue(bench.pmc_profile(4096, 5, 20), 4096, 20, 100e-6) is None
    # (3) unstamped profile outside the repository: cannot tell -> stale is None or decided by git; never a crash
    prof.write_text(json.dumps({'profiles': [rec]}))
    assert bench.pmc_provenance()['stale'] in (None, True, False)


def test_committed_profile_provenance_is_decidable_in_this_checkout():
    import bench
    if not os.path.exists(bench.PMC_PROFILE):
        return
    src = bench.pmc_provenance()
    assert src is not None and src['file'].startswith('profiles/')
    assert 'csrc_sha_profiled' in src or 'profile_commit' in src or src['stale'] is None


def test_reference_python_baselines_are_timed_on_this_host_or_labelled_off_host(monkeypatch):
    """EVERY number of the line has the unmodified reference's CPU path beside it — cpu_baseline.reference_python
    (ORCA loop), secondary.{sarl,om_sarl,cadrl,lstm_rl}.cpu_baseline (robot.act -> predict, decisions/s),
    secondary.sample_step.cpu_baseline (train-phase sampling, env-steps/s), secondary.h20.cpu_baseline.reference_python and
    secondary.config5_schedule.reference_estimate_s — timed ON THE BENCH HOST when a reference copy is present
    (/root/reference here, oracle/_ref on the GPU box) by ONE subprocess run; otherwise the committed figures, which must say
    that they come from the build container."""
    import bench
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ref_harness
    assert os.path.exists(bench.REFERENCE_PYTHON_PROFILE)
    r = json.load(open(bench.REFERENCE_PYTHON_PROFILE))
    assert r['cores'] == 1 and r['value'] > 0 and 'host_cpu' in r
    assert {d['policy'] for d in r['decision']['runs']} == {'sarl', 'sarl+om', 'cadrl', 'lstm_rl', 'lstm_rl+pairwise'}
    assert r['sampling']['env_steps'] > 0 and r['crowd20']['humans'] == 20

    def check(where):
        orca = bench.reference_python_baseline()
        assert orca['kind'] == 'reference' and orca['cores'] == 1 and orca['value'] > 0 and where in orca['host']
        for pol in ('sarl', 'sarl+om', 'cadrl', 'lstm_rl', 'lstm_rl+pairwise'):
            d = bench.reference_decision_baseline(pol)
            assert d['unit'] == 'decisions/s' and d['cores'] == 1 and d['kind'] == 'reference' and where in d['host']
            assert 1.0 < d['value'] < 1000.0 and 'robot.act' in d['sample']  # ~10 decisions/s/core
        sm = bench.reference_sampling_baseline()
        assert sm['unit'] == 'env-steps/s' and sm['cores'] == 1 and where in sm['host'] and 1.0 < sm['value'] < 1000.0
        est = bench.config5_schedule_estimate()
        assert est['schedule_from'].startswith('profiles/') and est['env_steps']['rl_sample'] > 100000
        assert est['reference_estimate_s']['rl_sample'] > 100 * est['device_s']['rl_sample']
        assert est['reference_estimate_s']['il_collect'] > est['device_s']['il_collect']
        return orca

    if ref_harness.available():
        bench._REFERENCE_RUN.clear()
        bench.reference_python_run(cases=12, decisions=2, sampling_seconds=0.5)
        live = check('THIS host')
        assert 'env-steps' in live['sample']
    # no reference copy (the subprocess fails): the committed figures, labelled
    bench._REFERENCE_RUN.clear()
    monkeypatch.setattr(bench.subprocess, 'run', lambda *a, **k: (_ for _ in ()).throw(OSError('no reference')))
    off = check('(build container, NOT this host')
    assert off['value'] == r['value'] and 'no reference copy on this machine' in off['host']
    bench._REFERENCE_RUN.clear()


def test_secondary_rows_carry_their_cpu_baseline(monkeypatch):
    """the wiring of bench.secondary(): every measured row gets the reference figure of ITS quantity (stand-in measurements)"""
    import bench
    monkeypatch.setattr(bench, 'measure_sarl', lambda B, H, om, *a, **k: {
        'value': 1.0, 'unit': 'env-steps/s', 'steps': 1, 'ms_per_step': 1.0, 'roofline': {'select_ms': 2.0},
        'config': {'workload': 'w'}})
    monkeypatch.setattr(bench, 'measure_policy_decision', lambda B, H, policy, lr: {'decisions_per_s': 1.0})
    monkeypatch.setattr(bench, 'measure_h20', lambda B, lr: {'r12': {}})
    monkeypatch.setattr(bench, 'measure_sample_step',
                        lambda lr, with_om=False, policy='sarl': {'value': (2.0 if with_om else 1.0) + (10.0 if policy == 'lstm_rl' else 0.0)})
    monkeypatch.setattr(bench, 'cpu_baseline_h20', lambda: {'kind': 'port'})
    seen = []
    monkeypatch.setattr(bench, 'reference_decision_baseline', lambda pol: seen.append(pol) or {'policy': pol})
    monkeypatch.setattr(bench, 'reference_sampling_baseline', lambda: {'unit': 'env-steps/s'})
    monkeypatch.setattr(bench, 'config5_schedule_estimate', lambda: {'reference_estimate_s': {}})
    out = bench.secondary(4096, 0)
    assert seen == ['sarl', 'sarl+om', 'cadrl', 'lstm_rl', 'lstm_rl+pairwise']
    assert out['sarl']['cpu_baseline'] == {'policy': 'sarl'} and out['om_sarl']['cpu_baseline'] == {'policy': 'sarl+om'}
    assert out['sarl']['decisions_per_s'] == 4096 / 2e-3
    assert out['cadrl']['cpu_baseline'] == {'policy': 'cadrl'} and out['lstm_rl']['cpu_baseline'] == {'policy': 'lstm_rl'}
    assert out['lstm_rl_pairwise']['cpu_baseline'] == {'policy': 'lstm_rl+pairwise'}
    assert out['sample_step']['cpu_baseline'] == {'unit': 'env-steps/s'} and out['h20']['cpu_baseline'] == {'kind': 'port'}
    assert out['sample_step']['value'] == 1.0 and out['sample_step']['om_sarl']['value'] == 2.0
    assert out['sample_step']['lstm_rl']['value'] == 11.0 and out['sample_step']['lstm_rl_om']['value'] == 12.0
    assert 'reference_estimate_s' in out['config5_schedule']


def test_distributed_init_failure_names_the_ipc_switch(monkeypatch):
    """the driver's 8-GPU run is the first RCCL world > 1 this code sees.  If torch.distributed cannot come up,
    the exit message names HSA_ENABLE_IPC_MODE_LEGACY and the value this process saw."""
    import pytest
    import torch.distributed as dist
    import bench

    def boom(*a, **k):
        raise RuntimeError('hipIpcGetMemHandle: invalid argument')
    monkeypatch.setattr(dist, 'init_process_group', boom)
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '1')
    with pytest.raises(SystemExit) as ei:
        bench.init_distributed('gloo', 0)
    msg = str(ei.value)
    assert 'HSA_ENABLE_IPC_MODE_LEGACY' in msg and "'1'" in msg and 'hipIpcGetMemHandle' in msg and 'dmabuf' in msg


def test_fill_probe_classifies_calls_by_the_engines_launch_counters(monkeypatch):
    """measure_fill_seconds: 1-step calls split by cn_launch_counts into calls that carried a ring fill and calls that did not;
    the fill's cost is the difference of the medians (here with a stand-in engine and stand-in events)."""
    import types
    import bench

    class Ev(object):
        clock = [0.0]

        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = Ev.clock[0]

        def elapsed_time(self, other):
            return other.t - self.t

    class Eng(object):
        def __init__(self):
            self.steps, self.fills = 0, 0

        def launch_counts(self):
            return {'ring_fills': self.fills}

        def rollout(self, n):
            self.steps += n
            fill = self.steps % 48 == 1
            self.fills += int(fill)
            Ev.clock[0] += 0.017 + (0.080 if fill else 0.0)  # ms

    fake = types.SimpleNamespace(cuda=types.SimpleNamespace(Event=Ev, synchronize=lambda: None))
    monkeypatch.setitem(sys.modules, 'torch', fake)
    assert abs(bench.measure_fill_seconds(Eng(), 48) - 80e-6) < 1e-9
    monkeypatch.setenv('CROWDNAV_AMD_RING_DEPTH', '5')
    assert bench.ring_depth() == 5
