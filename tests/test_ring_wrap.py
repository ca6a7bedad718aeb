"""The scenario ring in the regime the long launches of bench.py live in: an env re-uses a ring slot every ring_depth
episodes (slot = episode ordinal % ring_depth, `rollout_fused.h` / `step_kernels.h: load_from_ring`), the fill kernels
regenerate exactly the consumed slots (`ring_fill_*_kernel`, the `ring_filled_in / _out` swap in `fill_ring_if_needed`),
an env whose ring ran dry inside a launch pauses and resumes in the next one, and the asynchronous fill publishes slots one
by one (`ring_claim` / `ring_ready` carry the episode ordinal).  With the default depth of 48 an env wraps the ring after
~1 900 steps; CROWDNAV_AMD_RING_DEPTH = 3 (5 for the wave generators) makes it wrap every few dozen steps, and one run
goes through the default ring at full depth.  Everything is compared with the oracle's rollout
(oracle/crowd_oracle.cpp: co_rollout; episode c of env b seeded seed_base + (b + j B) % seed_mod, crowd_sim.py:272-283)."""
import contextlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a MI355X'
    import crowdnav_amd
    return crowdnav_amd


def _np(t):
    return t.detach().cpu().numpy()


@contextlib.contextmanager
def ring_depth(depth):
    """cn_create reads CROWDNAV_AMD_RING_DEPTH when the engine is built."""
    old = os.environ.get('CROWDNAV_AMD_RING_DEPTH')
    if depth is None:
        os.environ.pop('CROWDNAV_AMD_RING_DEPTH', None)
    else:
        os.environ['CROWDNAV_AMD_RING_DEPTH'] = str(depth)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop('CROWDNAV_AMD_RING_DEPTH', None)
        else:
            os.environ['CROWDNAV_AMD_RING_DEPTH'] = old


def _oracle(oracle_mod, B, steps, K, **cfg):
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    o.reset(1000 + np.arange(B))
    ep_index, cur_steps, cur_ret = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64)
    total, rec = o.rollout(steps, 1000, 500, K, ep_index, cur_steps, cur_ret)
    return o, total, rec, cur_steps, cur_ret


def _engine(amd, B, launches, K, depth, flags=0, **cfg):
    with ring_depth(depth):
        eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, flags=flags, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K)
    for n in launches:
        eng.rollout(n)
    eng.sync()
    return eng, bufs


def _records_in_order(bufs, rec, K):
    """every episode the engine finished is the oracle's episode of the same ordinal: outcome, length, return"""
    cnt = _np(bufs['ep_count'])
    assert (cnt <= rec['count']).all() and cnt.max() <= K
    out, steps, ret = _np(bufs['ep_outcome']), _np(bufs['ep_steps']), _np(bufs['ep_return'])
    for b in range(len(cnt)):
        k = cnt[b]
        assert np.array_equal(out[b, :k], rec['outcome'][b, :k]), b
        assert np.array_equal(steps[b, :k], rec['steps'][b, :k]), b
        assert np.allclose(ret[b, :k], rec['ret'][b, :k], rtol=0, atol=1e-12), b
    return cnt


def _mixed(total, lengths):
    """launch lengths cycling through `lengths` until they add up to `total`"""
    out, i = [], 0
    while sum(out) < total:
        out.append(min(lengths[i % len(lengths)], total - sum(out)))
        i += 1
    return out


# (humans, circle radius, ring depth, envs, steps inside the budget, launches beyond it): the fused kernel with the
# register-only generator; the general 5-half-plane kernel with the redo pool (crowded circle: most scenarios outrun the head
# generator); the 10-half-plane kernel with one generator wave per scenario (its episodes are longer)
LONG = [1, 2, 7, 50, 3, 120, 17, 200]
GEOMETRIES = [(5, 4.0, 3, 24, 420, LONG), (8, 2.6, 3, 16, 420, LONG), (12, 6.0, 5, 12, 720, LONG + [330, 20, 400])]


@pytest.mark.parametrize('H,R,depth,B,steps,_long', GEOMETRIES)
def test_ring_wraps_many_times_inside_its_budget(amd, oracle_mod, H, R, depth, B, steps, _long):
    """Launches never longer than the ring is deep (an env consumes at most one scenario per transition, so none can run
    dry): 420 / 720 steps in launches of 1 .. depth steps, every env ~10 episodes = 2-3 trips round the ring, and everything —
    episodes, the running episode's counters and return, the end state — equals the oracle's, transition for transition."""
    K = 64
    cfg = dict(num_humans=H, circle_radius=R, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, steps, K, **cfg)
    assert rec['count'].min() > 2 * depth  # every env went round the ring at least twice
    eng, bufs = _engine(amd, B, _mixed(steps, list(range(1, depth + 1)) + [depth, 1]), K, depth, **cfg)
    assert int(_np(bufs['transitions'])[0]) == total == B * steps
    cnt = _records_in_order(bufs, rec, K)
    assert np.array_equal(cnt, rec['count'])
    assert np.array_equal(_np(bufs['cur_steps']), cur_steps)
    assert np.allclose(_np(bufs['cur_return']), cur_ret, rtol=0, atol=1e-12)
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


@pytest.mark.parametrize('H,R,depth,B,_steps,launches', GEOMETRIES)
def test_ring_runs_dry_inside_long_launches(amd, oracle_mod, H, R, depth, B, _steps, launches):
    """Launches much longer than the ring's budget (1, 2, 7, 50, 3, 120, 17, 200 steps): an env that has played its
    `depth` resident scenarios pauses until the next launch refills the ring.  Paused envs are behind the oracle, never
    ahead; every episode they did finish is the oracle's, in order, to the last bit of outcome and length; the transitions counter is exactly the
    sum of what the envs ran; and envs that never paused end in the oracle's state."""
    steps, K = sum(launches), 64
    cfg = dict(num_humans=H, circle_radius=R, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, steps, K, **cfg)
    eng, bufs = _engine(amd, B, launches, K, depth, **cfg)
    cnt = _records_in_order(bufs, rec, K)
    ran = _np(bufs['ep_steps']).astype(np.int64)
    per_env = np.array([ran[b, :cnt[b]].sum() for b in range(B)]) + _np(bufs['cur_steps'])
    assert int(_np(bufs['transitions'])[0]) == per_env.sum() and (per_env <= steps).all()
    assert (per_env < steps).any(), 'no env ran its ring dry: the launches are not long enough for this test'
    full = per_env == steps
    state, want = _np(eng.get_state()[0]), o.get_state()[0]
    if full.any():
        assert np.abs(state[full] - want[full]).max() <= 1e-9
        assert np.array_equal(_np(bufs['cur_steps'])[full], cur_steps[full])
    # a waiting env holds the end state of its last episode and has consumed exactly `cnt` scenarios
    active = _np(bufs['active'])
    assert set(np.unique(active).tolist()) <= {1, 2}
    # ... and resumes: short launches inside the budget until every env has caught up with a later oracle time
    extra = 3 * depth
    for _ in range(extra):
        eng.rollout(1)
    eng.sync()
    assert (_np(bufs['active']) == 1).all()  # one-step launches never leave an env waiting


def test_full_depth_ring_2500_steps(amd, oracle_mod):
    """The default ring (48 episodes ahead) at full depth: 64 envs x 2 500 steps in launches of up to 400 steps — ~70
    episodes per env, so every env re-uses its first slots — against the oracle: records, running counters, end state."""
    B, steps, K = 64, 2500, 128
    cfg = dict(num_humans=5, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, steps, K, **cfg)
    assert rec['count'].min() > 48
    eng, bufs = _engine(amd, B, _mixed(steps, [400, 37, 400, 1, 250]), K, None, **cfg)
    assert int(_np(bufs['transitions'])[0]) == total == B * steps
    cnt = _records_in_order(bufs, rec, K)
    assert np.array_equal(cnt, rec['count'])
    assert np.array_equal(_np(bufs['cur_steps']), cur_steps)
    assert np.allclose(_np(bufs['cur_return']), cur_ret, rtol=0, atol=1e-12)
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


@pytest.mark.parametrize('depth,launches', [(5, [10] * 42), (3, [1, 2, 7, 50, 3, 120, 17, 200, 20])])
def test_asynchronous_fill_with_a_shallow_ring(amd, oracle_mod, depth, launches):
    """CN_FLAG_ASYNC_SCENARIO_FILL with a ring of 5 / 3 slots: every slot is claimed, generated and published many times
    over (ring_claim / ring_ready carry episode ordinals, step_kernels.h: ring_fill_wave_async_kernel), fill launches of
    consecutive rollout calls overlap on the side streams.  Timing decides when an env pauses, never what it plays: its
    finished episodes are the oracle's, in order, bit for bit; and the run makes progress (no env starves)."""
    B, K = 48, 64
    steps = sum(launches)
    cfg = dict(num_humans=10, circle_radius=3.2, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, steps, K, **cfg)
    eng, bufs = _engine(amd, B, launches, K, depth, flags=amd.FLAG_ASYNC_SCENARIO_FILL, **cfg)
    cnt = _records_in_order(bufs, rec, K)
    assert cnt.min() >= 1 and cnt.sum() >= 0.25 * rec['count'].sum()
    ran = _np(bufs['ep_steps']).astype(np.int64)
    per_env = np.array([ran[b, :cnt[b]].sum() for b in range(B)]) + _np(bufs['cur_steps'])
    assert int(_np(bufs['transitions'])[0]) == per_env.sum()
    # a second rollout on the same engine: cn_rollout_begin waits for the fill kernels of the first (they read its io
    # block and buffers) before it restarts the bookkeeping; the new episodes are the same ones again
    bufs2 = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K)
    for n in launches:
        eng.rollout(n)
    eng.sync()
    _records_in_order(bufs2, rec, K)


def test_rollout_step_with_a_shallow_ring(amd, oracle_mod):
    """cn_rollout_step (the value-network rollouts' transition: one bookkept step per call, actions from outside) with a
    3-deep ring: 400 calls, every env ~12 episodes, vs the oracle stepped and reset by hand."""
    import torch
    n, steps, K = 20, 400, 64
    cfg = dict(num_humans=5, robot_visible=1)
    with ring_depth(3):
        eng = amd.BatchedCrowdSim(num_envs=n, robot_policy=amd.ROBOT_EXTERNAL, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, record_capacity=K)
    o = oracle_mod.CrowdOracle(num_envs=n, robot_policy=0, **cfg)
    o.reset(1000 + np.arange(n))
    ep, cur = np.zeros(n, np.int64), np.zeros(n, np.int64)
    rec_steps, rec_outcome = [[] for _ in range(n)], [[] for _ in range(n)]
    rng = np.random.RandomState(3)
    for _ in range(steps):
        s = o.get_state()[0]
        to_goal = s[:, 0, 4:6] - s[:, 0, 0:2]
        act = to_goal / np.maximum(np.linalg.norm(to_goal, axis=1, keepdims=True), 1.0) + rng.uniform(-0.3, 0.3, (n, 2))
        eng.rollout_step(torch.from_numpy(act))
        out = o.step(act, update=True)
        cur += 1
        done = out['done'] != 0
        for b in np.nonzero(done)[0]:
            rec_steps[b].append(int(cur[b]))
            rec_outcome[b].append(int(out['info'][b]))
        ep += done
        cur[done] = 0
        if done.any():
            o.reset(1000 + (np.arange(n) + ep * n) % 500, mask=done.astype(np.uint8))
    eng.sync()
    assert ep.min() > 6  # twice round the ring
    assert int(_np(bufs['transitions'])[0]) == n * steps
    assert np.array_equal(_np(bufs['ep_count']), ep) and np.array_equal(_np(bufs['cur_steps']), cur)
    got_steps, got_out = _np(bufs['ep_steps']), _np(bufs['ep_outcome'])
    for b in range(n):
        k = len(rec_steps[b])
        assert got_steps[b, :k].tolist() == rec_steps[b] and got_out[b, :k].tolist() == rec_outcome[b]
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


@pytest.mark.parametrize('B,H,launches', [(1, 5, [60, 60]), (5, 3, [150]), (300, 5, [100, 20, 1]), (37, 20, [60, 45])])
def test_launch_epilogue_equals_the_boundary_kernels(amd, B, H, launches):
    """The rollout kernels' own tail (step_kernels.h: rollout_epilogue; cn_rollout_io.summary / .blocks): record blocks
    identical to cn_rollout_records', sums equal to cn_records_summary's (counts exactly, float sums to rounding: the
    summation tree differs), the transitions counter exact — for one workgroup, a partial last wave, several arrival
    groups and the one-env-per-workgroup geometry of 20 humans; and bitwise the same on a second run."""
    import torch
    K = 4

    def run():
        eng = amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=amd.ROBOT_ORCA, robot_visible=1,
                                  circle_radius=4.0 if H <= 5 else 9.0)
        bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K, boundary_records=2)
        for n in launches:
            eng.rollout(n)
        eng.sync()
        return eng, bufs

    eng, bufs = run()
    assert int(_np(bufs['transitions'])[0]) == B * sum(launches)
    assert torch.equal(bufs['blocks'], eng.rollout_records(max_records=2))
    want = _np(eng.records_summary(eng.rollout_records(), record_capacity=K))
    got = _np(bufs['summary'])
    assert np.array_equal(got[:5], want[:5]) and got[7] == want[7] and got[0] == _np(bufs['ep_count']).sum()
    assert np.abs(got[5:7] - want[5:7]).max() <= 1e-9 * max(1.0, np.abs(want[5:7]).max())
    _, bufs2 = run()
    assert torch.equal(bufs['summary'], bufs2['summary']) and torch.equal(bufs['blocks'], bufs2['blocks'])
