"""The 20-human shard's kernel (BASELINE configs[3]: `rollout_kernel<10, false, true, true>`, step_kernels.h) since round 4:
compact LDS layout (12 workgroups per CU), step parameters / episode bookkeeping / per-episode agent constants in LDS instead of
registers (three resident waves per SIMD), and the 3-of-4 env schedule of `launch_rollout` (crowdnav_amd.hip): a call of
3 q + r steps runs as one launch of r steps over all envs and FOUR launches of q steps over 3 B / 4 workgroups, sub-launch k
leaving out env 3 - k of every group of four.  None of this may change a bit of what an env plays: everything is compared
with the oracle's rollout (oracle/crowd_oracle.cpp: co_rollout, the restatement of crowd_sim.py:317-420 + explorer.py:50-72),
and the statistics a call leaves behind (cn_rollout_io.summary / .blocks) with the boundary kernels' own."""
import numpy as np
import pytest

from test_ring_wrap import _engine, _np, _oracle, _records_in_order, amd  # noqa: F401  (amd: module fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_schedule(monkeypatch):
    """cn_create reads CROWDNAV_AMD_SCHED_FORCE: split calls of >= 48 steps even where it saves no round (a handful of envs)"""
    monkeypatch.setenv('CROWDNAV_AMD_SCHED_FORCE', '1')


@pytest.mark.parametrize('B,R,launches', [(16, 9.0, [48, 100, 7, 61]), (8, 4.0, [150, 50]), (4, 12.0, [49, 49, 49])])
def test_env_schedule_plays_the_oracles_episodes(amd, oracle_mod, B, R, launches):
    """Launch lengths on both sides of the schedule's threshold (48 steps), divisible by three and not: episodes, counters of
    the running episode, returns and the end state equal the oracle's, transition for transition."""
    K = 32
    steps = sum(launches)
    cfg = dict(num_humans=20, circle_radius=R, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, steps, K, **cfg)
    eng, bufs = _engine(amd, B, launches, K, None, **cfg)
    assert int(_np(bufs['transitions'])[0]) == total == B * steps
    cnt = _records_in_order(bufs, rec, K)
    assert np.array_equal(cnt, rec['count']) and cnt.sum() > 0
    assert np.array_equal(_np(bufs['cur_steps']), cur_steps)
    assert np.allclose(_np(bufs['cur_return']), cur_ret, rtol=0, atol=1e-12)
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


def test_env_schedule_statistics_cover_every_env(amd):
    """The last sub-launch leaves one env of every four out and reports for it as well (rollout_epilogue: extra_env): the
    sums of a scheduled call equal cn_records_summary over cn_rollout_records' blocks, the record blocks are the packed
    ones, the transitions counter is exact, and a second run leaves the same bits."""
    import torch
    B, K = 36, 4

    def run():
        eng = amd.BatchedCrowdSim(num_envs=B, num_humans=20, robot_policy=amd.ROBOT_ORCA, robot_visible=1, circle_radius=9.0)
        bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=K, boundary_records=2)
        for n in (60, 49, 90):
            eng.rollout(n)
        eng.sync()
        return eng, bufs

    eng, bufs = run()
    assert int(_np(bufs['transitions'])[0]) == B * 199
    assert torch.equal(bufs['blocks'], eng.rollout_records(max_records=2))
    want = _np(eng.records_summary(eng.rollout_records(), record_capacity=K))
    got = _np(bufs['summary'])
    assert np.array_equal(got[:5], want[:5]) and got[7] == want[7] and got[0] == _np(bufs['ep_count']).sum() > B
    assert np.abs(got[5:7] - want[5:7]).max() <= 1e-9 * max(1.0, np.abs(want[5:7]).max())
    _, bufs2 = run()
    assert torch.equal(bufs['summary'], bufs2['summary']) and torch.equal(bufs['blocks'], bufs2['blocks'])


def test_env_schedule_with_asynchronous_fill(amd, oracle_mod):
    """The schedule next to CN_FLAG_ASYNC_SCENARIO_FILL on the reference's own geometry (4 m circle): timing decides when an
    env pauses, never what it plays — its finished episodes are the oracle's, in order."""
    B, K = 16, 64
    launches = [60, 60, 120, 51]
    cfg = dict(num_humans=20, circle_radius=4.0, robot_visible=1)
    o, total, rec, cur_steps, cur_ret = _oracle(oracle_mod, B, sum(launches), K, **cfg)
    eng, bufs = _engine(amd, B, launches, K, None, flags=amd.FLAG_ASYNC_SCENARIO_FILL, **cfg)
    cnt = _records_in_order(bufs, rec, K)
    assert cnt.min() >= 1
    ran = _np(bufs['ep_steps']).astype(np.int64)
    per_env = np.array([ran[b, :cnt[b]].sum() for b in range(B)]) + _np(bufs['cur_steps'])
    assert int(_np(bufs['transitions'])[0]) == per_env.sum()


def test_shard_kernel_with_external_robot_actions(amd, oracle_mod):
    """cn_rollout_step on the 20-human geometry — the value-network rollouts' transition: one bookkept step per call, the
    robot's action from outside — runs the same compact-layout kernel (never the schedule: one step per call).  150 calls,
    every env through several episodes, against the oracle stepped and reset by hand."""
    import torch
    n, steps, K = 12, 150, 64
    cfg = dict(num_humans=20, robot_visible=1, circle_radius=5.0)
    eng = amd.BatchedCrowdSim(num_envs=n, robot_policy=amd.ROBOT_EXTERNAL, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, record_capacity=K)
    o = oracle_mod.CrowdOracle(num_envs=n, robot_policy=0, **cfg)
    o.reset(1000 + np.arange(n))
    ep, cur = np.zeros(n, np.int64), np.zeros(n, np.int64)
    rec_steps, rec_outcome = [[] for _ in range(n)], [[] for _ in range(n)]
    rng = np.random.RandomState(5)
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
    assert ep.min() >= 2
    assert int(_np(bufs['transitions'])[0]) == n * steps
    assert np.array_equal(_np(bufs['ep_count']), ep) and np.array_equal(_np(bufs['cur_steps']), cur)
    got_steps, got_out = _np(bufs['ep_steps']), _np(bufs['ep_outcome'])
    for b in range(n):
        k = len(rec_steps[b])
        assert got_steps[b, :k].tolist() == rec_steps[b] and got_out[b, :k].tolist() == rec_outcome[b]
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9


@pytest.mark.parametrize('async_fill', [False, True])
def test_scenario_cache_of_a_small_seed_set_changes_nothing(amd, oracle_mod, monkeypatch, async_fill):
    """Round 5: the wave generators keep the scenarios of a rollout whose episode seeds come from a small set (seed_mod <=
    4096; step_kernels.h: cached_scenario_wave) — here SEVEN seeds on the reference's 4 m circle, so nearly every auto-reset is
    a cache copy.  Episodes, records, counters and end states are the oracle's (which generates every scenario afresh), and
    equal bit for bit to a run with the cache switched off (CROWDNAV_AMD_SCENARIO_CACHE=0)."""
    import torch
    B, K, launches = 16, 64, [60, 120, 90]
    cfg = dict(num_humans=20, circle_radius=4.0, robot_visible=1)
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    o.reset(1000 + np.arange(B) % 7)
    total, rec, cur = o.rollout_full(sum(launches), 1000, 7, K)

    def run(cache):
        monkeypatch.setenv('CROWDNAV_AMD_SCENARIO_CACHE', '1' if cache else '0')
        eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, flags=amd.FLAG_ASYNC_SCENARIO_FILL if async_fill else 0,
                                  **cfg)
        bufs = eng.rollout_begin(seed_base=1000, seed_mod=7, episode_limit=-1, record_capacity=K)
        for n in launches:
            eng.rollout(n)
            if async_fill:
                eng.sync()  # (drains the fill streams: every env finds its scenarios, the two runs stay comparable)
        eng.sync()
        return eng, bufs

    def check(eng, bufs):
        cnt = _np(bufs['ep_count'])
        assert cnt.min() >= 2 and cnt.sum() > 7 * 6 and (cnt <= rec['count']).all()  # far more episodes than seeds
        if not async_fill:  # (with the asynchronous fill an env may wait for a scenario: behind the oracle, never ahead)
            assert int(_np(bufs['transitions'])[0]) == total
            assert np.array_equal(cnt, rec['count'])
            assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9
        live = np.arange(K)[None, :] < cnt[:, None]
        for key, name in (('ep_outcome', 'outcome'), ('ep_steps', 'steps'), ('ep_time', 'time'), ('ep_danger', 'danger')):
            assert np.array_equal(np.where(live, _np(bufs[key]), 0), np.where(live, rec[name], 0)), key

    eng, bufs = run(True)
    check(eng, bufs)
    eng2, bufs2 = run(False)
    check(eng2, bufs2)
    if not async_fill:  # the same bits with and without the cache
        for k in bufs:
            assert torch.equal(bufs[k], bufs2[k]), k
        assert torch.equal(eng.get_state()[0], eng2.get_state()[0])
