"""Exact distance ties in crowds of more than 10 agents: RVO2 visits an agent's candidate neighbours in kd-tree order
(/root/reference crowd_sim/envs/policy/orca.py:99-128 with max_neighbors = 10; oracle/rvo2_oracle.cpp:162-256), so which of
two EQUALLY distant candidates comes first in the neighbour list — or falls off its end — is decided by the tree, and the tree
partitions a permutation that persists with the simulator.  Random scenes never tie (0 ties in 322 560 agent-steps of the
20-human fixtures); these scenes are lattices of float32-exact coordinates with hundreds of ties per env.  The kernel
(crowdnav_amd/csrc/kd_order.h) must reproduce the oracle's ORCA velocities bit for bit: with fresh simulators, and with
simulators whose permutations carry the history of earlier steps."""
import numpy as np
import pytest

from test_kd_order_emulation import LEAF, device_neighbours, partition_bits, shared_tree

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope='module')
def amd():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need a MI355X'
    import crowdnav_amd
    return crowdnav_amd


def _np(t):
    return t.detach().cpu().numpy()


def lattice_state(rng, B, A, step=0.5, span=3.0, speed=True):
    """[B, A, 8] states on a lattice: positions multiples of `step` (exact in float32), pairwise distinct and at least 0.75
    apart (no deep overlaps), velocities multiples of 0.25, goals anywhere."""
    m = int(round(span / step))
    st = np.zeros((B, A, 8))
    for b in range(B):
        pts = []
        while len(pts) < A:
            p = (rng.randint(-m, m + 1) * step, rng.randint(-m, m + 1) * step)
            if all((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 >= 0.75 ** 2 for q in pts):
                pts.append(p)
        st[b, :, 0:2] = np.array(pts)
        if speed:
            st[b, :, 2:4] = rng.randint(-3, 4, size=(A, 2)) * 0.25
        st[b, :, 4:6] = rng.uniform(-4, 4, size=(A, 2))
        st[b, :, 6] = 0.3
        st[b, :, 7] = 1.0
    return st


def tie_statistics(state, robot_visible):
    """(agents whose 10 nearest contain an exact tie or tie with the 11th, agents whose neighbour list under RVO2's order
    differs from the index-order tie-break) for fresh simulators — computed with the CPU emulation of the device's rules"""
    ties = differs = 0
    for env in state:
        A = len(env)
        pos = [(f32(x), f32(y)) for x, y in env[:, 0:2]]
        for q in range(A):
            members = list(range(A)) if (q == 0 or robot_visible) else list(range(1, A))
            if len(members) <= LEAF:
                continue
            others = [a for a in range(1, A) if a != q] + ([0] if (q > 0 and robot_visible) else [])
            row = [q] + others
            loc = {a: i for i, a in enumerate(row)}
            lp = [pos[a] for a in row]
            nodes = shared_tree(lp, list(range(len(row))))
            lrow = list(range(len(row)))
            for b, e, nl, left in nodes:
                partition_bits(lrow, b, e, nl, left)
            kd = [o for _, o in device_neighbours(lp, lrow, nodes, 0, 10, 100.0)]
            cands = []
            for c, a in enumerate(others):
                dx, dy = f32(pos[q][0] - pos[a][0]), f32(pos[q][1] - pos[a][1])
                cands.append((f32(f32(dx * dx) + f32(dy * dy)), c, loc[a]))
            cands.sort()
            d = [x[0] for x in cands[:11]]
            ties += len(d) != len(set(d))
            differs += [x[2] for x in cands[:10]] != kd
    return ties, differs


CASES = [(12, 1), (20, 1), (20, 0), (10, 1), (11, 0), (33, 1)]


@pytest.mark.parametrize('H,visible', CASES)
def test_forced_ties_with_fresh_simulators(amd, oracle_mod, H, visible):
    """cn_orca on lattice scenes, every simulator freshly built: velocities of every agent bit-identical to the oracle's."""
    rng = np.random.RandomState(100 + H + visible)
    B = 24
    cfg = dict(num_humans=H, robot_visible=visible)
    state = lattice_state(rng, B, H + 1, span=3.0 if H <= 20 else 5.0)
    ties, differs = tie_statistics(state[:6], visible)
    assert ties > 20 and differs > 5, (ties, differs)  # the scenes are tie-heavy AND tell the two orders apart
    eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, **cfg)
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    for obj in (eng, o):
        obj.set_state(state, np.zeros(B))
    got, want = _np(eng.orca()), o.orca()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the same scene through cn_step (update and lookahead), fresh simulators again
    eng.drop_sims()
    o.drop_sims()
    for update in (False, True):
        g, w = eng.step(None, update=update, want_obs=False), o.step(None, update=update)
        assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32)), update
        for k in ('reward', 'done', 'info', 'dmin', 'action'):
            assert np.array_equal(_np(g[k]), w[k]), (k, update)
    assert np.array_equal(_np(eng.get_state()[0]), o.get_state()[0])


@pytest.mark.parametrize('H,visible', [(12, 1), (20, 1), (20, 0)])
def test_ties_are_decided_by_the_history_of_the_permutation(amd, oracle_mod, H, visible):
    """The kd-tree partitions its permutation in place and keeps it: free-running steps from random scenes scramble it, then
    the agents are teleported onto a lattice WITHOUT rebuilding the simulators (cn_set_state keeps them, like the oracle's
    set_state), four times over; after a reset (new humans, the robot's simulator lives on) once more.  Every output of
    every step equals the oracle's bit for bit."""
    rng = np.random.RandomState(7 * H + visible)
    B = 16
    cfg = dict(num_humans=H, robot_visible=visible, circle_radius=6.0)
    eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, **cfg)
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)

    def both_step(n):
        for _ in range(n):
            g, w = eng.step(None, update=True, want_obs=False), o.step(None, update=True)
            assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32))
            for k in ('reward', 'done', 'info', 'dmin', 'action'):
                assert np.array_equal(_np(g[k]), w[k]), k

    o.reset(300 + np.arange(B))
    eng.drop_sims()
    eng.set_state(o.get_state()[0], np.zeros(B))
    for rnd in range(4):
        both_step(6)  # the crowd moves: agents cross split planes, the permutations are re-partitioned
        state = lattice_state(rng, B, H + 1, step=0.5 if rnd % 2 else 0.25)
        for obj in (eng, o):
            obj.set_state(state, np.zeros(B))
        both_step(2)  # first step: ties everywhere; second: off the lattice again
    seeds = 900 + np.arange(B)
    o.reset(seeds)
    eng.reset(seeds)  # new Human objects: their simulators are rebuilt; the robot's is kept
    state = lattice_state(rng, B, H + 1)
    for obj in (eng, o):
        obj.set_state(state, np.zeros(B))
    both_step(3)
    assert np.array_equal(_np(eng.get_state()[0]), o.get_state()[0])


@pytest.mark.parametrize('H,visible,envs_per_wave,waves', [(12, 1, 2, 1), (20, 1, 3, 1), (20, 0, 2, 1), (12, 1, 4, 2), (20, 1, 1, 2)])
def test_ties_with_several_envs_per_workgroup(amd, oracle_mod, H, visible, envs_per_wave, waves):
    """Large batches pack several envs into a workgroup (more than 4096 envs of 10+ humans; CROWDNAV_AMD_ENVS_PER_WAVE forces
    it here) and a workgroup may have more than one wave (CROWDNAV_AMD_WAVES_PER_BLOCK): the kd-trees are then built by the
    cooperative builder (LDS min / max per env, kd_build_trees) instead of the wave-uniform one, and the fallback by the
    shuffle-round program.  Lattice scenes, history-carrying simulators: bit for bit vs the oracle."""
    import os
    rng = np.random.RandomState(31 * H + 7 * visible + envs_per_wave)
    B = 13  # a partial last workgroup
    cfg = dict(num_humans=H, robot_visible=visible, circle_radius=6.0)
    old = {k: os.environ.get(k) for k in ('CROWDNAV_AMD_ENVS_PER_WAVE', 'CROWDNAV_AMD_WAVES_PER_BLOCK')}
    os.environ['CROWDNAV_AMD_ENVS_PER_WAVE'] = str(envs_per_wave)
    os.environ['CROWDNAV_AMD_WAVES_PER_BLOCK'] = str(waves)
    try:
        eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, **cfg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    o.reset(500 + np.arange(B))
    eng.drop_sims()
    eng.set_state(o.get_state()[0], np.zeros(B))
    for rnd in range(3):
        for t in range(4 if rnd else 0):
            g, w = eng.step(None, update=True, want_obs=False), o.step(None, update=True)
            assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32)), (rnd, t)
        state = lattice_state(rng, B, H + 1)
        for obj in (eng, o):
            obj.set_state(state, np.zeros(B))
        for t in range(2):
            g, w = eng.step(None, update=True, want_obs=False), o.step(None, update=True)
            assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32)), (rnd, t)
            for k in ('reward', 'done', 'info', 'dmin'):
                assert np.array_equal(_np(g[k]), w[k]), k
    # the rollout kernel on the same geometry: a few steps, then a lattice teleport
    eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=4)
    eng.rollout(30)
    o.reset(1000 + np.arange(B))
    o.rollout(30, 1000, 500, 4, np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64))
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9
    state = lattice_state(rng, B, H + 1)
    for obj in (eng, o):
        obj.set_state(state, np.zeros(B))
    g, w = eng.step(None, update=True, want_obs=False), o.step(None, update=True)
    assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32))


def test_rollout_keeps_simulators_across_steps_and_renews_them_at_resets(amd, oracle_mod):
    """cn_rollout at 12 humans carries the permutations in LDS from step to step and rebuilds the humans' simulators at every
    auto-reset; afterwards the agents are teleported onto a lattice: the tie order of the very next step depends on what the
    rollout left behind."""
    B, H, steps = 12, 12, 150
    cfg = dict(num_humans=H, robot_visible=1, circle_radius=6.0)
    eng = amd.BatchedCrowdSim(num_envs=B, robot_policy=amd.ROBOT_ORCA, **cfg)
    eng.rollout_begin(seed_base=1000, seed_mod=500, episode_limit=-1, record_capacity=8)
    for n in (40, 1, 60, 49):
        eng.rollout(n)
    o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    o.reset(1000 + np.arange(B))
    ep_index, cur_steps, cur_ret = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64)
    _, rec = o.rollout(steps, 1000, 500, 8, ep_index, cur_steps, cur_ret)
    assert rec['count'].min() >= 1
    assert np.abs(_np(eng.get_state()[0]) - o.get_state()[0]).max() <= 1e-9
    state = lattice_state(np.random.RandomState(5), B, H + 1)
    for obj in (eng, o):
        obj.set_state(state, np.zeros(B))
    for _ in range(2):
        g, w = eng.step(None, update=True, want_obs=False), o.step(None, update=True)
        assert np.array_equal(_np(g['orca_vel']).view(np.uint32), w['orca_vel'].view(np.uint32))
