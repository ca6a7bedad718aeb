"""The CPU oracle (oracle/crowd_oracle.cpp, a batched restatement of the reference hot path) against the
fixtures that the UNMODIFIED reference Python produced (oracle/gen_golden.py).  This is what pins the oracle;
the GPU parity tests then compare the HIP path with the oracle and with the same fixtures."""
import numpy as np
import pytest

from conftest import TRAJ_FIXTURES, episodes_of, flat_steps, load_golden


@pytest.mark.parametrize('name', sorted(TRAJ_FIXTURES))
def test_teacher_forced_steps_bit_exact(oracle_mod, name):
    g = load_golden(name)
    before, after, gtime = flat_steps(g)
    n = len(before)
    # a fresh policy object per case in the random-attribute fixture, and fresh human policies per reset:
    # a fresh oracle (no captured radii) per teacher-forced batch reproduces that for every step
    o = oracle_mod.CrowdOracle(num_envs=n, robot_policy=1, **TRAJ_FIXTURES[name])
    o.set_state(before, gtime)
    out = o.step(None, update=True)
    state, gt = o.get_state()
    assert np.array_equal(out['reward'], g['rewards'])
    assert np.array_equal(out['done'], g['dones'])
    assert np.array_equal(out['info'], g['infos'])
    assert np.array_equal(out['action'], g['actions'])
    danger = g['infos'] == 1
    assert np.array_equal(out['dmin'][danger], g['dmins'][danger])
    assert np.array_equal(state, after)
    assert np.array_equal(gt, gtime + 0.25)


@pytest.mark.parametrize('name', ['traj_invisible_h5.npz', 'traj_visible_h5.npz', 'traj_visible_h20.npz'])
def test_free_running_episodes_bit_exact(oracle_mod, name):
    g = load_golden(name)
    for e in episodes_of(g):
        o = oracle_mod.CrowdOracle(num_envs=1, robot_policy=1, **TRAJ_FIXTURES[name])
        o.set_state(e['states'][:1], np.zeros(1))
        for t in range(len(e['actions'])):
            out = o.step(None, update=True)
            assert out['reward'][0] == e['rewards'][t] and out['done'][0] == e['dones'][t]
            assert out['info'][0] == e['infos'][t]
            assert np.array_equal(o.get_state()[0][0], e['states'][t + 1])
        assert out['done'][0] == 1


def test_lookahead_does_not_mutate(oracle_mod):
    g = load_golden('traj_visible_h5.npz')
    before, _, gtime = flat_steps(g)
    o = oracle_mod.CrowdOracle(num_envs=len(before), robot_policy=1, num_humans=5, robot_visible=1)
    o.set_state(before, gtime)
    out = o.step(None, update=False)
    state, gt = o.get_state()
    assert np.array_equal(state, before) and np.array_equal(gt, gtime)
    assert np.array_equal(out['reward'], g['rewards'])


def test_mt19937_matches_numpy_stream(oracle_mod):
    g = load_golden('resets.npz')
    for s in (0, 1000, 2000, 4294965295):
        assert np.array_equal(oracle_mod.mt_random(s, 700), g['mt_random_%d' % s])


RESET_SPECS = {
    'test_h5': dict(num_humans=5), 'train_h5': dict(num_humans=5), 'val_h5': dict(num_humans=5),
    'test_h5_random': dict(num_humans=5, randomize_attributes=1),
    'test_h5_square': dict(num_humans=5, scenario_rule=1),
    'test_h10': dict(num_humans=10), 'test_h20': dict(num_humans=20),
}


@pytest.mark.parametrize('name', sorted(RESET_SPECS))
def test_reset_matches_reference_generator(oracle_mod, name):
    """Scenario generation vs the reference's own generator (numpy cos/sin vs libm: <= 1e-12)."""
    g = load_golden('resets.npz')
    want, seeds = g[name + '_states'], g[name + '_seeds']
    o = oracle_mod.CrowdOracle(num_envs=len(seeds), **RESET_SPECS[name])
    o.reset(seeds)
    got, gt = o.get_state()
    assert np.all(gt == 0.0)
    assert np.abs(got - want).max() <= 1e-12
    # attributes and the robot row involve no trigonometry: exact
    assert np.array_equal(got[:, :, 6:], want[:, :, 6:]) and np.array_equal(got[:, 0], want[:, 0])


def test_rollout_reproduces_500_case_anchor(oracle_mod):
    """Explorer bookkeeping over the 500 test cases: 213 ReachGoal / 284 Collision / 3 Timeout, 15 190 steps
    (SURVEY.md Appendix D), per-case outcome, length and discounted return."""
    g = load_golden('outcomes_500.npz')
    for tag, vis in (('invisible', 0), ('visible', 1)):
        B = 500
        o = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, robot_visible=vis)
        o.reset(1000 + np.arange(B))
        ep_index = np.zeros(B, np.int32)
        cur_steps = np.zeros(B, np.int32)
        cur_ret = np.zeros(B, np.float64)
        _, rec = o.rollout(100, 1000, 500, 4, ep_index, cur_steps, cur_ret)
        assert np.all(rec['count'] >= 1)
        assert np.array_equal(rec['outcome'][:, 0], g[tag + '_info'])
        assert np.array_equal(rec['steps'][:, 0], g[tag + '_steps'])
        assert np.allclose(rec['ret'][:, 0], g[tag + '_return'], rtol=0, atol=1e-12)
    assert np.bincount(g['invisible_info'], minlength=5).tolist() == [0, 0, 213, 284, 3]
    assert int(g['invisible_steps'].sum()) == 15190
    assert np.bincount(g['visible_info'], minlength=5).tolist() == [0, 0, 500, 0, 0]
    assert int(g['visible_steps'].sum()) == 20037


def test_unicycle_robot_steps_vs_reference(oracle_mod):
    """ActionRot kinematics (crowd_sim.py:339-341, agent.py:115-135): the oracle's unicycle robot against transitions
    of the unmodified reference driven by a unicycle SARL policy (sarl_unicycle.npz).  numpy's cos/sin vs libm: 1e-12."""
    g = load_golden('sarl_unicycle.npz')
    n = len(g['states'])
    o = oracle_mod.CrowdOracle(num_envs=n, robot_policy=0, robot_visible=1, robot_kinematics=1)
    o.set_state(g['states'], g['gtime'])
    o.set_theta(g['theta'])
    out = o.step(g['action'], update=True)
    assert np.array_equal(out['done'], g['step_done']) and np.array_equal(out['info'], g['step_info'])
    assert np.abs(out['reward'] - g['step_reward']).max() <= 1e-12
    state, _ = o.get_state()
    assert np.abs(state - g['next_states']).max() <= 1e-12
    assert np.abs(o.get_theta() - g['next_theta']).max() <= 1e-12
    assert np.array_equal(state[:, 1:], g['next_states'][:, 1:])  # humans do not depend on the robot's kinematics


SURVEY_KAT_CASE0 = [  # test case 0, ORCA robot, robot invisible — an INDEPENDENT float32 restatement
    ((-0.051465511322021484, 0.45335352420806885),
     '7424d63eb76eca3e 54b22d3f3cac3d3c 8e4c2cbf7e45a4bd cef9a9be10f4113f 100e253ff8dc75be'),
    ((0.08838461339473724, 0.512067437171936),
     '937cd13e6ac1e13e d51f243f9bd3923c ea9c24bf57a289bd b972adbe7e250a3f e88f313f5b7393be'),
    ((0.07480747997760773, 0.5604285597801208),
     'ff69c03e5c7df23e c2471a3fb26bff3c efb71dbfd3e653bd f3b4a1be45cd053f 8fbd533fe931aabe'),
]


def test_survey_known_answer_velocities(oracle_mod):
    """Two independent restatements of RVO2 (the survey's throw-away Python float32 one and this C++ oracle) agree bit
    for bit on the first three steps of test case 0 — robot action and every human's ORCA velocity as float32 hex."""
    g = load_golden('resets.npz')
    assert int(g['test_h5_seeds'][0]) == 1000
    o = oracle_mod.CrowdOracle(num_envs=1, num_humans=5, robot_policy=1, robot_visible=0)
    o.set_state(g['test_h5_states'][:1], np.zeros(1))
    for action, hexes in SURVEY_KAT_CASE0:
        out = o.step(None, update=True)
        assert tuple(out['action'][0]) == action and out['reward'][0] == 0.0
        assert ' '.join(v.tobytes().hex() for v in out['orca_vel'][0][1:]) == hexes
    assert tuple(o.get_state()[0][0][1][:2]) != (0.0, 0.0)


def test_committed_fixtures_regenerate_bit_for_bit_from_the_unmodified_reference(tmp_path):
    """the fixtures are not a one-off.  With the reference on this machine (/root/reference or oracle/_ref),
    oracle/gen_golden.py — the UNMODIFIED reference Python on the float32 rvo2 restatement — rewrites traj_visible_h5.npz
    and outcomes_500.npz into a scratch directory in a subprocess; every array must equal the committed file's bit for bit
    and the key sets must match (no stale schema)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    import ref_harness
    if not ref_harness.available():
        pytest.skip('the reference is not on this machine')
    code = ("import sys; sys.path.insert(0, %r); import gen_golden as g; g.OUT = %r; "
            "g.trajectories('traj_visible_h5.npz', list(range(10)), robot_visible=True); g.outcomes_500()"
            % (os.path.join(root, 'oracle'), str(tmp_path)))
    subprocess.run([sys.executable, '-c', code], check=True, capture_output=True, timeout=300,
                   env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    for name in ('traj_visible_h5.npz', 'outcomes_500.npz'):
        new, old = dict(np.load(str(tmp_path / name))), dict(load_golden(name))
        assert sorted(new) == sorted(old)
        for k in new:
            assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape and new[k].tobytes() == old[k].tobytes(), (name, k)
    info = np.load(str(tmp_path / 'outcomes_500.npz'))['invisible_info']
    assert np.bincount(info, minlength=5).tolist() == [0, 0, 213, 284, 3]   # the paper's ORCA row (0.43 / 0.57)
