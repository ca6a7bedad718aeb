"""[action_space] query_env = false (multi_human_rl.py:37-42, 65-88): the value networks decide on a constant-velocity
model of the humans and MultiHumanRL.compute_reward instead of the env's onestep_lookahead.  Fixtures from the unmodified
reference (oracle/gen_golden_sarl.py: sarl_noquery_*.npz, lstm_rl_noquery_om.npz).  Rewards are float64 end-point
arithmetic (bit-identical; the unicycle end point goes through device cos/sin: 1e-12); features 5e-6, network output and
values 1e-6, arg-max equal wherever the reference's top two values are apart."""
import numpy as np
import pytest
import torch

from conftest import load_golden, report_argmax


def _load(net, g):
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


def _check_select(eng, out, g, reward_tol):
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    n = len(g['states'])
    got_r = cpu(eng.sarl_export('reward'))
    if reward_tol == 0:
        assert np.array_equal(got_r, g['rewards'])
    else:
        assert np.abs(got_r - g['rewards']).max() <= reward_tol
    assert np.abs(cpu(eng.sarl_export('next_obs')) - g['next_obs']).max() == 0.0
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 4
    assert np.array_equal(cpu(out['best'])[clear], g['best'][clear])
    assert np.array_equal(cpu(out['action'])[clear], g['action'][clear])


def test_fixture_rewards_are_compute_reward_not_the_lookahead_cpu():
    """The fixture really exercises the other branch: its rewards equal a numpy restatement of compute_reward on the
    constant-velocity next states and differ from what the env's lookahead fixtures hold."""
    g = load_golden('sarl_noquery_om.npz')
    s, acts = g['states'], g['action_space']
    for d in (0, 5, len(s) - 1):
        robot, humans = s[d, 0], s[d, 1:]
        nh = humans[:, :2] + humans[:, 2:4] * 0.25
        assert np.array_equal(nh, g['next_obs'][d][:, :2]) and np.array_equal(humans[:, 2:4], g['next_obs'][d][:, 2:4])
        for a in (0, 7, 40, 80):
            nav = robot[:2] + acts[a] * 0.25
            dmin, hit = np.inf, False
            for h, p in zip(humans, nh):
                dist = np.linalg.norm((nav[0] - p[0], nav[1] - p[1])) - robot[6] - h[6]
                if dist < 0:
                    hit = True
                    break
                dmin = min(dmin, dist)
            goal = np.linalg.norm((nav[0] - robot[4], nav[1] - robot[5])) < robot[6]
            want = -0.25 if hit else (1 if goal else ((dmin - 0.2) * 0.5 * 0.25 if dmin < 0.2 else 0))
            assert g['rewards'][d][a] == want


@pytest.mark.gpu
def test_sarl_without_query_env_vs_reference():
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork
    g = load_golden('sarl_noquery_om.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, with_om=True, query_env=False)
    net = _load(ValueNetwork(61, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4), g)
    eng.sarl_set_weights(net.state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check_select(eng, out, g, 0)


@pytest.mark.gpu
def test_lstm_rl_without_query_env_sorts_the_humans_vs_reference():
    """LSTM-RL: the joint state LstmRL.predict sorted by decreasing distance is what gets propagated, so the LSTM sees
    the humans in that order (with query_env the lookahead returns env order)."""
    import crowdnav_amd
    from crowdnav_amd.compat.lstm_rl import ValueNetwork1
    g = load_golden('lstm_rl_noquery_om.npz')
    n = len(g['states'])
    # the fixture does re-order: some decision's nearest-last order differs from env order
    d0 = np.linalg.norm(g['states'][:, 1:, :2] - g['states'][:, :1, :2], axis=2)
    assert (np.argsort(-d0, axis=1, kind='stable') != np.arange(5)).any()
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, model='lstm_rl', with_om=True, mlp1_dims=(50, 1),
                       mlp3_dims=(150, 100, 100, 1), query_env=False)
    eng.sarl_set_weights(_load(ValueNetwork1(61, 6, [150, 100, 100, 1], 50), g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check_select(eng, out, g, 0)


@pytest.mark.gpu
def test_unicycle_sarl_without_query_env_vs_reference():
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork
    g = load_golden('sarl_noquery_unicycle.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                       robot_visible=0, robot_kinematics=crowdnav_amd.UNICYCLE)
    eng.set_state(g['states'], g['gtime'])
    eng.set_theta(g['theta'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, query_env=False)
    net = _load(ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4), g)
    eng.sarl_set_weights(net.state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check_select(eng, out, g, 1e-12)


@pytest.mark.gpu
def test_gym_surface_sarl_policy_without_query_env_follows_reference_episode():
    """policy.config with query_env = false on the reference's surface: configure no longer refuses it, robot.act picks the
    reference's actions step for step."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    g = load_golden('sarl_noquery_om.npz')
    cfg = c.default_env_config({('robot', 'visible'): 'true'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    policy = c.policy_factory['sarl']()
    policy.configure(default_policy_config({('sarl', 'with_om'): 'true', ('action_space', 'query_env'): 'false'}))
    assert policy.query_env is False
    _load(policy.get_model(), g)
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_phase('test')
    policy.set_device(torch.device('cpu'))
    policy.set_env(env)
    ob = env.reset('test', 17)
    env._eng.set_state(g['states'][:1], np.zeros(1))  # the reference's exact initial state
    env._pull()
    ob = [h.get_observable_state() for h in env.humans]
    for d in range(12):
        assert np.array_equal(np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans]), g['states'][d][1:, :4])
        action = robot.act(ob)
        assert (action.vx, action.vy) == tuple(g['action'][d])
        assert np.abs(np.array(policy.action_values) - g['values'][d]).max() <= 1e-6
        ob, reward, done, info = env.step(action)
        assert reward == g['step_reward'][d] and done == bool(g['step_done'][d])
        if done:
            break


def test_cadrl_has_no_constant_velocity_branch_abi():
    """CADRL.predict always queries the env (cadrl.py:150): the flag is refused for that model before any device work."""
    import ctypes as C
    from crowdnav_amd import _lib
    lib = _lib.load()
    cfg = _lib.CnSarlConfig(n_actions=81, model=1, constant_velocity_model=1)
    assert hasattr(cfg, 'constant_velocity_model') and C.sizeof(cfg) == 112
    assert lib.cn_sarl_configure(None, C.byref(cfg), None) == _lib.CN_ERR_INVALID  # NULL engine first
