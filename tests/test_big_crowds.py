"""Value networks on crowds beyond the one-tile kernels (more than 8 humans; BASELINE configs[3] holds 20): occupancy maps
(multi_human_rl.py:109-163), LSTM-RL's decreasing-distance ordering (lstm_rl.py:96-103) and the CADRL minimum
(cadrl.py:156-168) stream through the tile in chunks.  Fixtures: the unmodified reference at 12 humans
(oracle/gen_golden_sarl.py: *_h12.npz).  Tolerances as in test_sarl.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden, report_argmax


def _load(net, g):
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


def _select(g, **cfg):
    import crowdnav_amd
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=g['states'].shape[1] - 1,
                                       robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, **cfg)
    return eng


def _check(eng, out, g):
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.array_equal(cpu(eng.sarl_export('reward')), g['rewards'])
    assert np.array_equal(cpu(eng.sarl_export('next_obs')), g['next_obs'])
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= 1 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])


def test_big_crowd_fixtures_hold_twelve_humans_cpu():
    for name in ('sarl_om_h12.npz', 'lstm_rl_om_h12.npz', 'cadrl_h12.npz'):
        g = load_golden(name)
        assert g['states'].shape[1] == 13 and g['inputs'].shape[2] == 12


@pytest.mark.gpu
@pytest.mark.parametrize('kernels', ['by-size', 'register-resident'])
def test_om_sarl_twelve_humans_vs_reference(kernels, monkeypatch):
    """... on the chunked LDS kernel this batch size selects, and on sarl_reg_chunk_kernel<4, true> (forced: 3 chunks of 4)."""
    from crowdnav_amd.compat.sarl import ValueNetwork
    if kernels == 'register-resident':
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')
    g = load_golden('sarl_om_h12.npz')
    eng = _select(g, with_om=True)
    eng.sarl_set_weights(_load(ValueNetwork(61, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4),
                               g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    assert np.abs(eng.sarl_export('om').cpu().numpy() - g['inputs'][:, 0, :, 13:]).max() <= 5e-6
    _check(eng, out, g)


@pytest.mark.gpu
@pytest.mark.parametrize('kernels', ['by-size', 'register-resident'])
def test_lstm_rl_twelve_humans_vs_reference(kernels, monkeypatch):
    """... on the LDS kernel this batch size selects, and on lstm_reg_kernel (forced): its v_exp_f32 / v_rcp_f32 sigmoid and
    tanh through 12 LSTM steps stay within 1e-6 of the unmodified reference's network outputs."""
    from crowdnav_amd.compat.lstm_rl import ValueNetwork1
    if kernels == 'register-resident':
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')
    g = load_golden('lstm_rl_om_h12.npz')
    eng = _select(g, model='lstm_rl', with_om=True, mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_load(ValueNetwork1(61, 6, [150, 100, 100, 1], 50), g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check(eng, out, g)
    # the replay-memory state of the train phase: humans by decreasing distance (stable), 12 > the register permutation
    got = eng.sarl_transform(sort_humans=True).cpu().numpy()
    s = g['states']
    d = np.linalg.norm(s[:, 1:, :2] - s[:, :1, :2], axis=2)
    order = np.argsort(-d, axis=1, kind='stable')
    env_order = eng.sarl_transform(sort_humans=False).cpu().numpy()
    for b in range(len(s)):
        assert np.abs(got[b][:, :13] - env_order[b][order[b]][:, :13]).max() == 0.0
        assert (order[b] != np.arange(12)).any()


@pytest.mark.gpu
def test_lstm_rl_pairwise_twelve_humans_vs_torch():
    """lstm_rl.ValueNetwork2 (interaction module) at 12 humans against the torch module on the device's own inputs."""
    from crowdnav_amd.compat.lstm_rl import ValueNetwork2
    g = load_golden('lstm_rl_om_h12.npz')
    torch.manual_seed(5)
    net = ValueNetwork2(61, 6, [150, 100, 100, 50], [150, 100, 100, 1], 50)
    eng = _select(g, model='lstm_rl', with_om=True, mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1),
                  interaction_dims=(150, 100, 100, 50))
    eng.sarl_set_weights(net.state_dict())
    eng.sarl_select()
    eng.sync()
    X = eng.sarl_export('X').cpu()
    n = X.shape[0]
    with torch.no_grad():
        want = net(X.reshape(n * 81, 12, 61)).reshape(n, 81).numpy()
    assert np.abs(eng.sarl_export('V').cpu().numpy() - want).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('kernels', ['by-size', 'register-resident'])
def test_cadrl_twelve_humans_vs_reference(kernels, monkeypatch):
    from crowdnav_amd.compat.cadrl import ValueNetwork
    if kernels == 'register-resident':
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')  # cadrl_reg_kernel<4>, 3 chunks
    g = load_golden('cadrl_h12.npz')
    eng = _select(g, model='cadrl', mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_load(ValueNetwork(13, [150, 100, 100, 1]), g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check(eng, out, g)


# ---- unicycle robots for the other two value networks (cadrl.py:90-98, 119-125, 207-211 apply to every policy) --------
def _unicycle_select(g, **cfg):
    import crowdnav_amd
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1,
                                       robot_kinematics=crowdnav_amd.UNICYCLE)
    eng.set_state(g['states'], g['gtime'])
    eng.set_theta(g['theta'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, **cfg)
    return eng


def _check_unicycle(eng, out, g):
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.abs(cpu(eng.sarl_export('reward')) - g['rewards']).max() <= 1e-12  # device cos / sin in the swept distance
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= len(g['states']) // 4 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])
    assert np.array_equal(cpu(out['action'])[clear], g['action'][clear])


@pytest.mark.gpu
def test_cadrl_unicycle_vs_reference():
    from crowdnav_amd.compat.cadrl import ValueNetwork
    g = load_golden('cadrl_unicycle.npz')
    eng = _unicycle_select(g, model='cadrl', mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_load(ValueNetwork(13, [150, 100, 100, 1]), g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check_unicycle(eng, out, g)


@pytest.mark.gpu
def test_lstm_rl_unicycle_vs_reference():
    from crowdnav_amd.compat.lstm_rl import ValueNetwork1
    g = load_golden('lstm_rl_unicycle.npz')
    eng = _unicycle_select(g, model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_load(ValueNetwork1(13, 6, [150, 100, 100, 1], 50), g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    _check_unicycle(eng, out, g)


def test_unicycle_value_network_policies_configure_cpu():
    """compat: CADRL / LSTM-RL no longer refuse [action_space] kinematics = unicycle."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    for name in ('cadrl', 'lstm_rl'):
        policy = c.policy_factory[name]()
        policy.configure(default_policy_config({('action_space', 'kinematics'): 'unicycle'}))
        assert policy.kinematics == 'unicycle'
