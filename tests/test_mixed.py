"""The `mixed` scenario rule (crowd_sim/envs/crowd_sim.py:103-151): a per-episode number of humans — static obstacles or
moving humans — against fixtures from the UNMODIFIED reference (oracle/gen_golden_mixed.py).  The engine and the oracle
keep 5 human slots; humans the rule left out are parked far away, behind the present ones."""
import numpy as np
import pytest

from conftest import load_golden, report_argmax

FIXTURES = ['mixed.npz', 'mixed_invisible.npz']
SLOTS = 5


def padded(rows):
    """Reference agent rows [1 + 5, 8] (NaN where the episode has no human) -> engine state with parked humans."""
    out = rows.copy()
    for i in range(1, 1 + SLOTS):
        if np.isnan(out[i, 0]):
            out[i] = [1.0e6 + 100.0 * i, 1.0e6, 0.0, 0.0, 1.0e6 + 100.0 * i, 1.0e6, 0.3, 1.0]
    return out


def episodes(g):
    s0 = a0 = 0
    for T, n in zip(g['ep_steps'].tolist(), g['ep_count'].tolist()):
        yield dict(states=g['ep_states'][s0:s0 + T + 1], actions=g['ep_actions'][a0:a0 + T],
                   rewards=g['ep_rewards'][a0:a0 + T], dones=g['ep_dones'][a0:a0 + T],
                   infos=g['ep_infos'][a0:a0 + T], count=n)
        s0 += T + 1
        a0 += T


def check_reset(got_state, got_count, g):
    want, count = g['reset_states'], g['reset_count']
    assert np.array_equal(got_count, count)
    for b in range(len(count)):
        n = int(count[b])
        assert np.abs(got_state[b, :1 + n] - want[b, :1 + n]).max() <= 1e-12     # cos / sin: numpy vs libm / device
        assert np.array_equal(got_state[b, :1 + n, 6:], want[b, :1 + n, 6:])     # radius, v_pref
        assert np.all(got_state[b, 1 + n:, 0] >= 5.0e5) and np.all(got_state[b, 1 + n:, 2:4] == 0.0)
        assert np.array_equal(got_state[b, 1 + n:, :2], got_state[b, 1 + n:, 4:6])  # parked: goal = position
    # the placeholder human of a static scenario that drew zero obstacles (crowd_sim.py:121-124)
    for b in np.flatnonzero((g['reset_human_num'] == 0) & (count == 1)):
        assert tuple(got_state[b, 1, [0, 1, 4, 5]]) == (0.0, -10.0, 0.0, -10.0)


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize('name', FIXTURES)
def test_oracle_mixed_reset_matches_reference_generator(oracle_mod, name):
    g = load_golden(name)
    n = len(g['reset_count'])
    o = oracle_mod.CrowdOracle(num_envs=n, num_humans=SLOTS, scenario_rule=2, robot_visible=int(g['robot_visible']))
    draws = o.reset(1000 + np.arange(n))
    check_reset(o.get_state()[0], o.human_count(), g)
    # the stream position the scenario leaves behind: np.random.seed(seed) + that many random() calls reproduce the
    # first two draws of the rule (static? how many?)
    for b in range(0, n, 17):
        rs = np.random.RandomState(1000 + b)
        static, prob = rs.random_sample() < 0.2, rs.random_sample()
        assert draws[b] >= 2 and (static or g['reset_human_num'][b] >= 1) and 0.0 <= prob < 1.0


@pytest.mark.parametrize('name', FIXTURES)
def test_oracle_mixed_episodes_bit_exact(oracle_mod, name):
    g = load_golden(name)
    for e in episodes(g):
        o = oracle_mod.CrowdOracle(num_envs=1, num_humans=SLOTS, robot_policy=1, scenario_rule=2,
                                   robot_visible=int(g['robot_visible']))
        o.set_state(padded(e['states'][0])[None], np.zeros(1))
        n = e['count']
        for t in range(len(e['actions'])):
            out = o.step(None, update=True)
            assert out['reward'][0] == e['rewards'][t] and out['done'][0] == e['dones'][t]
            assert out['info'][0] == e['infos'][t] and np.array_equal(out['action'][0], e['actions'][t])
            state = o.get_state()[0][0]
            assert np.array_equal(state[:1 + n], e['states'][t + 1][:1 + n])
            assert np.array_equal(state[1 + n:], padded(e['states'][0])[1 + n:])  # parked humans never move
        assert out['done'][0] == 1


def test_mixed_needs_five_to_eight_slots_abi():
    """cn_create validates the rule before it looks for a device."""
    import ctypes as C
    from crowdnav_amd import _lib
    from crowdnav_amd.engine import default_config
    lib = _lib.load()
    for humans in (4, 9):
        cfg = _lib.CnConfig(**default_config(num_envs=4, num_humans=humans, scenario_rule=_lib.MIXED))
        handle = C.c_void_p()
        assert lib.cn_create(C.byref(cfg), C.byref(handle)) == _lib.CN_ERR_UNSUPPORTED
        assert b'mixed' in lib.cn_last_error()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_engine_mixed_reset_matches_reference_generator(name):
    import crowdnav_amd
    g = load_golden(name)
    n = len(g['reset_count'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=SLOTS, scenario_rule=crowdnav_amd.MIXED,
                                       robot_visible=int(g['robot_visible']))
    eng.reset(1000 + np.arange(n))
    check_reset(eng.get_state()[0].cpu().numpy(), eng.human_count().cpu().numpy(), g)


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_engine_mixed_episodes_bit_exact(name):
    """All fixture episodes side by side (one env each), teacher-forced start, ORCA robot on device."""
    import crowdnav_amd
    g = load_golden(name)
    eps = list(episodes(g))
    B = len(eps)
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=SLOTS, robot_policy=crowdnav_amd.ROBOT_ORCA,
                                       scenario_rule=crowdnav_amd.MIXED, robot_visible=int(g['robot_visible']))
    start = np.stack([padded(e['states'][0]) for e in eps])
    eng.set_state(start, np.zeros(B))
    assert np.array_equal(eng.human_count().cpu().numpy(), [e['count'] for e in eps])
    for t in range(max(len(e['actions']) for e in eps)):
        out = eng.step(None, update=True, want_obs=False)
        state = eng.get_state()[0].cpu().numpy()
        rew, done, info, act = (out[k].cpu().numpy() for k in ('reward', 'done', 'info', 'action'))
        for b, e in enumerate(eps):
            if t >= len(e['actions']):
                continue
            n = e['count']
            assert rew[b] == e['rewards'][t] and done[b] == e['dones'][t] and info[b] == e['infos'][t], (b, t)
            assert np.array_equal(act[b], e['actions'][t])
            assert np.array_equal(state[b, :1 + n], e['states'][t + 1][:1 + n])
            assert np.array_equal(state[b, 1 + n:], start[b, 1 + n:])


@pytest.mark.gpu
def test_engine_mixed_rollout_equals_oracle(oracle_mod):
    """Fused rollout with in-kernel auto-reset under the mixed rule: same episode records as the oracle."""
    import crowdnav_amd
    B, cfg = 96, dict(num_humans=SLOTS, robot_visible=1, scenario_rule=2)
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, robot_policy=crowdnav_amd.ROBOT_ORCA, **cfg)
    bufs = eng.rollout_begin(seed_base=1000, seed_mod=500, record_capacity=8)
    eng.rollout(260)
    eng.sync()
    ora = oracle_mod.CrowdOracle(num_envs=B, robot_policy=1, **cfg)
    ora.reset(1000 + np.arange(B))
    total, rec = ora.rollout(260, 1000, 500, 8, np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.float64))
    assert int(bufs['transitions'].cpu()[0]) == total
    assert np.array_equal(bufs['ep_count'].cpu().numpy(), rec['count'])
    for key, mine in (('outcome', 'ep_outcome'), ('steps', 'ep_steps')):
        assert np.array_equal(bufs[mine].cpu().numpy(), rec[key]), key
    # device cos / sin vs libm in the scenario generator: starts agree to 1e-12, hence returns / states to ~1e-9
    assert np.abs(bufs['ep_return'].cpu().numpy() - rec['ret']).max() <= 1e-9
    assert np.abs(eng.get_state()[0].cpu().numpy() - ora.get_state()[0]).max() <= 1e-9
    assert np.array_equal(eng.human_count().cpu().numpy(), ora.human_count())


@pytest.mark.gpu
def test_gym_surface_mixed_rule():
    """compat.CrowdSim with test_sim = mixed: env.humans / env.human_num as the reference leaves them, episodes run."""
    import crowdnav_amd.compat as c
    g = load_golden('mixed.npz')
    cfg = c.default_env_config({('sim', 'test_sim'): 'mixed', ('sim', 'train_val_sim'): 'mixed',
                                ('robot', 'visible'): 'true'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    robot.set_policy(c.policy_factory['orca']())
    env.set_robot(robot)
    for case in (0, 1, 2, 3, 4, 5, 11):
        ob = env.reset('test', case)
        assert len(ob) == len(env.humans) == g['reset_count'][case] and env.human_num == g['reset_human_num'][case]
        want = g['reset_states'][case]
        got = np.array([[h.px, h.py, h.vx, h.vy, h.gx, h.gy, h.radius, h.v_pref] for h in env.humans])
        assert np.abs(got - want[1:1 + len(env.humans)]).max() <= 1e-12
    eps = list(episodes(g))
    ob = env.reset('test', 3)
    done, t = False, 0
    while not done:
        ob, reward, done, info = env.step(robot.act(ob))
        assert len(ob) == eps[3]['count'] and abs(reward - eps[3]['rewards'][t]) <= 1e-9  # start differs by <= 1e-12
        t += 1
    assert t == len(eps[3]['actions'])


# ---- value networks acting under the mixed rule (a different number of humans per episode) -----------------------------
def _parked_states(states):
    """fixture rows padded with NaN -> the engine's representation: absent humans parked at rest far away, behind the
    present ones (scenario_device.h: kParkedX)."""
    s = states.copy()
    for b in range(len(s)):
        for i in range(1, 6):
            if np.isnan(s[b, i, 0]):
                x = 1.0e6 + 100.0 * i
                s[b, i] = [x, 1.0e6, 0.0, 0.0, x, 1.0e6, 0.3, 1.0]
    return s


@pytest.mark.gpu
@pytest.mark.parametrize('kernels', ['by-size', 'register-resident', 'narrow-tiles'])
@pytest.mark.parametrize('policy', ['sarl', 'cadrl', 'lstm_rl'])
def test_value_networks_mask_the_absent_humans_of_a_mixed_episode(policy, kernels, monkeypatch):
    """SARL (attention: mean and softmax over the humans present), CADRL (minimum over them) and LSTM-RL (one LSTM step per
    human present) on episodes with 1, 2, 3 and 5 humans, vs the unmodified reference acting under test_sim = mixed.  At this
    batch size the LDS kernels run; CROWDNAV_AMD_SARL_REG=2 forces the register-resident ones; CROWDNAV_AMD_SARL_NARROW=2 the
    narrow tiles of the single-episode sampling route (round 6: SARL and CADRL mask a group's absent humans there as well,
    LSTM-RL skips their steps) — the launch counter says which ran."""
    import torch
    import crowdnav_amd
    if kernels == 'register-resident':
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')
    monkeypatch.setenv('CROWDNAV_AMD_SARL_NARROW', '2' if kernels == 'narrow-tiles' else '0')
    from crowdnav_amd.compat import cadrl, lstm_rl, sarl
    g = load_golden('mixed_sarl.npz')
    pre = policy + '_'
    states = _parked_states(g[pre + 'states'])
    n = len(states)
    assert sorted(set(g[pre + 'count'].tolist())) == [1, 2, 3, 5]
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1,
                                       scenario_rule=crowdnav_amd.MIXED)
    eng.set_state(states, g[pre + 'gtime'])
    assert np.array_equal(eng.human_count().cpu().numpy(), g[pre + 'count'])
    params = {k[len(pre + 'param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre + 'param_')}
    if policy == 'sarl':
        net = sarl.ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
        cfg = {}
    elif policy == 'cadrl':
        net = cadrl.ValueNetwork(13, [150, 100, 100, 1])
        cfg = dict(model='cadrl', mlp3_dims=(150, 100, 100, 1))
    else:
        net = lstm_rl.ValueNetwork1(13, 6, [150, 100, 100, 1], 50)
        cfg = dict(model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    net.load_state_dict(params)
    eng.sarl_configure(actions=g[pre + 'action_space'], gamma=0.9, **cfg)
    eng.sarl_set_weights(net.state_dict())
    out = eng.sarl_select()
    eng.sync()
    assert eng.launch_counts()['sarl_narrow'] == (1 if kernels == 'narrow-tiles' else 0)
    values = out['values'].cpu().numpy()
    assert np.abs(values - g[pre + 'values']).max() <= 1e-6
    top2 = np.sort(g[pre + 'values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), out['best'].cpu().numpy(), g[pre + 'best'], g[pre + 'values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 4
    assert np.array_equal(out['best'].cpu().numpy()[clear], g[pre + 'best'][clear])
    assert np.array_equal(out['action'].cpu().numpy()[clear], g[pre + 'action'][clear])


@pytest.mark.gpu
def test_mixed_value_networks_need_five_slots():
    import crowdnav_amd
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=4, num_humans=6, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                       scenario_rule=crowdnav_amd.MIXED)
    acts = np.zeros((81, 2))
    with pytest.raises(crowdnav_amd.CrowdNavAmdError):
        eng.sarl_configure(actions=acts)  # 6 slots stream through the chunked kernel, which does not mask


@pytest.mark.gpu
def test_mixed_value_network_lds_limit_at_the_boundary_width():
    """under the mixed rule the one-tile SARL kernel must hold a tile's activations AND the pipelined side buffer
    in 160 KiB of LDS (include/crowdnav_amd.h: cn_sarl_configure).  At 5 humans a first mlp1 layer of up to 160 fits (the
    shipped 150 pads to it: 154 112 B), 176 does not (164 928 B, although the activations alone — 153 408 B — would): configured
    and run / refused with a message that says why.  Outside the mixed rule the same 176-wide network streams through the
    chunked kernel."""
    import torch
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork
    acts = np.zeros((81, 2))
    acts[1:, 0] = np.linspace(-1, 1, 80)

    def engine(rule):
        return crowdnav_amd.BatchedCrowdSim(num_envs=4, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                            robot_visible=1, scenario_rule=rule)

    def run(eng, width, compare):
        eng.sarl_configure(actions=acts, mlp1_dims=(width, 100))
        torch.manual_seed(0)
        net = ValueNetwork(13, 6, [width, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
        eng.sarl_set_weights(net.state_dict())
        eng.reset(1000 + np.arange(4))
        sel = eng.sarl_select()
        eng.sync()
        assert np.isfinite(sel['values'].cpu().numpy()).all()
        if compare:  # (a mixed episode's absent humans are masked out of the attention: compared in the test above)
            with torch.no_grad():
                want = net(eng.sarl_export('X').cpu().reshape(4 * 81, 5, 13)).reshape(4, 81).numpy()
            assert np.abs(eng.sarl_export('V').cpu().numpy() - want).max() <= 1e-5

    run(engine(crowdnav_amd.MIXED), 160, False)
    with pytest.raises(crowdnav_amd.CrowdNavAmdError) as ei:
        engine(crowdnav_amd.MIXED).sarl_configure(actions=acts, mlp1_dims=(176, 100))
    assert ei.value.status == -2  # CN_ERR_UNSUPPORTED
    assert '160 KiB' in str(ei.value) and 'side buffer' in str(ei.value)
    run(engine(crowdnav_amd.CIRCLE_CROSSING), 160, True)   # one-tile kernel
    run(engine(crowdnav_amd.CIRCLE_CROSSING), 176, True)   # chunked kernel: the same network, streamed
