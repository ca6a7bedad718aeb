"""RL-phase sampling of BASELINE configs[4] (train.py:147-157): epsilon-greedy device SARL episodes in lock step,
replay states from cn_sarl_transform, batched TD targets, device replay ring — against fixtures produced by the
UNMODIFIED reference's Explorer.run_k_episodes(k, 'train', update_memory=True) (oracle/gen_golden_rl.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

RL_FIXTURES = ['rl_sarl_plain.npz', 'rl_sarl_om.npz', 'rl_cadrl.npz', 'rl_lstm_rl.npz', 'rl_lstm_rl_om.npz',
               'rl_lstm_rl2.npz']


# ------------------------------------------------------------------------------------------------ CPU
def test_device_replay_memory_is_the_reference_ring_cpu():
    """push_batch == the same pushes one by one into the reference-style list ring (memory.py:12-18)."""
    from crowdnav_amd.compat.trainer import DeviceReplayMemory, ReplayMemory
    torch.manual_seed(1)
    dev, ref = DeviceReplayMemory(10, 'cpu'), ReplayMemory(10)
    for n in (3, 4, 5, 12, 25, 1, 7, 10):
        st, v = torch.randn(n, 2, 3), torch.randn(n)
        dev.push_batch(st, v)
        for i in range(n):
            ref.push((st[i], v[i].reshape(1)))
        assert len(dev) == len(ref) and dev.position == ref.position and dev.is_full() == ref.is_full()
        for i in range(len(ref)):
            assert torch.equal(dev[i][0], ref[i][0]) and torch.equal(dev[i][1], ref[i][1])
    dev.push((torch.ones(2, 3), torch.tensor([2.0])))
    ref.push((torch.ones(2, 3), torch.tensor([2.0])))
    assert all(torch.equal(dev[i][0], ref[i][0]) for i in range(10))
    with pytest.raises(IndexError):
        dev[10]
    dev.clear()
    assert len(dev) == 0


def test_trainer_on_device_memory_cpu():
    from crowdnav_amd.compat.sarl import ValueNetwork
    from crowdnav_amd.compat.trainer import DeviceReplayMemory, Trainer
    torch.manual_seed(0)
    mem = DeviceReplayMemory(64, 'cpu')
    x = torch.randn(50, 5, 13)
    mem.push_batch(x, x[:, 0, 0] * 0.1)
    seen = torch.cat([b[1] for b in mem.batches(16)])
    assert seen.shape == (50, 1) and torch.equal(seen.sort(0)[0], mem.values[:50].sort(0)[0])  # one permutation
    assert [b[0].shape[0] for b in mem.batches(16)] == [16, 16, 16, 2]
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    tr = Trainer(net, mem, torch.device('cpu'), batch_size=16)
    with pytest.raises(ValueError):
        tr.optimize_epoch(1)
    tr.set_learning_rate(0.01)
    first = tr.optimize_epoch(1)
    later = tr.optimize_epoch(25)
    assert later < first and np.isfinite(tr.optimize_batch(3))


def test_target_model_updates_in_place_and_td_values_cpu():
    """Explorer.update_target_model (explorer.py:26-27: copy.deepcopy): the first call copies, later calls with the same
    architecture overwrite the copy's parameters in place (same values; a captured graph of its forward stays valid), another
    architecture is copied afresh; _td_values on a CPU model is the plain forward."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import ValueNetwork
    torch.manual_seed(2)
    a = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    b = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    ex = c.Explorer(None, None, torch.device('cpu'), None, 0.9)
    ex.update_target_model(a)
    first = ex.target_model
    assert first is not a and all(torch.equal(p, q) for p, q in zip(first.parameters(), a.parameters()))
    ex.update_target_model(b)
    assert ex.target_model is first and all(torch.equal(p, q) for p, q in zip(first.parameters(), b.parameters()))
    with torch.no_grad():
        next(b.parameters()).add_(1.0)                     # the source moves on: the target keeps what it was given
    assert not torch.equal(next(first.parameters()), next(b.parameters()))
    x = torch.randn(7, 5, 13)
    with torch.no_grad():
        assert torch.equal(ex._td_values(x), first(x).reshape(-1))
    b.eval()
    ex.update_target_model(b)                                # in place, and the mode travels with it as it would with deepcopy
    assert ex.target_model is first and not first.training
    b.train()
    ex.update_target_model(b)
    assert first.training
    other = ValueNetwork(13, 6, [64, 32], [32, 16], [64, 32, 32, 1], [32, 32, 1], True, 1.0, 4)
    ex.update_target_model(other)
    assert ex.target_model is not first and ex.target_model is not other


@pytest.mark.gpu
def test_td_value_graph_follows_parameters_that_move():
    """the captured forward of the target network reads the parameters' storage — parameters that get NEW storage
    (.to(), .half(), load_state_dict(assign=True), a user assigning explorer.target_model) must be captured again, not replayed
    on the old weights."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import ValueNetwork
    torch.manual_seed(4)
    dev = torch.device('cuda:0')
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4).to(dev)
    ex = c.Explorer(None, None, dev, None, 0.9)
    ex.update_target_model(net)
    x = torch.randn(9, 5, 13, device=dev)
    with torch.no_grad():
        v0 = ex._td_values(x).clone()
        assert ex._td_graph is not None and torch.allclose(v0, ex.target_model(x).reshape(-1), atol=1e-6)
        graph0 = ex._td_graph['graph']
        for p_ in ex.target_model.parameters():              # the same module, other storage, other values
            p_.data = p_.data.clone() * 0.5
        v1 = ex._td_values(x)
        assert ex._td_graph['graph'] is not graph0
        assert torch.allclose(v1, ex.target_model(x).reshape(-1), atol=1e-6) and not torch.allclose(v1, v0)


def test_rl_fixture_is_self_consistent_cpu():
    """The reference's memory holds exactly the steps of its ReachGoal / Collision episodes; terminal targets are the
    terminal rewards (explorer.py:111-113)."""
    for name in RL_FIXTURES:
        g = load_golden(name)
        kept = [e for e in range(int(g['k'])) if g['ep_outcome'][e] in (2, 3)]
        assert sum(int(g['ep_steps'][e]) for e in kept) == len(g['memory_values'])
        row = 0
        for e in kept:
            n = int(g['ep_steps'][e])
            assert g['memory_values'][row + n - 1] == np.float32(g['ep_rewards'][e][n - 1])
            row += n


# ------------------------------------------------------------------------------------------------ GPU
def _setup(g):
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    with_om, visible = bool(int(g['with_om'])), bool(int(g['robot_visible']))
    name = str(g['policy']) if 'policy' in g else 'sarl'
    cfg = c.default_env_config({('robot', 'visible'): 'true' if visible else 'false'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    policy = c.policy_factory[name]()
    overrides = {(name, 'with_om'): 'true' if with_om else 'false'} if name != 'cadrl' else {}
    if 'pairwise' in g and int(g['pairwise']):
        overrides[('lstm_rl', 'with_interaction_module')] = 'true'
    policy.configure(default_policy_config(overrides))
    policy.get_model().load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items()
                                        if k.startswith('param_')})
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_device(torch.device('cpu'))
    policy.set_env(env)
    policy.set_epsilon(float(g['epsilon']))
    return c, env, robot, policy


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['CROWDNAV_AMD_SARL_NARROW', 'CROWDNAV_AMD_SARL_FUSED_STEP', 'CROWDNAV_AMD_RL_PINNED'])
def test_single_episode_sampling_with_pinned_histories_off_the_two_launch_route(route, monkeypatch):
    """The reward / min-distance / action histories of a single-episode call live in pinned host memory (round 6).  Only the
    two-launch route orders a step's outputs before its info code; on the one-tile kernels (NARROW=0: cn_sarl_explore READS the
    action row back over the host link) and on the three-launch route (FUSED_STEP=0) the host waits for the device before it
    reads them; PINNED=0 keeps them on the device.  Same episodes, same replay memory as the reference every time."""
    from crowdnav_amd.compat.trainer import DeviceReplayMemory
    monkeypatch.setenv(route, '0')
    g = load_golden('rl_sarl_plain.npz')
    c, env, robot, policy = _setup(g)
    k = int(g['k'])
    mem = DeviceReplayMemory(100000, 'cuda:0')
    dev = torch.device('cuda:0')
    policy.get_model().to(dev)
    policy.set_device(dev)
    ex = c.Explorer(env, robot, dev, mem, float(g['gamma']), target_policy=policy)
    ex.max_envs = 1
    ex.update_target_model(policy.get_model())
    env.case_counter['train'] = int(g['first_case'])
    ex.run_k_episodes(k, 'train', update_memory=True, episode=0)
    lb = ex.last_batch
    assert lb['outcome'] == g['ep_outcome'].tolist() and lb['steps'] == g['ep_steps'].tolist()
    for e in range(k):
        assert lb['actions'][e] == g['ep_actions'][e][:int(g['ep_steps'][e])].tolist(), e
    assert len(mem) == len(g['memory_values'])
    states = torch.stack([mem[i][0].cpu() for i in range(len(mem))]).numpy()
    values = torch.cat([mem[i][1].cpu() for i in range(len(mem))]).numpy()
    assert np.abs(states - g['memory_states']).max() <= 5e-6 and np.abs(values - g['memory_values']).max() <= 1e-6
    counts = ex._rl_engine_cache[1].launch_counts()
    if route == 'CROWDNAV_AMD_RL_PINNED':
        assert counts['sarl_decide_steps'] == counts['sarl_narrow'] > 0 and not ex._rl_hist[3].is_cuda and ex._rl_hist[2].is_cuda
    else:
        assert counts['sarl_decide_steps'] == 0 and not ex._rl_hist[2].is_cuda   # pinned histories, the host waited


@pytest.mark.gpu
def test_rl_engine_cache_follows_everything_the_engine_is_built_from():
    """Explorer._rl_engine keeps ONE engine between the 10 000 single-episode calls of train.py — found again by a fast key of
    what engine_config reads (by value, or by identity for the config object, the policy and its action-space list): any of
    those changing must give another engine, nothing else may."""
    g = load_golden('rl_sarl_plain.npz')
    c, env, robot, policy = _setup(g)
    policy.build_action_space(robot.v_pref)
    ex = c.Explorer(env, robot, torch.device('cpu'), None, float(g['gamma']), target_policy=policy)
    e0 = ex._rl_engine(1, 5, 'circle_crossing')
    assert ex._rl_engine(1, 5, 'circle_crossing') is e0           # the fast path
    env.discomfort_dist = env.discomfort_dist + 0.05             # a value engine_config reads
    e1 = ex._rl_engine(1, 5, 'circle_crossing')
    assert e1 is not e0 and ex._rl_engine(1, 5, 'circle_crossing') is e1
    policy.action_space = list(policy.action_space)              # a rebuilt table (same values: still a new engine, by identity)
    e2 = ex._rl_engine(1, 5, 'circle_crossing')
    assert e2 is not e1
    assert ex._rl_engine(2, 5, 'circle_crossing') is not e2      # another batch size
    robot.time_step = 0.25                                       # not part of the engine: no new engine
    e3 = ex._rl_engine(2, 5, 'circle_crossing')
    assert ex._rl_engine(2, 5, 'circle_crossing') is e3


@pytest.mark.gpu
@pytest.mark.parametrize('randomize,humans', [(False, 5), (True, 5), (False, 12), (True, 14)])
def test_explore_continues_each_envs_numpy_stream(randomize, humans):
    """cn_sarl_explore draws from the stream np.random.seed(seed) + the scenario's random() calls left behind — with the
    lane-per-scenario generator (up to 8 humans) and with the wave-per-scenario one, whose window of tempered words runs up
    to a block ahead of the read position (it keeps the previous block's state for exactly this)."""
    import crowdnav_amd
    B, K = 48, 81
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                       robot_visible=1, randomize_attributes=int(randomize),
                                       circle_radius=4.0 if humans == 5 else (5.0 if not randomize else 8.0))
    acts = np.stack([np.arange(K), -np.arange(K)], axis=1).astype(np.float64)
    eng.sarl_configure(actions=acts)
    seeds = 2000 + 17 * np.arange(B)
    draws = eng.reset(seeds).cpu().numpy()
    rngs = []
    for b in range(B):
        rs = np.random.RandomState(int(seeds[b]))
        rs.random_sample(int(draws[b]))
        rngs.append(rs)
    for rnd, eps in enumerate((1.0, 0.4, 0.0, 0.7)):
        mask = np.ones(B, np.uint8)
        mask[rnd::5] = 0                                  # masked envs draw nothing
        best = torch.full((B,), 5, dtype=torch.int32, device=eng.device)
        best[1] = -1                                      # an env at its goal returns before the draw
        sel = dict(best=best, action=torch.zeros(B, 2, dtype=torch.float64, device=eng.device))
        eng.sarl_explore(sel, eps, mask=mask)
        eng.sync()
        got_best, got_act, got_exp = (sel[k].cpu().numpy() for k in ('best', 'action', 'explored'))
        for b in range(B):
            if not mask[b] or b == 1:
                want_best, want_exp = (-1 if b == 1 else 5), 0
            else:
                p = rngs[b].random_sample()
                want_exp = int(p < eps)
                want_best = int(rngs[b].choice(K)) if want_exp else 5
            assert got_best[b] == want_best and got_exp[b] == want_exp, (rnd, b)
            if want_exp:
                assert tuple(got_act[b]) == (want_best, -want_best)


@pytest.mark.gpu
def test_explore_without_a_kept_stream_fails_loudly():
    import crowdnav_amd
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=4, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.sarl_configure(actions=np.zeros((81, 2)))
    eng.rollout_begin(seed_base=1000, seed_mod=500, record_capacity=2)  # the ring path does not keep numpy streams
    sel = dict(best=torch.zeros(4, dtype=torch.int32, device=eng.device),
               action=torch.zeros(4, 2, dtype=torch.float64, device=eng.device))
    eng.sarl_explore(sel, 0.5)
    with pytest.raises(crowdnav_amd.CrowdNavAmdError):
        eng.sync()
    eng.sync()  # the flag is cleared once reported


@pytest.mark.gpu
@pytest.mark.parametrize('with_om', [False, True])
def test_transform_matches_host_transform(with_om):
    """cn_sarl_transform == the policy's own transform (the vectorised mirror of MultiHumanRL.transform that is itself
    checked against the reference) on live states, dense and strided output."""
    import crowdnav_amd
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config, occupancy_maps, rotate
    from crowdnav_amd.compat.types import ObservableState
    B, H = 32, 5
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    policy = c.policy_factory['sarl']()
    policy.configure(default_policy_config({('sarl', 'with_om'): 'true' if with_om else 'false'}))
    policy.build_action_space(1.0)
    eng.sarl_configure(**policy.engine_kwargs())
    eng.reset(3000 + np.arange(B))
    rng = np.random.RandomState(0)
    D = policy.input_dim()
    traj = torch.zeros(B, 3, H, D, dtype=torch.float32, device=eng.device)
    for t in range(3):
        eng.step(rng.uniform(-0.7, 0.7, size=(B, 2)), update=True, want_obs=False)
        st = eng.get_state()[0].cpu().numpy()                                   # [B, A, 8] px py vx vy gx gy r vpref
        dense = eng.sarl_transform()
        eng.sarl_transform(out=traj[:, t], env_stride=3 * H * D)
        eng.sync()
        me = np.concatenate([st[:, 0, [0, 1, 2, 3, 6, 4, 5, 7]], np.full((B, 1), np.pi / 2)], axis=1)   # FullState order
        hum = st[:, 1:][:, :, [0, 1, 2, 3, 6]]
        joint = torch.Tensor(np.concatenate([np.repeat(me[:, None], H, axis=1), hum], axis=2))         # [B, H, 14] f32
        want = rotate(joint.reshape(B * H, 14)).reshape(B, H, 13)
        if with_om:
            maps = torch.stack([occupancy_maps([ObservableState(*row) for row in hum[b].tolist()], policy.cell_num,
                                               policy.cell_size, policy.om_channel_size) for b in range(B)])
            want = torch.cat([want, maps], dim=2)
        assert torch.equal(dense.cpu(), traj[:, t].cpu())
        assert (dense.cpu() - want).abs().max().item() <= 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize('B,humans,n_act,with_om', [(1, 5, 81, False), (3, 5, 81, False), (16, 5, 81, False), (2, 3, 81, False),
                                                    (40, 5, 81, False), (6, 1, 13, False), (9, 2, 5, False),
                                                    (1, 5, 81, True), (3, 5, 81, True), (2, 3, 81, True), (9, 2, 5, True),
                                                    (1, 5, 81, 'lstm_rl'), (3, 4, 81, 'lstm_rl'), (2, 5, 81, 'lstm_rl + maps')])
def test_sample_step_is_the_five_calls_it_replaces(B, humans, n_act, with_om, monkeypatch):
    """cn_sarl_sample_step (ABI v8) = alive &= ~done; cn_sarl_select; cn_sarl_explore(mask = alive); cn_sarl_transform;
    cn_step.  Three engines on the same seeds and weights for 104 steps (every episode ends, envs leave `alive`, the epsilon-greedy
    draws continue each env's numpy stream; the histories start from zero): (a) the one call on the narrow-tile route (two launches per step — the network, then decision + transition + the
    next decision's ORCA velocities in one kernel, with another entry point in between every 13 steps — and, with
    CROWDNAV_AMD_SARL_FUSED_STEP=0, three: ORCA, the network with the decision by its last workgroup, the transition; forced with
    CROWDNAV_AMD_SARL_NARROW=2: by size it is taken up to one workgroup per CU, 9 envs of 5 humans; 40 envs are 1080 tiles and
    five envs per wave of the deciding workgroup), (b) the one call with CROWDNAV_AMD_SARL_NARROW=0 (the general route inside
    the call), (c) the five calls by hand on the one-tile kernels.  Every history is the same bits.  The last two cases have a
    small action table — FEWER narrow tiles than envs (6 envs x 13 actions of one human are 5 tiles, 9 x 5 of two humans
    6): every env's replay-memory state must still be written.  with_om (round 6): occupancy maps on the narrow route — the
    maps of the next decision come from sarl_decide_step_kernel (streamed calls) or sarl_lookahead_kernel (after another entry
    point); the launch counters prove which kernels ran.  'lstm_rl' (round 6): lstm_rl.ValueNetwork1 on the narrow tiles
    (sarl_narrow_kernel<true>); its replay-memory states are in LstmRL.predict's order (humans by decreasing distance)."""
    import ctypes as C
    import crowdnav_amd
    from crowdnav_amd._lib import check
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(21)
    lstm = isinstance(with_om, str)
    if lstm:
        with_om = with_om.endswith('maps')
    D = 61 if with_om else 13
    if lstm:
        net = lstm_rl.ValueNetwork1(D, 6, [150, 100, 100, 1], 50)
        net_kwargs = dict(model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    else:
        net = ValueNetwork(D, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
        net_kwargs = {}
    space, _, _ = build_action_space(1.0)
    T = 104

    def nudge(eng, t):
        # the humans' ORCA velocities the fused kernel left for the next decision belong to the state it wrote; any
        # other entry point may change that state (here: every human 1-3 cm aside, alternating cn_set_state and a plain cn_step
        # every route gets the same nudge at the same step)
        st, gt = eng.get_state()
        st[:, 1:, 0] += 0.01 * (1 + (t // 13) % 3)
        eng.set_state(st, gt)

    def run(narrow, one_call, fused='1'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_NARROW', narrow)
        monkeypatch.setenv('CROWDNAV_AMD_SARL_FUSED_STEP', fused)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=0)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space[:n_act]]), with_om=with_om, **net_kwargs)
        eng.sarl_set_weights(net.state_dict())
        eng.reset(7000 + np.arange(B))
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=eng.device)  # noqa: E731
        traj, rew, inf, dmn = z((B, T, humans, D), torch.float32), z((T, B), torch.float64), z((T, B), torch.uint8), z((T, B), torch.float64)
        act, alive, done, action = z((T, B), torch.int32), z((B,), torch.uint8), z((B,), torch.uint8), z((B, 2), torch.float64)
        alive.fill_(1)
        alive_hist = []
        if one_call:
            step = eng.sarl_sampler(traj, rew, inf, dmn, act, alive, done, action)
            for t in range(T):
                step(t, 0.3)
                alive_hist.append(alive.clone())
                if t % 13 == 12:
                    nudge(eng, t)   # another entry point CHANGES the state: the next call must not trust the velocities the last one left
        else:
            lib, h, V = eng._lib, eng._h, C.c_void_p
            for t in range(T):
                alive.masked_fill_(done.view(torch.bool), 0)
                best = V(act.data_ptr() + 4 * B * t)
                check(lib.cn_sarl_select(h, None, best, V(action.data_ptr())))
                check(lib.cn_sarl_explore(h, 0.3, V(alive.data_ptr()), best, V(action.data_ptr()), None))
                check(lib.cn_sarl_transform(h, V(traj.data_ptr() + 4 * humans * D * t), T * humans * D, 1 if lstm else 0))
                check(lib.cn_step(h, V(action.data_ptr()), 1, V(rew.data_ptr() + 8 * B * t), V(done.data_ptr()), V(inf.data_ptr() + B * t),
                                  V(dmn.data_ptr() + 8 * B * t), None, None, None))
                alive_hist.append(alive.clone())
                if t % 13 == 12:
                    nudge(eng, t)
        eng.sync()
        counts = eng.launch_counts()
        if one_call:  # the route: narrow tiles for every step; decision + transition + next ORCA (+ maps) as one kernel when fused
            assert counts['sarl_narrow'] == (T if narrow == '2' else 0)
            assert counts['sarl_decide_steps'] == (T if narrow == '2' and fused == '1' else 0)
        out = [x.cpu().numpy() for x in (traj, rew, inf, dmn, act, torch.stack(alive_hist), eng.get_state()[0], action)]
        eng.close()
        return out

    a, a2, b, c = run('2', True), run('2', True, fused='0'), run('0', True), run('0', False)
    assert (a[5][-1] == 0).all() and (a[5][0] == 1).all()   # every episode ended (time_limit / time_step = 100 steps at the latest)
    # the three routes that step every env at every call agree everywhere; the two-launch route skips an env once its episode is
    # over (round 6: its rows of the histories are not written any more, its state stays as the last step left it), so it is
    # compared where the env was still sampling: alive after call t = the env took step t
    for x2, y, w in zip(a2, b, c):
        assert np.array_equal(x2, y) and np.array_equal(x2, w)
    sampled = a[5].astype(bool)                               # [T, B]
    assert np.array_equal(a[5], b[5])
    assert np.array_equal(a[0][sampled.T], b[0][sampled.T])   # replay-memory states [B, T, H, D]
    for k in (1, 2, 3, 4):                                    # reward, info, dmin, chosen action [T, B]
        assert np.array_equal(a[k][sampled], b[k][sampled]), k


@pytest.mark.gpu
@pytest.mark.parametrize('name,model_on_gpu', [('rl_sarl_plain.npz', False), ('rl_sarl_plain.npz', True), ('rl_cadrl.npz', True),
                                               ('rl_lstm_rl.npz', False), ('rl_lstm_rl.npz', True), ('rl_lstm_rl_om.npz', True),
                                               ('rl_sarl_om.npz', False), ('rl_sarl_om.npz', True)])
def test_single_episode_sampling_calls_reproduce_the_reference_memory(name, model_on_gpu):
    """train.py:156-170 samples ONE episode per call: the same fixtures with max_envs = 1 — every episode its own one-env batch
    (the narrow-tile route of cn_sarl_sample_step for SARL, one episode's slices of the histories as replay rows).  With the
    model on the GPU (`examples/train_sarl.py --gpu`) the TD targets come from cn_sarl_values (13-wide rows) or the graph-replayed
    forward of the target network, whose parameters a second update_target_model overwrites in place."""
    from crowdnav_amd.compat.trainer import DeviceReplayMemory
    g = load_golden(name)
    c, env, robot, policy = _setup(g)
    k = int(g['k'])
    mem = DeviceReplayMemory(100000, 'cuda:0')
    dev = torch.device('cuda:0' if model_on_gpu else 'cpu')
    if model_on_gpu:
        policy.get_model().to(dev)
        policy.set_device(dev)
    ex = c.Explorer(env, robot, dev, mem, float(g['gamma']), target_policy=policy)
    ex.max_envs = 1
    scrambled = __import__('copy').deepcopy(policy.get_model())
    with torch.no_grad():
        for p_ in scrambled.parameters():
            p_.mul_(0.5)
    ex.update_target_model(scrambled)
    first = ex.target_model
    ex.update_target_model(policy.get_model())   # same architecture: in place
    assert ex.target_model is first
    assert all(torch.equal(a, b) for a, b in zip(ex.target_model.state_dict().values(), policy.get_model().state_dict().values()))
    env.case_counter['train'] = int(g['first_case'])
    ex.run_k_episodes(k, 'train', update_memory=True, episode=0)
    lb = ex.last_batch
    assert lb['outcome'] == g['ep_outcome'].tolist() and lb['steps'] == g['ep_steps'].tolist()
    for e in range(k):
        assert lb['actions'][e] == g['ep_actions'][e][:int(g['ep_steps'][e])].tolist(), e
    assert len(mem) == len(g['memory_values'])
    states = torch.stack([mem[i][0].cpu() for i in range(len(mem))]).numpy()
    values = torch.cat([mem[i][1].cpu() for i in range(len(mem))]).numpy()
    if states.ndim == 2:
        states = states[:, None, :]
    assert np.abs(states - g['memory_states']).max() <= 5e-6 and np.abs(values - g['memory_values']).max() <= 1e-6
    if model_on_gpu and name in ('rl_sarl_plain.npz', 'rl_lstm_rl.npz'):
        # 13-wide rows (round 6, ABI v11): the TD targets are cn_sarl_values on an engine that only holds the target's weights —
        # uploaded again after the second update_target_model (the scrambled ones would fail the 1e-6 above)
        assert ex._td_engine is not None and getattr(ex, '_td_graph', None) is None
        kept = sum(1 for o in lb['outcome'] if o in (2, 3))   # ReachGoal / Collision episodes enter the memory: one launch each
        assert ex._td_engine['eng'].launch_counts()['sarl_narrow'] == kept >= 1
        # ... and follows the target's parameters: changed in place (version counters) or given new storage (addresses)
        x = torch.stack([mem[i][0] for i in range(min(len(mem), 7))]).reshape(-1, states.shape[1], 13)
        with torch.no_grad():
            v0 = ex._td_values(x).clone()
            for p_ in ex.target_model.parameters():
                p_.mul_(0.5)
            v1 = ex._td_values(x).clone()
            assert torch.allclose(v1, ex.target_model(x).reshape(-1), atol=2e-6) and not torch.allclose(v1, v0)
            for p_ in ex.target_model.parameters():
                p_.data = p_.data.clone() * 2.0
            v2 = ex._td_values(x)
            assert torch.allclose(v2, ex.target_model(x).reshape(-1), atol=2e-6) and torch.allclose(v2, v0, atol=2e-6)
    elif model_on_gpu and not name.startswith('rl_lstm_rl'):
        assert ex._td_graph is not None   # (an nn.LSTM forward may refuse capture: then the eager path ran, with a warning)
    # which route sampled: SARL and LSTM-RL (with or without occupancy maps: round 6) and CADRL take the narrow tiles + the fused
    # decision / transition kernel — two launches per streamed step
    counts = ex._rl_engine_cache[1].launch_counts()
    steps_issued = counts['sarl_narrow']
    assert steps_issued >= int(g['ep_steps'].sum()) and counts['sarl_decide_steps'] == steps_issued


@pytest.mark.gpu
@pytest.mark.parametrize('name', RL_FIXTURES)
@pytest.mark.parametrize('device_memory', [False, True])
def test_batched_rl_sampling_reproduces_the_reference_memory(name, device_memory):
    """Explorer.run_k_episodes(k, 'train', update_memory=True) with the epsilon-greedy SARL robot: same episodes
    (action for action), same outcomes, and the same replay memory as the unmodified reference produced."""
    from crowdnav_amd.compat.trainer import DeviceReplayMemory, ReplayMemory
    g = load_golden(name)
    c, env, robot, policy = _setup(g)
    k = int(g['k'])
    mem = DeviceReplayMemory(100000, 'cuda:0') if device_memory else ReplayMemory(100000)
    ex = c.Explorer(env, robot, torch.device('cpu'), mem, float(g['gamma']), target_policy=policy)
    ex.update_target_model(policy.get_model())
    env.case_counter['train'] = int(g['first_case'])
    ex.run_k_episodes(k, 'train', update_memory=True, episode=0)
    lb = ex.last_batch
    assert env.case_counter['train'] == int(g['first_case']) + k
    assert lb['outcome'] == g['ep_outcome'].tolist() and lb['steps'] == g['ep_steps'].tolist()
    for e in range(k):
        n = int(g['ep_steps'][e])
        assert lb['actions'][e] == g['ep_actions'][e][:n].tolist(), e
    assert len(mem) == len(g['memory_values'])
    states = torch.stack([mem[i][0].cpu() for i in range(len(mem))]).numpy()
    values = torch.cat([mem[i][1].cpu() for i in range(len(mem))]).numpy()
    if 'policy' in g and str(g['policy']) == 'cadrl':
        assert states.shape[1:] == (13,)                            # CADRL.transform: one human, no human axis
        states = states[:, None, :]
    assert np.abs(states - g['memory_states']).max() <= 5e-6       # float32 rotate: device vs torch CPU libm
    assert np.abs(values - g['memory_values']).max() <= 1e-6       # TD targets through the target network
    assert ex.last_stats['collision_rate'] == float(np.mean(g['ep_outcome'] == 3))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['rl_sarl_plain.npz', 'rl_cadrl.npz', 'rl_lstm_rl_om.npz'])
def test_batched_rl_sampling_equals_the_gym_surface_loop(name):
    """The lock-step batch against the reference's own loop (_run_sequential) on the device-backed gym surface."""
    from crowdnav_amd.compat.trainer import ReplayMemory
    g = load_golden(name)

    def collect(sequential):
        c, env, robot, policy = _setup(g)
        mem = ReplayMemory(100000)
        ex = c.Explorer(env, robot, torch.device('cpu'), mem, 0.9, target_policy=policy)
        ex.update_target_model(policy.get_model())
        env.case_counter['train'] = 40
        assert ex.memory is mem
        if sequential:
            policy.set_phase('train')
            stats = ex._run_sequential(5, 'train', True, False)
            ex._report(5, 'train', None, False, *stats)
        else:
            ex.run_k_episodes(5, 'train', update_memory=True)
        return mem, dict(ex.last_stats)

    mem_b, stats_b = collect(False)
    mem_s, stats_s = collect(True)
    assert len(mem_b) == len(mem_s)
    for key in ('success_rate', 'collision_rate', 'too_close', 'collision_cases', 'timeout_cases'):
        assert stats_b[key] == stats_s[key], key
    assert abs(stats_b['total_reward'] - stats_s['total_reward']) <= 1e-9
    if len(mem_b):
        xb = torch.stack([m[0].cpu() for m in mem_b.memory])
        xs = torch.stack([m[0].cpu() for m in mem_s.memory])
        vb = torch.cat([m[1].cpu() for m in mem_b.memory])
        vs = torch.cat([m[1].cpu() for m in mem_s.memory])
        assert (xb - xs).abs().max().item() <= 5e-6 and (vb - vs).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_graph_captured_sgd_step_equals_eager_step():
    """Trainer on the device replay ring: the hipGraph replay of one SGD step updates the network exactly like the
    eager step (same batches: same torch seed), and capturing itself does not train."""
    import copy
    from crowdnav_amd.compat.sarl import ValueNetwork
    from crowdnav_amd.compat.trainer import DeviceReplayMemory, Trainer
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    base = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4).to(dev)
    mem = DeviceReplayMemory(1000, dev)
    x = torch.randn(400, 5, 13, device=dev)
    mem.push_batch(x, x[:, 0, 0] * 0.1 + 0.2)
    nets = {}
    for mode in ('graph', 'eager'):
        net = copy.deepcopy(base)
        tr = Trainer(net, mem, dev, batch_size=100)
        tr._graph_failed = mode == 'eager'
        tr.set_learning_rate(0.01)
        torch.manual_seed(7)
        tr.optimize_epoch(3)
        loss = tr.optimize_batch(5)
        assert np.isfinite(loss)
        assert (tr._graph is not None) == (mode == 'graph') and not (mode == 'graph' and tr._graph_failed)
        nets[mode] = net
    for (k, a), (_, b) in zip(nets['graph'].state_dict().items(), nets['eager'].state_dict().items()):
        assert (a - b).abs().max().item() <= 1e-6, k
    assert any((a - b).abs().max().item() > 1e-4
               for a, b in zip(nets['graph'].state_dict().values(), base.state_dict().values()))  # it did train


@pytest.mark.gpu
@pytest.mark.parametrize('with_om', [False, True])
def test_lstm_rl_transform_orders_humans_by_decreasing_distance(with_om):
    """cn_sarl_transform for CN_MODEL_LSTM_RL == LstmRL.predict's re-ordering followed by the policy's transform."""
    import crowdnav_amd
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config, occupancy_maps, rotate
    from crowdnav_amd.compat.types import ObservableState
    B, H = 24, 5
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    policy = c.policy_factory['lstm_rl']()
    policy.configure(default_policy_config({('lstm_rl', 'with_om'): 'true' if with_om else 'false'}))
    policy.build_action_space(1.0)
    eng.sarl_configure(**policy.engine_kwargs())
    eng.reset(4000 + np.arange(B))
    rng = np.random.RandomState(1)
    for t in range(3):
        eng.step(rng.uniform(-0.7, 0.7, size=(B, 2)), update=True, want_obs=False)
    st = eng.get_state()[0].cpu().numpy()
    st[3, 2, :2] = st[3, 4, :2] = st[3, 0, :2] + np.array([1.5, 2.0])       # an exact tie: stable, original order kept
    eng.set_state(st, np.zeros(B))
    got = eng.sarl_transform().cpu()
    for b in range(B):
        me = st[b, 0]
        d = [np.linalg.norm(np.array((st[b, 1 + j, 0], st[b, 1 + j, 1])) - np.array((me[0], me[1]))) for j in range(H)]
        order = sorted(range(H), key=lambda j: d[j], reverse=True)
        hum = st[b, 1:][order][:, [0, 1, 2, 3, 6]]
        self_row = np.array([me[0], me[1], me[2], me[3], me[6], me[4], me[5], me[7], np.pi / 2])
        joint = torch.Tensor(np.concatenate([np.repeat(self_row[None], H, axis=0), hum], axis=1))
        want = rotate(joint)
        if with_om:
            want = torch.cat([want, occupancy_maps([ObservableState(*row) for row in hum.tolist()], policy.cell_num,
                                                   policy.cell_size, policy.om_channel_size)], dim=1)
        assert (got[b] - want).abs().max().item() <= 5e-6, b
    assert sorted(range(H), key=lambda j: np.hypot(*(st[3, 1 + j, :2] - st[3, 0, :2])), reverse=True).index(1) + 1 == \
        sorted(range(H), key=lambda j: np.hypot(*(st[3, 1 + j, :2] - st[3, 0, :2])), reverse=True).index(3)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['il_sarl_om.npz', 'il_lstm_rl.npz'])
def test_batched_imitation_collection_reproduces_the_reference_memory(name):
    """Explorer.run_k_episodes(k, 'train', update_memory=True, imitation_learning=True) (train.py:115-129): the replay
    memory the unmodified reference filled from its ORCA demonstrations — device episodes, cn_sarl_transform in env
    order (LSTM-RL included: imitation learning stores MultiHumanRL.transform of the demonstrator's state, unsorted),
    Monte-Carlo returns of explorer.py:100-105."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    from crowdnav_amd.compat.trainer import DeviceReplayMemory
    g = load_golden(name)
    pname, with_om, visible = str(g['policy']), bool(int(g['with_om'])), bool(int(g['robot_visible']))
    cfg = c.default_env_config({('robot', 'visible'): 'true' if visible else 'false'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    target = c.policy_factory[pname]()
    target.configure(default_policy_config({(pname, 'with_om'): 'true' if with_om else 'false'}))
    target.set_device(torch.device('cpu'))
    il = c.policy_factory['orca']()
    il.multiagent_training, il.safety_space = target.multiagent_training, (0 if visible else 0.15)
    robot.set_policy(il)
    env.set_robot(robot)
    mem = DeviceReplayMemory(100000, 'cuda:0')
    ex = c.Explorer(env, robot, torch.device('cpu'), mem, float(g['gamma']), target_policy=target)
    env.case_counter['train'] = int(g['first_case'])
    ex.run_k_episodes(int(g['k']), 'train', update_memory=True, imitation_learning=True)
    assert len(mem) == len(g['memory_values'])
    states = mem.states[:len(mem)].cpu().numpy()
    values = mem.values[:len(mem), 0].cpu().numpy()
    assert np.abs(states - g['memory_states']).max() <= 5e-6
    assert np.array_equal(values, g['memory_values'])  # float64 Monte-Carlo sums, rounded once to float32
