"""SARL robot decision: the torch mirror of the value network (CPU) and the HIP pipeline behind cn_sarl_select
(GPU) against fixtures produced by the unmodified reference SARL.predict (oracle/gen_golden_sarl.py).
Tolerances (BASELINE north_star: positions/velocities 1e-5, here the float32 network): features 5e-6, network
output and action values 1e-6 absolute (values are O(0.1); measured 5e-8); the arg-max must agree whenever the reference's top
two values are further apart than that tolerance.  Lookahead rewards and next human states are float64 env
arithmetic and must be bit-identical."""
import numpy as np
import pytest
import torch

from conftest import load_golden, report_argmax

FIXTURES = ['sarl_plain.npz', 'sarl_om.npz']
SELECT_FIXTURES = FIXTURES + ['sarl_h12.npz']  # 12 humans: more than a tile's LDS holds, streamed in chunks of 5


def _mirror(g):
    from crowdnav_amd.compat.sarl import ValueNetwork
    in_dim = 13 + (48 if int(g['with_om']) else 0)
    net = ValueNetwork(in_dim, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


@pytest.mark.parametrize('name', FIXTURES)
def test_value_network_mirror_matches_reference_cpu(name):
    g = load_golden(name)
    net = _mirror(g)
    assert sum(p.numel() for p in net.parameters()) == (103702 if int(g['with_om']) else 96502)
    x = torch.from_numpy(g['inputs'])  # [decisions, 81, H, in_dim]
    d, k, h, f = x.shape
    with torch.no_grad():
        out = net(x.reshape(d * k, h, f)).reshape(d, k).numpy()
    assert np.abs(out - g['net_out']).max() <= 2e-6


def test_action_space_matches_reference_cpu():
    from crowdnav_amd.compat.sarl import build_action_space
    g = load_golden('sarl_plain.npz')
    space, speeds, rotations = build_action_space(1.0, 5, 16)
    assert np.array_equal(np.array([[a.vx, a.vy] for a in space]), g['action_space'])
    assert len(space) == 81 and speeds[-1] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('name', SELECT_FIXTURES)
def test_sarl_select_vs_reference(name):
    import crowdnav_amd
    g = load_golden(name)
    n = len(g['states'])
    with_om = bool(int(g['with_om']))
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=g['states'].shape[1] - 1,
                                       robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=int(g['robot_visible']))
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, with_om=with_om)
    eng.sarl_set_weights(_mirror(g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.array_equal(cpu(eng.sarl_export('reward')), g['rewards'])          # float64 lookahead: exact
    assert np.array_equal(cpu(eng.sarl_export('next_obs')), g['next_obs'])
    X = cpu(eng.sarl_export('X'))
    assert X.shape == g['inputs'].shape
    assert np.abs(X[..., :13] - g['inputs'][..., :13]).max() <= 5e-6
    if with_om:
        assert np.abs(X[..., 13:] - g['inputs'][..., 13:]).max() <= 5e-6
        assert np.abs(cpu(eng.sarl_export('om')) - g['inputs'][:, 0, :, 13:]).max() <= 5e-6
    V = cpu(eng.sarl_export('V'))
    assert np.abs(V - g["net_out"]).max() <= 1e-6
    values = cpu(out['values'])
    assert np.abs(values - g["values"]).max() <= 1e-6
    best = cpu(out['best'])
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), best, g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-5
    assert clear.sum() >= n // 4
    assert np.array_equal(best[clear], g['best'][clear])
    chosen = cpu(out['action'])
    assert np.array_equal(chosen[clear], g['action'][clear])
    # wherever the arg-max differs it is a numerical tie: the value picked is within tolerance of the maximum
    assert np.all(g['values'][np.arange(n), best] >= g['values'].max(axis=1) - 4e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('humans,with_om', [(5, False), (8, False), (9, False), (20, False), (7, True)])
def test_sarl_mlp_vs_torch_fp32_random_inputs(humans, with_om):
    """The MFMA value network alone against the torch fp32 module on random states (bigger, non-fixture batch);
    9 and 20 humans stream through the tile in chunks (sarl_mlp_chunked_kernel: a partial and four full chunks)."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(3)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 37  # not a multiple of 16 groups
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                       robot_visible=1, circle_radius=4.0 if humans <= 9 else 10.0)
    eng.reset(1000 + np.arange(B))
    eng.step(np.zeros((B, 2)), update=True)  # humans get non-zero velocities
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
    eng.sarl_set_weights(net.state_dict())
    out = eng.sarl_select()
    X = eng.sarl_export('X').cpu()
    V = eng.sarl_export('V').cpu().numpy()
    with torch.no_grad():
        want = net(X.reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert np.abs(V - want).max() <= 2e-5
    assert np.all(out['best'].cpu().numpy() >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize('humans,B,with_om', [(5, 1, False), (5, 3, False), (5, 16, False), (1, 2, False), (2, 3, False), (3, 2, False),
                                              (4, 5, False), (8, 2, False), (5, 1, True), (5, 3, True), (2, 3, True), (3, 4, True), (8, 2, True)])
def test_narrow_tile_value_network_is_bit_identical_to_the_one_tile_kernel(humans, B, with_om, monkeypatch):
    """A few decisions (train.py's single-episode sampling: one env, 81 groups) run sarl_narrow_kernel: tiles of 16 / H whole
    groups, one per workgroup, X built in LDS.  CROWDNAV_AMD_SARL_NARROW=0 keeps sarl_feature_kernel + sarl_mlp_pipe_kernel on
    the same engine configuration: V, the chosen actions and the exported X / next states are the same BITS (the narrow kernel
    sums over a group's humans, slices attention.4 and orders every layer's k loop as the one-tile kernel does), and both are
    within 2e-5 of the torch module.  81 B groups are never a multiple of the 3 / 4 / 5 / 8 / 16 groups of a narrow tile.
    with_om (round 6): 61-wide rows — the 48 map columns of a row come from the per-(env, human) occupancy maps, mlp1.0 runs its
    16 k-steps on the tile; the launch counter proves the narrow route ran."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(11)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    space, _, _ = build_action_space(1.0)
    got = {}
    for narrow in ('2', '0'):  # (2 = whenever the configuration allows it: by size the narrow tiles stop at one workgroup per CU)
        monkeypatch.setenv('CROWDNAV_AMD_SARL_NARROW', narrow)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
        eng.reset(5000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        # (8 humans stream through the tile in chunks: no narrow tiles, both runs take the same kernels)
        assert eng.launch_counts()['sarl_narrow'] == (1 if narrow == '2' and humans <= 5 else 0)
        got[narrow] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), out['action'].cpu().numpy(),
                       eng.sarl_export('X').cpu(), eng.sarl_export('next_obs').cpu().numpy())
        if with_om:
            got[narrow] += (eng.sarl_export('om').cpu().numpy(),)
        eng.close()
    with torch.no_grad():
        want = net(got['2'][3].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert np.abs(got['2'][0] - want).max() <= 2e-5
    for a, b in zip(got['2'], got['0']):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.gpu
@pytest.mark.parametrize('humans,B', [(5, 1), (5, 4), (1, 3), (3, 2), (8, 2)])
def test_narrow_tile_cadrl_network_is_bit_identical_to_the_one_tile_kernel(humans, B, monkeypatch):
    """cadrl.ValueNetwork (the row MLP, then the minimum over a group's humans) on the narrow tiles: the same bits as
    sarl_feature_kernel + cadrl_mlp_kernel (CROWDNAV_AMD_SARL_NARROW=0), within 2e-5 of the torch module."""
    import crowdnav_amd
    from crowdnav_amd.compat import cadrl
    from crowdnav_amd.compat.sarl import build_action_space
    torch.manual_seed(13)
    net = cadrl.ValueNetwork(13, [150, 100, 100, 1])
    space, _, _ = build_action_space(1.0)
    got = {}
    for narrow in ('2', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_NARROW', narrow)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
        eng.reset(6000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='cadrl', mlp3_dims=(150, 100, 100, 1))
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[narrow] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), out['action'].cpu().numpy(),
                       out['values'].cpu().numpy(), eng.sarl_export('X').cpu())
        eng.close()
    with torch.no_grad():
        rows = net(got['2'][4].reshape(B * 81 * humans, 13)).reshape(B, 81, humans)
    assert np.abs(got['2'][0] - rows.min(dim=2).values.numpy()).max() <= 2e-5
    for a, b in zip(got['2'], got['0']):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.gpu
@pytest.mark.parametrize('humans,B,with_om', [(5, 1, False), (5, 3, False), (5, 1, True), (5, 2, True), (1, 3, False), (2, 3, True),
                                              (3, 2, False), (4, 2, True), (8, 2, False)])
def test_narrow_tile_lstm_rl_network_is_bit_identical_to_the_one_tile_kernel(humans, B, with_om, monkeypatch):
    """lstm_rl.ValueNetwork1 on the narrow tiles (round 6; sarl_narrow_kernel<true>: rows = the tile's 16 / H groups, the humans
    are the LSTM's steps, W_ih x_t of every step up front, W_hh held in registers across the steps): the same bits as
    sarl_feature_kernel + lstm_mlp_kernel (CROWDNAV_AMD_SARL_NARROW=0), within 2e-5 of the torch module; the launch counter
    proves which route ran (61-wide rows at 8 humans need more LDS than a workgroup has: both runs take the one-tile kernels)."""
    import crowdnav_amd
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import build_action_space
    torch.manual_seed(70 + humans)
    d = 61 if with_om else 13
    net = lstm_rl.ValueNetwork1(d, 6, [150, 100, 100, 1], 50)
    space, _, _ = build_action_space(1.0)
    got = {}
    for narrow in ('2', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_NARROW', narrow)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
        eng.reset(7000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='lstm_rl', mlp1_dims=(50, 1),
                           mlp3_dims=(150, 100, 100, 1), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        assert eng.launch_counts()['sarl_narrow'] == (1 if narrow == '2' else 0)
        got[narrow] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), out['action'].cpu().numpy(),
                       out['values'].cpu().numpy(), eng.sarl_export('X').cpu(), eng.sarl_export('next_obs').cpu().numpy())
        eng.close()
    with torch.no_grad():
        want = net(got['2'][4].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert np.abs(got['2'][0] - want).max() <= 2e-5
    for a, b in zip(got['2'], got['0']):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.gpu
@pytest.mark.parametrize('with_om', [False, True, 'maps inside the kernel'])
def test_register_resident_and_lds_value_networks_agree(with_om, monkeypatch):
    """5 humans at the shipped widths run sarl_reg_kernel (activations in registers); CROWDNAV_AMD_SARL_REG=0 keeps the LDS
    pipe kernel on the same engine configuration.  Same inputs, same weights: both within 2e-5 of torch and within 1e-6 of
    each other (they add the bias at opposite ends of the same fma chain), including a last tile of padding groups.  With
    occupancy maps the default hoists their half of mlp1.0 out of the action loop (sarl_om_term_kernel + sarl_reg_kernel<4, 5,
    true>); CROWDNAV_AMD_SARL_OM_HOIST=0 keeps all 16 k-steps inside sarl_reg_kernel<16, 5>."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    if with_om == 'maps inside the kernel':
        monkeypatch.setenv('CROWDNAV_AMD_SARL_OM_HOIST', '0')
        with_om = True
    torch.manual_seed(5)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 203  # 16 443 groups = 1027 full tiles + 11 groups: more tiles than the 1024 persistent waves, and a ragged last one
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):  # 1 = by size: 1028 tiles are more than the 512 up to which the LDS kernel finishes first
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81, 5, d)).reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0] - want).max() <= 2e-5 and np.abs(got['0'][0] - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 1e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize('humans', [1, 2, 3, 4])
@pytest.mark.parametrize('with_om', [False, True])
def test_register_resident_value_network_for_one_to_four_humans(humans, with_om, monkeypatch):
    """sarl_reg_kernel<XKS, NT> for crowds of 1..4 humans (NT N tiles per wave; 3 / 2 waves per SIMD at 1 / 2 humans) against the
    torch module and the LDS kernel on the same engine configuration, with a ragged last tile and more tiles than resident
    waves at 3 and 4 humans."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    if with_om and humans == 1:
        pytest.skip('occupancy maps need another human (multi_human_rl.py:117): cn_sarl_configure refuses, as the reference does')
    torch.manual_seed(20 + humans)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 203
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=1)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0] - want).max() <= 2e-5 and np.abs(got['0'][0] - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 1e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize('humans,with_om', [(6, False), (6, True), (7, False), (9, True), (10, False), (13, True)])
def test_register_resident_value_network_for_more_than_five_humans(humans, with_om, monkeypatch):
    """sarl_reg_chunk_kernel<NT, PRE>: the humans pass in chunks of 3 or 4 (6 = 3 + 3, 7 = 4 + 3 with one repeated and masked
    row, 9 = 3 x 3, 10 = 4 + 4 + 2, 13 = 4 x 4 - 3), mlp1's output parked in the per-wave scratch between the two passes; against
    the torch module and the chunked LDS kernel, on more tiles than resident waves and a ragged last tile."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(80 + humans)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 203
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=1, circle_radius=6.0)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0] - want).max() <= 2e-5 and np.abs(got['0'][0] - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 2e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize('humans', [1, 2, 3, 4, 5, 6, 7, 11])
def test_register_resident_cadrl_value_network(humans, monkeypatch):
    """cadrl_reg_kernel<NT> (cadrl.ValueNetwork with the activations in registers, minimum over the humans in the epilogue)
    against the torch module and the LDS kernels, on more tiles than resident waves and a ragged last tile; more than 5 humans
    pass in chunks (6 = 3 + 3, 7 = 4 + 3 with one repeated row, 11 = 4 + 4 + 3 with one)."""
    import crowdnav_amd
    from crowdnav_amd.compat import cadrl
    from crowdnav_amd.compat.sarl import build_action_space
    torch.manual_seed(40 + humans)
    net = cadrl.ValueNetwork(13, [150, 100, 100, 1])
    B = 203
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=1, circle_radius=4.0 if humans <= 5 else 6.0)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='cadrl', mlp3_dims=(150, 100, 100, 1))
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81 * humans, 13)).reshape(B * 81, humans).min(dim=1).values.reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0].reshape(B, 81) - want).max() <= 2e-5 and np.abs(got['0'][0].reshape(B, 81) - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 1e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize('humans,with_om', [(1, False), (3, False), (5, False), (5, True), (7, False), (7, True)])
def test_register_resident_lstm_rl_value_network(humans, with_om, monkeypatch):
    """lstm_reg_kernel (lstm_rl.ValueNetwork1: gate layer and cell update in registers, 5 tiles side by side per wave, any number
    of humans) against the torch module and the LDS kernels: 5 569 tiles = one full round of 5-tile bundles on the 1024 waves,
    then single tiles."""
    import crowdnav_amd
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import build_action_space
    torch.manual_seed(60 + humans)
    d = 61 if with_om else 13
    net = lstm_rl.ValueNetwork1(d, 6, [150, 100, 100, 1], 50)
    B = 1100
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=1)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='lstm_rl', mlp1_dims=(50, 1),
                           mlp3_dims=(150, 100, 100, 1), with_om=with_om)
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0] - want).max() <= 2e-5 and np.abs(got['0'][0] - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 2e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize('humans,with_om', [(1, False), (5, False), (5, True), (7, False)])
def test_register_resident_lstm_rl_pairwise_value_network(humans, with_om, monkeypatch):
    """lstm2_reg_kernel (lstm_rl.ValueNetwork2, lstm_rl.py:36-66: mlp1 on every human's row in front of the cell, all of it in
    registers, 3 tiles side by side per wave) against the torch module and the LDS kernels: 5 569 tiles = one full round of
    3-tile bundles on the 1024 waves, a partial round, then single tiles."""
    import crowdnav_amd
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import build_action_space
    torch.manual_seed(80 + humans)
    d = 61 if with_om else 13
    net = lstm_rl.ValueNetwork2(d, 6, [150, 100, 100, 50], [150, 100, 100, 1], 50)
    B = 1100
    space, _, _ = build_action_space(1.0)
    got = {}
    for reg in ('1', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=1)
        eng.reset(3000 + np.arange(B))
        eng.step(np.zeros((B, 2)), update=True)
        eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='lstm_rl', mlp1_dims=(50, 1),
                           mlp3_dims=(150, 100, 100, 1), with_om=with_om, interaction_dims=(150, 100, 100, 50))
        eng.sarl_set_weights(net.state_dict())
        out = eng.sarl_select()
        got[reg] = (eng.sarl_export('V').cpu().numpy(), out['best'].cpu().numpy(), eng.sarl_export('X').cpu())
    with torch.no_grad():
        want = net(got['1'][2].reshape(B * 81, humans, d)).reshape(B, 81).numpy()
    assert torch.equal(got['1'][2], got['0'][2])
    assert np.abs(got['1'][0] - want).max() <= 2e-5 and np.abs(got['0'][0] - want).max() <= 2e-5
    assert np.abs(got['1'][0] - got['0'][0]).max() <= 2e-6
    assert (got['1'][1] == got['0'][1]).mean() > 0.99


@pytest.mark.gpu
def test_register_resident_lstm_rl_pairwise_on_the_reference_fixture_and_under_the_mixed_rule(monkeypatch):
    """Forced onto a small batch (CROWDNAV_AMD_SARL_REG=2), lstm2_reg_kernel reproduces the network outputs of the unmodified
    reference's LstmRL with the interaction module (lstm_rl2_om.npz), and stops an episode's recurrence at its last present
    human under the `mixed` rule like the LDS kernel does."""
    import crowdnav_amd
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import build_action_space
    monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')
    g = load_golden('lstm_rl2_om.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, model='lstm_rl', with_om=True, mlp1_dims=(50, 1),
                       mlp3_dims=(150, 100, 100, 1), interaction_dims=(150, 100, 100, 50))
    eng.sarl_set_weights(_lstm2_mirror(g).state_dict())
    out = eng.sarl_select()
    assert np.abs(eng.sarl_export('V').cpu().numpy() - g['net_out']).max() <= 2e-6
    assert np.abs(out['values'].cpu().numpy() - g['values']).max() <= 2e-6
    torch.manual_seed(12)
    net = lstm_rl.ValueNetwork2(13, 6, [150, 100, 100, 50], [150, 100, 100, 1], 50)
    space, _, _ = build_action_space(1.0)
    B, got = 64, {}
    for reg in ('2', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        e2 = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1,
                                          scenario_rule=crowdnav_amd.MIXED)
        e2.reset(1000 + np.arange(B))
        e2.step(np.zeros((B, 2)), update=True)
        e2.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), model='lstm_rl', mlp1_dims=(50, 1),
                          mlp3_dims=(150, 100, 100, 1), interaction_dims=(150, 100, 100, 50))
        e2.sarl_set_weights(net.state_dict())
        e2.sarl_select()
        got[reg] = e2.sarl_export('V').cpu().numpy()
        counts = e2.human_count().cpu().numpy()
    assert len(set(counts.tolist())) >= 3 and counts.min() < 5
    assert np.abs(got['2'] - got['0']).max() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('model,humans', [('sarl', 5), ('sarl', 3), ('lstm_rl', 5)])
def test_values_of_caller_held_states_vs_torch(model, humans):
    """cn_sarl_values (ABI v11): V of joint states the caller holds — replay-memory rows [n, H, 13] — under the engine's weights,
    by the narrow-tile network kernel reading the rows where they lie: against the torch module on the same rows (what
    target_model(next_states) is to explorer.py:113-116), for every n from one state to the engine's capacity; and the refusals."""
    import crowdnav_amd
    from crowdnav_amd.compat import lstm_rl
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(90 + humans)
    if model == 'sarl':
        net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
        kw = {}
    else:
        net = lstm_rl.ValueNetwork1(13, 6, [150, 100, 100, 1], 50)
        kw = dict(model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    space, _, _ = build_action_space(1.0)
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=2, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), **kw)  # (never reset: the env state is not read)
    eng.sarl_set_weights(net.state_dict())
    # rows a replay memory would hold: the transform of real states
    src = crowdnav_amd.BatchedCrowdSim(num_envs=162, num_humans=humans, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    src.reset(500 + np.arange(162))
    src.step(np.zeros((162, 2)), update=True)
    src.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), **kw)
    rows = src.sarl_transform(sort_humans=(model == 'lstm_rl')).contiguous()
    assert tuple(rows.shape) == (162, humans, 13)
    dev_net = net.to(rows.device)
    with torch.no_grad():
        want = dev_net(rows).reshape(-1)
    for n in (1, 2, 5, 53, 54, 161, 162):
        got = eng.sarl_values(rows[:n].contiguous())
        eng.sync()
        assert float((got - want[:n]).abs().max()) <= 2e-6, n
    with pytest.raises(crowdnav_amd.CrowdNavAmdError):
        eng.sarl_values(torch.cat([rows, rows[:1]]))  # more states than the engine's tiles hold
    # the decision's own network kernel gives the same bits for the same rows (same kernel, rows built in LDS instead)
    before = eng.launch_counts()
    eng.sarl_values(rows[:54].contiguous())
    after = eng.launch_counts()
    assert after['sarl_narrow'] - before['sarl_narrow'] == 1


@pytest.mark.gpu
@pytest.mark.parametrize('with_om', [False, True])
def test_value_network_at_the_full_benchmark_size_vs_torch(with_om):
    """BASELINE configs[2] at full size: 4096 envs x 81 actions x 5 humans = 20 736 tiles through the register-resident kernel
    (20 full rounds of the 1024 persistent waves + a quarter round) — every one of the 331 776 network outputs against the
    torch fp32 module on the same device, and the decision's arg-max against the combined values."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    torch.manual_seed(7)
    d = 61 if with_om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 4096
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.reset(2000 + np.arange(B))
    eng.step(np.zeros((B, 2)), update=True)
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om)
    eng.sarl_set_weights(net.state_dict())
    out = eng.sarl_select()
    X, V = eng.sarl_export('X'), eng.sarl_export('V')
    dev_net = net.to(X.device)
    with torch.no_grad():
        want = torch.cat([dev_net(X.reshape(B * 81, 5, d)[i:i + 65536]) for i in range(0, B * 81, 65536)]).reshape(B, 81)
    assert float((V.reshape(B, 81) - want).abs().max()) <= 2e-5
    values = out['values']
    assert torch.equal(out['best'].long(), values.argmax(dim=1)) or \
        float((values.gather(1, out['best'].long()[:, None])[:, 0] - values.max(dim=1).values).abs().max()) == 0.0


@pytest.mark.gpu
def test_register_resident_value_network_on_the_reference_fixtures_and_under_the_mixed_rule(monkeypatch):
    """Small batches run the LDS kernel by default; forced (CROWDNAV_AMD_SARL_REG=2) the register-resident kernel reproduces the
    reference fixture's network outputs, and masks an episode's absent humans under the `mixed` rule exactly like the LDS
    kernel does (attention mean and softmax over the humans present)."""
    import crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', '2')
    for name in ('sarl_plain.npz', 'sarl_om.npz'):
        g = load_golden(name)
        n = len(g['states'])
        eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                           robot_visible=int(g['robot_visible']))
        eng.set_state(g['states'], g['gtime'])
        eng.sarl_configure(actions=g['action_space'], gamma=0.9, with_om=bool(int(g['with_om'])))
        eng.sarl_set_weights(_mirror(g).state_dict())
        out = eng.sarl_select()
        assert np.abs(eng.sarl_export('V').cpu().numpy() - g['net_out']).max() <= 1e-6
        assert np.abs(out['values'].cpu().numpy() - g['values']).max() <= 1e-6
    # mixed rule: 1..5 humans per episode, the absent ones parked and masked
    torch.manual_seed(11)
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    space, _, _ = build_action_space(1.0)
    B, got = 64, {}
    for reg in ('2', '0'):
        monkeypatch.setenv('CROWDNAV_AMD_SARL_REG', reg)
        e2 = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1,
                                          scenario_rule=crowdnav_amd.MIXED)
        e2.reset(1000 + np.arange(B))
        e2.step(np.zeros((B, 2)), update=True)
        e2.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]))
        e2.sarl_set_weights(net.state_dict())
        e2.sarl_select()
        got[reg] = e2.sarl_export('V').cpu().numpy()
        counts = e2.human_count().cpu().numpy()
    assert len(set(counts.tolist())) >= 3 and counts.min() < 5
    assert np.abs(got['2'] - got['0']).max() <= 1e-6


def _joint_rows(g, d, a):
    """float32 joint rows [propagate(self, action a) | next human h] of decision d, as MultiHumanRL.predict builds them."""
    s, act = g['states'][d], g['action_space'][a]
    me = [s[0, 0] + act[0] * 0.25, s[0, 1] + act[1] * 0.25, act[0], act[1], s[0, 6], s[0, 4], s[0, 5], s[0, 7], np.pi / 2]
    return torch.cat([torch.Tensor([tuple(me) + tuple(h)]) for h in g['next_obs'][d].tolist()], dim=0)


@pytest.mark.parametrize('name', FIXTURES)
def test_host_rotate_and_occupancy_maps_match_reference_cpu(name):
    """The host-side feature builders used by SARL.transform (replay memory) vs the reference's own outputs."""
    from crowdnav_amd.compat.sarl import occupancy_maps, rotate
    from crowdnav_amd.compat.types import ObservableState
    g = load_golden(name)
    for d in (0, 5, 17):
        for a in (0, 1, 40, 80):
            assert np.abs(rotate(_joint_rows(g, d, a)).numpy() - g['inputs'][d, a, :, :13]).max() <= 1e-6
        if int(g['with_om']):
            hs = [ObservableState(*row) for row in g['next_obs'][d].tolist()]
            assert np.abs(occupancy_maps(hs, 4, 1.0, 3).numpy() - g['inputs'][d, 0, :, 13:]).max() <= 1e-6


def _sarl_setup(g):
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    with_om, visible = bool(int(g['with_om'])), bool(int(g['robot_visible']))
    cfg = c.default_env_config({('robot', 'visible'): 'true' if visible else 'false'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    policy = c.policy_factory['sarl']()
    policy.configure(default_policy_config({('sarl', 'with_om'): 'true' if with_om else 'false'}))
    policy.get_model().load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items()
                                        if k.startswith('param_')})
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_phase('test')
    policy.set_device(torch.device('cpu'))
    policy.set_env(env)
    return c, env, robot, policy


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_gym_surface_with_sarl_policy_follows_reference_episode(name):
    """env.reset / robot.act (SARL.predict on device) / env.step exactly as the reference loop, first fixture episode."""
    g = load_golden(name)
    c, env, robot, policy = _sarl_setup(g)
    case = 3 if int(g['with_om']) else 0
    ob = env.reset('test', case)
    env._eng.set_state(g['states'][:1], np.zeros(1))  # the reference's exact initial state
    env._pull()
    ob = [h.get_observable_state() for h in env.humans]
    for d in range(8):
        assert np.array_equal(np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans]), g['states'][d][1:, :4])
        action = robot.act(ob)
        assert (action.vx, action.vy) == tuple(g['action'][d])
        assert np.abs(np.array(policy.action_values) - g['values'][d]).max() <= 1e-6
        ob, reward, done, info = env.step(action)
        assert reward == g['rewards'][d][g['best'][d]]
        if done:
            break


@pytest.mark.gpu
def test_explorer_batched_sarl_equals_sequential():
    g = load_golden('sarl_plain.npz')
    c, env, robot, policy = _sarl_setup(g)
    ex = c.Explorer(env, robot, 'cpu', gamma=0.9)
    ex.run_k_episodes(6, 'val')  # batched: select + step + masked reset on 6 envs
    batched, outcome, steps = dict(ex.last_stats), list(ex.last_batch['outcome']), list(ex.last_batch['steps'])
    env.case_counter['val'] = 0
    stats = ex._run_sequential(6, 'val', False, False)
    ex._report(6, 'val', None, False, *stats)
    for key in ('success_rate', 'collision_rate', 'too_close', 'collision_cases', 'timeout_cases'):
        assert ex.last_stats[key] == batched[key], key
    assert abs(ex.last_stats['total_reward'] - batched['total_reward']) < 1e-6
    assert len(outcome) == 6 and all(o in (2, 3, 4) for o in outcome) and min(steps) >= 1


def test_replay_memory_and_trainer_cpu():
    from crowdnav_amd.compat.sarl import ValueNetwork
    from crowdnav_amd.compat.trainer import ReplayMemory, Trainer
    torch.manual_seed(0)
    mem = ReplayMemory(8)
    for i in range(11):  # wraps: capacity 8
        mem.push((torch.randn(5, 13), torch.tensor([0.1 * i])))
    assert len(mem) == 8 and mem.is_full() and mem.position == 3 and float(mem[0][1]) == pytest.approx(0.8)
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    tr = Trainer(net, mem, torch.device('cpu'), batch_size=4)
    with pytest.raises(ValueError):
        tr.optimize_batch(1)
    tr.set_learning_rate(0.01)
    first = tr.optimize_epoch(1)
    later = tr.optimize_epoch(20)
    assert later < first and tr.optimize_batch(3) >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('policy', ['sarl', 'cadrl', 'lstm_rl'])
def test_train_schedule_smoke_config5(policy):
    """The reference's train.py schedule end to end on tiny sizes: IL from device ORCA demonstrations, RL with
    epsilon-greedy device SARL decisions, torch trainer, batched val/test evaluation."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = mod.parser().parse_args(['--il-episodes', '6', '--il-epochs', '3', '--train-episodes', '3', '--train-batches',
                                    '4', '--evaluation-interval', '2', '--val-size', '4', '--test-size', '4',
                                    '--target-update-interval', '2', '--batch-size', '16', '--policy', policy])
    out = mod.run(args)
    assert out['memory'] > 50 and out['il_loss'] is not None and out['rl_loss'] is not None
    assert np.isfinite(out['il_loss']) and np.isfinite(out['rl_loss'])
    s = out['stats']
    assert 0.0 <= s['success_rate'] <= 1.0 and abs(s['success_rate'] + s['collision_rate'] - 1.0) <= 1.0


@pytest.mark.gpu
def test_train_checkpoints_and_resume(tmp_path):
    """train.py's weight files (il_model.pth after imitation learning, rl_model.pth every checkpoint_interval) and
    --resume (load rl_model.pth, 100 warm-up episodes at epsilon_end, write resumed_rl_model.pth), train.py:106-145."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    common = ['--il-episodes', '6', '--il-epochs', '2', '--train-episodes', '2', '--train-batches', '2',
              '--evaluation-interval', '5', '--val-size', '3', '--test-size', '3', '--checkpoint-interval', '1',
              '--batch-size', '16', '--sample-episodes', '8', '--seed', '0', '--output-dir', str(tmp_path)]
    first = mod.run(mod.parser().parse_args(common))
    assert first['il_loss'] is not None and (tmp_path / 'il_model.pth').exists() and (tmp_path / 'rl_model.pth').exists()
    again = mod.run(mod.parser().parse_args(common))          # il_model.pth is there: imitation learning is skipped
    assert again['il_loss'] is None and again['timing']['il_collect_s'] == 0.0
    resumed = mod.run(mod.parser().parse_args(common + ['--resume']))
    assert (tmp_path / 'resumed_rl_model.pth').exists() and resumed['il_loss'] is None
    assert resumed['timing']['rl_env_steps'] > 0  # (how many of those episodes end in the memory depends on the weights)
    state = torch.load(tmp_path / 'resumed_rl_model.pth', map_location='cpu')
    assert 'mlp1.0.weight' in state and 'mlp3.6.bias' in state  # the reference's state_dict keys


@pytest.mark.gpu
@pytest.mark.parametrize('name,with_om', [('sarl', False), ('sarl', True), ('cadrl', False), ('lstm_rl', True)])
def test_batched_imitation_collection_equals_sequential(name, with_om):
    """Explorer.run_k_episodes(k, 'train', update_memory=True, imitation_learning=True): the lock-step batched
    collection fills the replay memory with the same (state, value) pairs, in the same order, as the reference loop."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    from crowdnav_amd.compat.trainer import ReplayMemory

    def collect(force_sequential):
        cfg = c.default_env_config()
        env = c.CrowdSim()
        env.configure(cfg)
        robot = c.Robot(cfg, 'robot')
        target = c.policy_factory[name]()
        target.configure(default_policy_config({(name, 'with_om'): 'true' if with_om else 'false'} if name != 'cadrl'
                                               else None))
        target.set_device(torch.device('cpu'))
        il = c.policy_factory['orca']()
        il.multiagent_training, il.safety_space = target.multiagent_training, 0.15
        robot.set_policy(il)
        env.set_robot(robot)
        mem = ReplayMemory(100000)
        ex = c.Explorer(env, robot, torch.device('cpu'), mem, 0.9, target_policy=target)
        if force_sequential:
            robot.policy.set_phase('train')
            stats = ex._run_sequential(7, 'train', True, True)
            ex._report(7, 'train', None, False, *stats)
        else:
            ex.run_k_episodes(7, 'train', update_memory=True, imitation_learning=True)
        return mem, dict(ex.last_stats), env.case_counter['train']

    mem_b, stats_b, cc_b = collect(False)
    mem_s, stats_s, cc_s = collect(True)
    assert cc_b == cc_s == 7 and len(mem_b) == len(mem_s) > 50
    for key in ('success_rate', 'collision_rate', 'too_close', 'collision_cases', 'timeout_cases'):
        assert stats_b[key] == stats_s[key], key
    xb = torch.stack([m[0] for m in mem_b.memory])
    xs = torch.stack([m[0] for m in mem_s.memory])
    vb = torch.cat([m[1] for m in mem_b.memory])
    vs = torch.cat([m[1] for m in mem_s.memory])
    assert xb.shape == xs.shape and (xb - xs).abs().max() <= 5e-6  # torch vectorised vs per-state atan2/cos/sin: 1 ulp at |x| ~ 8
    assert torch.equal(vb, vs)


def _cadrl_mirror(g):
    from crowdnav_amd.compat.cadrl import ValueNetwork
    net = ValueNetwork(13, [150, 100, 100, 1])
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


def test_cadrl_value_network_mirror_cpu():
    g = load_golden('cadrl_plain.npz')
    net = _cadrl_mirror(g)
    x = torch.from_numpy(g['inputs'])  # [decisions, 81, H, 13]
    with torch.no_grad():
        out = net(x).squeeze(-1).min(dim=2)[0].numpy()  # min over humans (cadrl.py:162)
    assert np.abs(out - g['net_out']).max() <= 2e-6


@pytest.mark.gpu
def test_cadrl_select_vs_reference():
    """§8(f): the CADRL head on the same device pipeline, vs the unmodified reference CADRL.predict."""
    import crowdnav_amd
    g = load_golden('cadrl_plain.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, model='cadrl', mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_cadrl_mirror(g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.array_equal(cpu(eng.sarl_export('reward')), g['rewards'])
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 2 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])
    assert np.array_equal(cpu(out['action'])[clear], g['action'][clear])


@pytest.mark.gpu
def test_policy_weights_are_reuploaded_only_when_they_change():
    g = load_golden('sarl_plain.npz')
    c, env, robot, policy = _sarl_setup(g)
    ob = env.reset('test', 0)
    a0 = robot.act(ob)
    stamp = env._eng._sarl_weights_stamp
    assert robot.act(ob) == a0 and env._eng._sarl_weights_stamp == stamp  # nothing changed: no re-upload
    with torch.no_grad():
        for p in policy.model.parameters():
            p.mul_(-1.0)  # what an optimizer step does: in-place update
    robot.act(ob)
    assert env._eng._sarl_weights_stamp != stamp
    vals_flipped = np.array(policy.action_values)
    policy.model.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    assert robot.act(ob) == a0 and not np.allclose(vals_flipped, np.array(policy.action_values))


def _lstm_mirror(g):
    from crowdnav_amd.compat.lstm_rl import ValueNetwork1
    in_dim = 13 + (48 if int(g['with_om']) else 0)
    net = ValueNetwork1(in_dim, 6, [150, 100, 100, 1], 50)
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


def test_lstm_rl_value_network_mirror_cpu():
    g = load_golden('lstm_rl_om.npz')
    net = _lstm_mirror(g)
    x = torch.from_numpy(g['inputs'])
    d, k, h, f = x.shape
    with torch.no_grad():
        out = net(x.reshape(d * k, h, f)).reshape(d, k).numpy()
    assert np.abs(out - g['net_out']).max() <= 2e-6


@pytest.mark.gpu
def test_lstm_rl_select_vs_reference():
    """§8(f): the LSTM-RL head (OM-LSTM-RL: occupancy maps on) on the device pipeline vs the reference's LstmRL.predict."""
    import crowdnav_amd
    g = load_golden('lstm_rl_om.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, model='lstm_rl', with_om=True, mlp1_dims=(50, 1),
                       mlp3_dims=(150, 100, 100, 1))
    eng.sarl_set_weights(_lstm_mirror(g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.array_equal(cpu(eng.sarl_export('reward')), g['rewards'])
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 4 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])


def _lstm2_mirror(g):
    from crowdnav_amd.compat.lstm_rl import ValueNetwork2
    in_dim = 13 + (48 if int(g['with_om']) else 0)
    net = ValueNetwork2(in_dim, 6, [150, 100, 100, 50], [150, 100, 100, 1], 50)
    net.load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param_')})
    return net


def test_lstm_rl_pairwise_value_network_mirror_cpu():
    """lstm_rl.ValueNetwork2 (with_interaction_module = true): the torch mirror reproduces the reference's outputs."""
    g = load_golden('lstm_rl2_om.npz')
    net = _lstm2_mirror(g)
    x = torch.from_numpy(g['inputs'])
    d, k, h, f = x.shape
    with torch.no_grad():
        out = net(x.reshape(d * k, h, f)).reshape(d, k).numpy()
    assert np.abs(out - g['net_out']).max() <= 2e-6


@pytest.mark.gpu
def test_lstm_rl_pairwise_select_vs_reference():
    """LSTM-RL with the pairwise interaction module (ValueNetwork2: mlp1 per human in front of the LSTM) on the device
    pipeline vs the unmodified reference's LstmRL.predict."""
    import crowdnav_amd
    g = load_golden('lstm_rl2_om.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.set_state(g['states'], g['gtime'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9, model='lstm_rl', with_om=True, mlp1_dims=(50, 1),
                       mlp3_dims=(150, 100, 100, 1), interaction_dims=(150, 100, 100, 50))
    eng.sarl_set_weights(_lstm2_mirror(g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.array_equal(cpu(eng.sarl_export('reward')), g['rewards'])
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 4 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])


@pytest.mark.gpu
def test_sarl_unicycle_select_vs_reference():
    """SARL with [action_space] kinematics = unicycle (ActionRot table, propagate / rotate with the heading feature,
    lookahead rewards through the unicycle collision test) vs the unmodified reference."""
    import crowdnav_amd
    g = load_golden('sarl_unicycle.npz')
    n = len(g['states'])
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL,
                                       robot_visible=1, robot_kinematics=crowdnav_amd.UNICYCLE)
    eng.set_state(g['states'], g['gtime'])
    eng.set_theta(g['theta'])
    eng.sarl_configure(actions=g['action_space'], gamma=0.9)
    eng.sarl_set_weights(_mirror(g).state_dict())
    out = eng.sarl_select()
    eng.sync()
    cpu = lambda t: t.cpu().numpy()  # noqa: E731
    assert np.abs(cpu(eng.sarl_export('reward')) - g['rewards']).max() <= 1e-12
    assert np.abs(cpu(eng.sarl_export('X')) - g['inputs']).max() <= 5e-6
    assert np.abs(cpu(eng.sarl_export('V')) - g['net_out']).max() <= 1e-6
    assert np.abs(cpu(out['values']) - g['values']).max() <= 1e-6
    top2 = np.sort(g['values'], axis=1)[:, -2:]
    import os as _os; report_argmax(_os.environ.get('PYTEST_CURRENT_TEST', ''), cpu(out['best']), g['best'], g['values'])
    clear = (top2[:, 1] - top2[:, 0]) > 4e-6
    assert clear.sum() >= n // 4 and np.array_equal(cpu(out['best'])[clear], g['best'][clear])
    assert np.array_equal(cpu(out['action'])[clear], g['action'][clear])


def test_unicycle_action_space_and_rotate_cpu():
    from crowdnav_amd.compat.sarl import build_action_space, rotate
    g = load_golden('sarl_unicycle.npz')
    space, _, rotations = build_action_space(1.0, 5, 16, 'unicycle')
    assert np.array_equal(np.array([list(a) for a in space]), g['action_space'])
    assert rotations[0] == -np.pi / 4 and rotations[-1] == np.pi / 4
    # joint rows of decision 0 / action 7 rebuilt the way MultiHumanRL.predict does, then the host rotate
    s, th, (v, r) = g['states'][0], g['theta'][0], g['action_space'][7]
    nth = th + r
    me = [s[0, 0] + v * np.cos(nth) * 0.25, s[0, 1] + v * np.sin(nth) * 0.25, v * np.cos(nth), v * np.sin(nth), s[0, 6],
          s[0, 4], s[0, 5], s[0, 7], nth]
    rows = torch.cat([torch.Tensor([tuple(me) + tuple(h)]) for h in g['next_obs'][0].tolist()], dim=0)
    assert np.abs(rotate(rows, 'unicycle').numpy() - g['inputs'][0, 7]).max() <= 2e-6


@pytest.mark.gpu
def test_gym_surface_with_unicycle_sarl_follows_reference_episode():
    """[action_space] kinematics = unicycle end to end on the reference's surface: ActionRot actions, robot heading."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    g = load_golden('sarl_unicycle.npz')
    cfg = c.default_env_config({('robot', 'visible'): 'true'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    policy = c.policy_factory['sarl']()
    policy.configure(default_policy_config({('action_space', 'kinematics'): 'unicycle'}))
    policy.get_model().load_state_dict({k[len('param_'):]: torch.from_numpy(v) for k, v in g.items()
                                        if k.startswith('param_')})
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_phase('test')
    policy.set_device(torch.device('cpu'))
    policy.set_env(env)
    assert robot.kinematics == 'unicycle'
    ob = env.reset('test', 10)
    env._eng.set_state(g['states'][:1], np.zeros(1))
    env._pull()
    ob = [h.get_observable_state() for h in env.humans]
    assert robot.theta == np.pi / 2
    for d in range(10):
        action = robot.act(ob)
        assert isinstance(action, c.ActionRot) and (action.v, action.r) == tuple(g['action'][d])
        ob, reward, done, info = env.step(action)
        assert abs(reward - g['step_reward'][d]) <= 1e-12 and done == bool(g['step_done'][d])
        assert abs(robot.theta - g['next_theta'][d]) <= 1e-12
        assert abs(robot.px - g['next_states'][d][0, 0]) <= 1e-12 and abs(robot.vy - g['next_states'][d][0, 3]) <= 1e-12
        if done:
            break


def test_batched_td_targets_equal_per_step_forward_cpu():
    """Explorer.update_memory (RL branch, explorer.py:107-113): one batched target-network forward for all next states
    gives the per-step values of the reference loop."""
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import ValueNetwork
    from crowdnav_amd.compat.trainer import ReplayMemory
    torch.manual_seed(1)
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    cfg = c.default_env_config()
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    robot.time_step = 0.25
    mem = ReplayMemory(100)
    ex = c.Explorer(env, robot, torch.device('cpu'), mem, 0.9)
    ex.update_target_model(net)
    states = [torch.randn(5, 13) for _ in range(9)]
    rewards = [0.0, -0.01, 0.0, 0.0, -0.02, 0.0, 0.0, 0.0, 1.0]
    ex.update_memory(states, [None] * 9, rewards, imitation_learning=False)
    assert len(mem) == 9
    gamma_bar = pow(0.9, 0.25 * robot.v_pref)
    for i in range(9):
        want = rewards[i] if i == 8 else rewards[i] + gamma_bar * ex.target_model(states[i + 1].unsqueeze(0)).data.item()
        assert torch.equal(mem[i][0], states[i]) and abs(float(mem[i][1]) - want) <= 1e-6
