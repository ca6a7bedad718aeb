"""Prints the actual numerical gaps of the SARL pipeline vs the reference fixtures, and times cn_sarl_select."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from test_sarl import _mirror
import crowdnav_amd
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
for name in ('sarl_plain.npz', 'sarl_om.npz'):
    g = load_golden(name); n = len(g['states']); with_om = bool(int(g['with_om']))
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, num_humans=5, robot_policy=0, robot_visible=int(g['robot_visible']))
    eng.set_state(g['states'], g['gtime']); eng.sarl_configure(actions=g['action_space'], with_om=with_om)
    eng.sarl_set_weights(_mirror(g).state_dict()); out = eng.sarl_select(); eng.sync()
    c = lambda t: t.cpu().numpy()
    print(name, 'X', np.abs(c(eng.sarl_export('X')) - g['inputs']).max(), 'V', np.abs(c(eng.sarl_export('V')) - g['net_out']).max(),
          'values', np.abs(c(out['values']) - g['values']).max(), 'argmax agree', (c(out['best']) == g['best']).mean())
for with_om in (False, True):
    B = 4096
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=0, robot_visible=1)
    eng.reset(2000 + np.arange(B)); eng.step(np.zeros((B, 2)), update=True)
    torch.manual_seed(0)
    net = ValueNetwork(61 if with_om else 13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=with_om); eng.sarl_set_weights(net.state_dict())
    for _ in range(3): eng.sarl_select(want_values=False)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(10): out = eng.sarl_select(want_values=False)
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    flop = 2 * (81 * 5 * (62050 + (7200 if with_om else 0)) + 81 * 33500) * B
    print('with_om', with_om, 'select ms', dt * 1e3, 'decisions/s', B / dt, 'TFLOP/s', flop / dt / 1e12)
