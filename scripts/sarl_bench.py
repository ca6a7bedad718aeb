"""Times cn_sarl_select at 4096 envs x 5 humans x 81 actions (BASELINE configs[2]); --humans for other crowds."""
import argparse, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
ap = argparse.ArgumentParser(); ap.add_argument('--om', type=int, default=0); ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--humans', type=int, default=5); args = ap.parse_args()
B = args.envs
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=args.humans, robot_policy=0, robot_visible=1)
eng.reset(2000 + np.arange(B)); eng.step(np.zeros((B, 2)), update=True)
torch.manual_seed(0)
net = ValueNetwork(61 if args.om else 13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
space, _, _ = build_action_space(1.0)
eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=bool(args.om)); eng.sarl_set_weights(net.state_dict())
for _ in range(3): eng.sarl_select(want_values=False)
eng.sync(); t0 = time.perf_counter()
for _ in range(args.iters): out = eng.sarl_select(want_values=False)
eng.sync(); dt = (time.perf_counter() - t0) / args.iters
flop = 2 * (81 * args.humans * (62050 + (7200 if args.om else 0)) + 81 * 33500) * B
print('humans', args.humans, 'with_om', args.om, 'select ms', round(dt * 1e3, 3), 'decisions/s', round(B / dt), 'TFLOP/s', round(flop / dt / 1e12, 2))
