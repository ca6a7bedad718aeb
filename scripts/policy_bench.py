"""Times cn_sarl_select for CADRL / LSTM-RL at 4096 envs x 81 actions; usage: policy_bench.py --policy cadrl|lstm_rl|lstm_rl2 --humans 5"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd
from crowdnav_amd.compat import cadrl, lstm_rl
from crowdnav_amd.compat.sarl import build_action_space
ap = argparse.ArgumentParser()
ap.add_argument('--policy', default='cadrl'); ap.add_argument('--humans', type=int, default=5)
ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
B, H = args.envs, args.humans
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=0, robot_visible=1)
eng.reset(2000 + np.arange(B)); eng.step(np.zeros((B, 2)), update=True)
torch.manual_seed(0)
space, _, _ = build_action_space(1.0)
acts = np.array([[a.vx, a.vy] for a in space])
head = 150 * 100 + 100 * 100 + 100
if args.policy == 'cadrl':
    net = cadrl.ValueNetwork(13, [150, 100, 100, 1])
    eng.sarl_configure(actions=acts, model='cadrl', mlp3_dims=(150, 100, 100, 1))
    flop = 2 * 81 * H * (13 * 150 + head) * B
elif args.policy == 'lstm_rl2':  # with_interaction_module = true (lstm_rl.ValueNetwork2)
    net = lstm_rl.ValueNetwork2(13, 6, [150, 100, 100, 50], [150, 100, 100, 1], 50)
    eng.sarl_configure(actions=acts, model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1), interaction_dims=(150, 100, 100, 50))
    flop = 2 * 81 * (H * (13 * 150 + 150 * 100 + 100 * 100 + 100 * 50 + 200 * (50 + 50)) + 56 * 150 + head) * B
else:
    net = lstm_rl.ValueNetwork1(13, 6, [150, 100, 100, 1], 50)
    eng.sarl_configure(actions=acts, model='lstm_rl', mlp1_dims=(50, 1), mlp3_dims=(150, 100, 100, 1))
    flop = 2 * 81 * (H * 200 * (13 + 50) + 56 * 150 + head) * B
eng.sarl_set_weights(net.state_dict())
for _ in range(3): eng.sarl_select(want_values=False)
eng.sync(); t0 = time.perf_counter()
for _ in range(args.iters): eng.sarl_select(want_values=False)
eng.sync(); dt = (time.perf_counter() - t0) / args.iters
print(args.policy, 'humans', H, 'select ms', round(dt * 1e3, 3), 'decisions/s', round(B / dt), 'TFLOP/s', round(flop / dt / 1e12, 2))
