"""Per-launch cost of cn_rollout as a function of the steps fused into one launch (steady state: desynchronised
episodes).  HIP-event time around each call, min / median over repeats; the intercept of the fit is the fixed cost per
launch, the slope the per-step cost of the slowest wave.  Usage: python scripts/launch_probe.py [--envs 4096 --humans 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--humans', type=int, default=5)
ap.add_argument('--repeats', type=int, default=40)
ap.add_argument('--circle-radius', type=float, default=4.0)
args = ap.parse_args()
eng = crowdnav_amd.BatchedCrowdSim(num_envs=args.envs, num_humans=args.humans, robot_policy=crowdnav_amd.ROBOT_ORCA,
                                   robot_visible=1, circle_radius=args.circle_radius)
bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, episode_limit=-1, record_capacity=4)
eng.rollout(300)
torch.cuda.synchronize()
xs, ys = [], []
for n in (1, 2, 5, 10, 20, 40, 100, 400):
    ev = []
    for _ in range(args.repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.rollout(n)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    d = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    print('steps %4d: min %8.1f us  median %8.1f us  max %8.1f us   median per step %6.2f us' %
          (n, d.min(), np.median(d), d.max(), np.median(d) / n))
    xs.append(n)
    ys.append(np.median(d) if n > 48 else d.min())
slope, intercept = np.polyfit(xs[:6], ys[:6], 1)
print('fit over 1..40 steps (launches without a ring fill): %.1f us fixed + %.2f us per step' % (intercept, slope))
