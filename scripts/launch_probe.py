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
ap.add_argument('--idle-us', type=float, nargs='*', default=[0, 100, 400, 2000],
                help='also time 20- and 5-step launches issued after the device sat idle this long (what bench.py --steps 20 sees)')
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

import time  # noqa: E402
for idle in args.idle_us:
    for n in (5, 20):
        d = []
        for _ in range(args.repeats):
            torch.cuda.synchronize()
            t_end = time.perf_counter() + idle * 1e-6
            while time.perf_counter() < t_end:
                pass
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            eng.rollout(n)
            e1.record()
            torch.cuda.synchronize()
            d.append((e0.elapsed_time(e1) * 1e3, (time.perf_counter() - t0) * 1e6))
        d = np.array(d)
        print('idle %6.0f us, steps %3d: events min %7.1f median %7.1f us   host launch..sync min %7.1f median %7.1f us' %
              (idle, n, d[:, 0].min(), np.median(d[:, 0]), d[:, 1].min(), np.median(d[:, 1])))

# launches in which no episode ends (fresh episodes, <= 27 steps in): the per-step slope without the reset tail
d = []
for _ in range(12):
    eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, episode_limit=-1, record_capacity=4)
    eng.rollout(2)
    row = []
    for n in (5, 10, 10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.rollout(n)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3)
    d.append(row)
d = np.array(d)
print('fresh episodes (no resets): 5 steps min %.1f us, 10 steps min %.1f us, next 10 steps min %.1f us' % tuple(d.min(axis=0)))
