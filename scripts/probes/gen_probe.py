"""Scenario generator alone (cn_reset, 20 humans, 4 m circle): time per pass of 64 attempts for ONE wave on an idle chip (the
dependent chain) and for many waves working on the same hard seed (the chip's attempt throughput).  Seeds by the oracle's
draw counts: 1793 = 3.15 M attempts, 1066 = 614 k, 1109 = 204 k, 1472 = 104 k; median 1.7 k.

    python scripts/probes/gen_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import crowdnav_amd


def run(B, seed, reps=3):
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=20, robot_visible=1)
    seeds = np.full(B, seed) if np.isscalar(seed) else np.asarray(seed)
    best, draws = None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        draws = eng.reset(seeds)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    eng.close()
    d = np.asarray(draws.cpu() if hasattr(draws, 'cpu') else draws, dtype=np.float64)
    return best, d


for seed in (1109, 1472, 1066):
    dt, d = run(1, seed)
    passes = d[0] / 3 / 64
    print('one wave, seed %d: %8.0f attempts  %8.2f ms  %6.3f us per pass of 64' % (seed, d[0] / 3, dt * 1e3, dt * 1e6 / passes))
for B in (256, 1024, 2048, 4096, 8192, 16384):
    dt, d = run(B, 1109)
    passes = d.sum() / 3 / 64
    print('%5d waves on seed 1109: %7.2f ms  %7.1f passes/us on the chip  (%.3f us per pass per wave-slot of 1024 SIMDs)'
          % (B, dt * 1e3, passes / dt / 1e6, dt * 1e6 / (passes / min(B, 1024 * 3))))
dt, d = run(4096, 1000 + np.arange(4096) % 1021)
print('4096 scenarios of the 1021 bench seeds: %.1f ms, %.0f attempts, %.1f M attempts/s' % (dt * 1e3, d.sum() / 3, d.sum() / 3 / dt / 1e6))
