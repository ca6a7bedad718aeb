"""Times seeded scenario generation (cn_reset) for H=5 and H=20 (reference defaults, R = 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, crowdnav_amd
for H, B in ((5, 4096), (20, 256), (20, 1024), (20, 4096)):
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H)
    eng.reset(1000 + np.arange(B)); torch.cuda.synchronize()
    t0 = time.perf_counter(); d = eng.reset(5000 + np.arange(B)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    d = d.cpu().numpy()
    print('H', H, 'B', B, 'reset ms', round(dt * 1e3, 2), 'draws mean', int(d.mean()), 'max', int(d.max()), flush=True)
