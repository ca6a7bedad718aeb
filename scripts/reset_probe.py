"""Times seeded scenario generation (cn_reset) for H=20 at the reference defaults (R = 4) with an attempt cap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, crowdnav_amd
cap = sys.argv[1] if len(sys.argv) > 1 else '22'
os.environ['CROWDNAV_AMD_MAX_ATTEMPTS_LOG2'] = cap
for H, B in ((20, 4096),):
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H)
    t0 = time.perf_counter()
    try:
        d = eng.reset(5000 + np.arange(B)); err = 'no error'
    except crowdnav_amd.CrowdNavAmdError as e:
        err = 'gave up somewhere'; d = None
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('cap 2^%s H %d B %d reset ms %.1f  %s' % (cap, H, B, dt * 1e3, err), flush=True)
