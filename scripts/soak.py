"""Long fused rollout: 1 M batched steps of 4096 envs (4.1 G transitions): state stays finite, bookkeeping adds up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, crowdnav_amd
B = 4096
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1)
bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=4)
t0 = time.perf_counter()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for _ in range(steps // 1000):
    eng.rollout(1000)
eng.sync(); dt = time.perf_counter() - t0
s, g = eng.get_state()
tr = int(bufs['transitions'].item())
print('steps', steps, 'transitions', tr, 'paused', B * steps - tr, 'episodes', int(bufs['ep_count'].sum().item()),
      'finite', bool(torch.isfinite(s).all().item()), 'max |pos|', float(s[:, :, :2].abs().max().item()),
      'rate M/s', round(tr / dt / 1e6, 1), 'outcomes', torch.bincount(bufs['ep_outcome'].flatten().long(), minlength=5).tolist())
