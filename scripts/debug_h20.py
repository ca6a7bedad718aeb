import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'oracle')
import numpy as np, torch
import crowdnav_amd, crowd_oracle
n = 256
cfg = dict(num_humans=20, robot_visible=1, circle_radius=12.0)
o = crowd_oracle.CrowdOracle(num_envs=n, robot_policy=1, **cfg)
o.reset(2000 + np.arange(n))
eng = crowdnav_amd.BatchedCrowdSim(num_envs=n, robot_policy=1, **cfg)
eng.set_state(o.get_state()[0], np.zeros(n))
for t in range(60):
    got = eng.step(None, update=True, want_obs=False); want = o.step(None, update=True)
    gv = got['orca_vel'].cpu().numpy(); wv = want['orca_vel']
    bad = np.argwhere(gv.view(np.uint32) != wv.view(np.uint32))
    sp = np.hypot(wv[..., 0], wv[..., 1]).max()
    if len(bad) or sp > 1.0001:
        print('step', t, 'mismatches', len(bad), 'oracle max speed', sp, 'first', bad[:3].tolist())
        if len(bad):
            b, a, _ = bad[0]; print(' got', gv[b, a], 'want', wv[b, a]); break
print('done; oracle max speed overall checked')
