import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_golden
import crowdnav_amd as amd, crowd_oracle
from test_gpu_parity import TRAJ_FIXTURES, flat_steps
name='traj_visible_h20.npz'
g=load_golden(name); before,_,gtime=flat_steps(g); cfg=TRAJ_FIXTURES[name]
eng=amd.BatchedCrowdSim(num_envs=len(before), robot_policy=amd.ROBOT_ORCA, **cfg); eng.set_state(before,gtime)
got=eng.orca().cpu().numpy()
o=crowd_oracle.CrowdOracle(num_envs=len(before), robot_policy=1, **cfg); o.set_state(before,gtime); want=o.orca()
bad=np.argwhere((got.view(np.uint32)!=want.view(np.uint32)).any(axis=2))
print(len(bad),'of',got.shape[0]*got.shape[1]); print(bad[:30].tolist())
for e,a in bad[:6]: print(e,a,got[e,a],want[e,a])
