"""Where do multi-millisecond GPU idle gaps appear in a long asynchronous stream of cn_sarl_select launches?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
want = 'values' in sys.argv
eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=5, robot_policy=0, robot_visible=1)
eng.reset(2000 + np.arange(4096)); eng.step(np.zeros((4096, 2)), update=True)
net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
space, _, _ = build_action_space(1.0)
eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space])); eng.sarl_set_weights(net.state_dict())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    eng.sarl_select(want_values=want)
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
host = np.array(host) * 1e3
print('median decision %.3f ms; decisions over 3 ms: %s' % (np.median(d), [(int(i), round(float(d[i]), 1)) for i in np.nonzero(d > 3)[0]]))
print('host enqueue median %.3f ms; over 3 ms: %s' % (np.median(host), [(int(i), round(float(host[i]), 1)) for i in np.nonzero(host > 3)[0]]))
