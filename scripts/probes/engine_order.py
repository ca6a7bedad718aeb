"""Does cn_sarl_select slow down on the second engine of a process?  usage: engine_order.py [keep] [om-first]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
keep = 'keep' in sys.argv
seq = [1, 0, 0, 1, 0] if 'om-first' in sys.argv else [0, 0, 1, 0, 0]
alive = []
space, _, _ = build_action_space(1.0)
for om in seq:
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=5, robot_policy=0, robot_visible=1)
    eng.reset(2000 + np.arange(4096)); eng.step(np.zeros((4096, 2)), update=True)
    torch.manual_seed(0)
    net = ValueNetwork(61 if om else 13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=bool(om)); eng.sarl_set_weights(net.state_dict())
    for _ in range(3): eng.sarl_select(want_values=False)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(20): eng.sarl_select(want_values=False)
    eng.sync(); dt = (time.perf_counter() - t0) / 20
    print('om', om, 'select ms %.3f' % (dt * 1e3), 'mem GB %.2f' % (torch.cuda.mem_get_info()[0] / 2**30), flush=True)
    if keep: alive.append(eng)
    else:
        del eng; torch.cuda.empty_cache()
