"""Shader-clock ticks per phase of sarl_narrow_kernel (profiling build), summed over workgroups / launches -> per workgroup.

    CROWDNAV_AMD_LIB=build/exp/lib_timing.so python scripts/probes/narrow_probe.py [envs]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd  # noqa: E402
from crowdnav_amd import _lib  # noqa: E402
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space  # noqa: E402

NAMES = ['zero + first fetch', 'features', 'mlp1.0', 'mlp1.2', 'mean + mlp2.0', 'mlp2.2 + att0 global', 'att0 local', 'att.2',
         'att.4', 'softmax', 'joint', 'mlp3.0', 'mlp3.2', 'mlp3.4', 'head']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = _lib.load()
try:
    probe = lib.cn_debug_sarl_cycles
    probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int]
except AttributeError:  # the product library: host-clock figures only
    probe = lambda out, reset: 0  # noqa: E731
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=0)
eng.reset(2000 + np.arange(B))
torch.manual_seed(0)
net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
space, _, _ = build_action_space(1.0)
eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]))
eng.sarl_set_weights(net.state_dict())
for _ in range(20):
    eng.sarl_select(want_values=False)
eng.sync()
assert probe(None, 1) == 0
iters = 200
t0 = time.perf_counter()
for _ in range(iters):
    eng.sarl_select(want_values=False)
eng.sync()
dt = (time.perf_counter() - t0) / iters
out = (C.c_ulonglong * 16)()
assert probe(out, 0) == 0
n = max(int(out[15]), 1)
total = sum(out[k] for k in range(15)) / n
print('cn_sarl_select: %.1f us per call (host clock, %d calls); %d workgroups timed, %.0f ticks each' % (dt * 1e6, iters, n, total))
for k, name in enumerate(NAMES):
    print('  %-22s %8.0f ticks' % (name, out[k] / n))

# the whole sampled step (cn_sarl_sample_step: ORCA, network + decision, transition), streamed without a host check
T = 100
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=eng.device)  # noqa: E731
traj, rew, inf, dmn = z((B, T, 5, 13), torch.float32), z((T, B), torch.float64), z((T, B), torch.uint8), z((T, B), torch.float64)
act, alive, done, action = z((T, B), torch.int32), z((B,), torch.uint8), z((B,), torch.uint8), z((B, 2), torch.float64)
step = eng.sarl_sampler(traj, rew, inf, dmn, act, alive, done, action)
for rep in range(3):
    eng.reset(2000 + np.arange(B))
    alive.fill_(1)
    done.zero_()
    eng.sync()
    t0 = time.perf_counter()
    for t in range(T):
        step(t, 0.1)
    eng.sync()
    print('cn_sarl_sample_step: %.1f us per step (host clock, %d steps streamed)' % ((time.perf_counter() - t0) / T * 1e6, T))
