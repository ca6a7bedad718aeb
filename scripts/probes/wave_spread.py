"""How uneven are the waves of ONE fused-rollout launch?  (profiling build: make -C crowdnav_amd/csrc exp NAME=timing
DEFS=-DCN_PHASE_TIMING; CROWDNAV_AMD_LIB=build/exp/lib_timing.so python scripts/probes/wave_spread.py)
For launches of 5 .. 1000 steps: shader-clock ticks of the step loop of the fastest / average / slowest wave per step, next to
the HIP-event duration of the launch — a launch ends with its slowest wave."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd  # noqa: E402
from crowdnav_amd import _lib  # noqa: E402

lib = _lib.load()
probe = lib.cn_debug_phase_cycles
probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int]
eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=5, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1)
bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=4)
eng.rollout(300)
eng.sync()
for n in (5, 20, 20, 20, 20, 100, 1000):
    assert probe(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.rollout(n)
    e1.record()
    eng.sync()
    out = (C.c_ulonglong * 16)()
    assert probe(out, 0) == 0
    waves = out[15]
    mean = sum(out[k] for k in range(9)) / waves / n
    fb = out[8] / waves / n
    print('%4d steps: launch %8.1f us = %6.2f us per step; ticks per wave-step: mean %6.0f  slowest wave %6.0f (x %.2f); '
          'fallback mean %5.0f; agents in fallback per wave-step %.3f'
          % (n, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / n, mean, out[14] / n, out[14] / n / mean, fb,
             out[9] / waves / n))
    print('           per phase (ticks per wave-step): pairs %5.0f  2-D solve %5.0f  collide %5.0f  bookkeeping %5.0f  fallback %5.0f; '
          'of the bookkeeping at episode ends: io block %5.0f  ring loads %5.0f' % tuple(out[k] / waves / n for k in (2, 3, 5, 7, 8, 0, 1)))
