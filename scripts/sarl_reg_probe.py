"""Per-layer shader-clock breakdown of sarl_reg_kernel (profiling build, -DCN_PHASE_TIMING): ticks per tile as wave 0 of every
workgroup sees them, against the issue time of the layer's MFMAs (32 cycles each, one wave per SIMD).

    CROWDNAV_AMD_LIB=build/exp/lib_timing.so python scripts/sarl_reg_probe.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd  # noqa: E402
from crowdnav_amd import _lib  # noqa: E402
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space  # noqa: E402

# (stage, MFMAs per tile at 5 humans / 13 input features)
STAGES = [('(tile start)', 0), ('mlp1.0 13->150', 10 * 4 * 5), ('mlp1.2 150->100', 7 * 38 * 5), ('mlp2.0 100->100', 7 * 25 * 5),
          ('mlp2.2 100->50 (-> LDS)', 4 * 25 * 5), ('mean + att0 global (1 N tile)', 7 * 25), ('att0 local 100->100', 7 * 25 * 5),
          ('att.2 100->100', 7 * 25 * 5), ('att.4 100->1', 25 * 5), ('softmax + weighted sum', 0),
          ('X prefetch + mlp3.0 56->150 (1 N tile)', 10 * 15), ('mlp3.2 150->100', 7 * 38), ('mlp3.4 100->100', 7 * 25),
          ('mlp3.6 100->1 + store', 25)]


def main():
    om = '--om' in sys.argv
    B = 4096
    lib = _lib.load()
    probe = lib.cn_debug_sarl_cycles
    probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int]
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=0, robot_visible=1)
    eng.reset(2000 + np.arange(B))
    eng.step(np.zeros((B, 2)), update=True)
    torch.manual_seed(0)
    net = ValueNetwork(61 if om else 13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=om)
    eng.sarl_set_weights(net.state_dict())
    for _ in range(2):
        eng.sarl_select(want_values=False)
    eng.sync()
    assert probe(None, 1) == 0
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        eng.sarl_select(want_values=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / iters
    out = (C.c_ulonglong * 16)()
    assert probe(out, 0) == 0
    tiles = out[15]
    per = np.array(out[:14], dtype=np.float64) / tiles
    print('cn_sarl_select %.3f ms (instrumented); %d tiles per launch (wave 0 of each workgroup); %.0f ticks per tile'
          % (dt * 1e3, tiles // iters, per.sum()))
    tot = 0
    for (name, mfma), c in zip(STAGES, per):
        tot += mfma * 32
        print('  %-40s %8.0f  %5.1f %%   MFMA issue %6d  (%4.0f %%)' % (name, c, 100 * c / per.sum(), mfma * 32,
                                                                     100 * mfma * 32 / c if c else 0))
    print('  MFMA issue total %d ticks per tile = %.0f %% of measured' % (tot, 100 * tot / per.sum()))


if __name__ == '__main__':
    main()
