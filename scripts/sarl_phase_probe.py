"""Per-layer shader-clock breakdown of sarl_mlp_kernel (profiling build, -DCN_PHASE_TIMING).

    CROWDNAV_AMD_LIB=build/exp/lib_timing.so python scripts/sarl_phase_probe.py
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

# (name, N, K, row tiles) of what runs between consecutive barriers; MFMA cycles = ctiles_max_per_SIMD * kpad * RT * 32
STAGES = [('stage X', None), ('mlp1.0 13->150', (150, 13, 5)), ('mlp1.2 150->100', (100, 150, 5)),
          ('mean + mlp2.0 100->100', (100, 100, 5)), ('mlp2.2 100->50 + att0 global', (50, 100, 5)),
          ('att0 local 100->100', (100, 100, 5)), ('att.2 100->100', (100, 100, 5)), ('att.4 100->1', (1, 100, 5)),
          ('softmax', None), ('weighted sum', None), ('mlp3.0 56->150', (150, 56, 1)), ('mlp3.2 150->100', (100, 150, 1)),
          ('mlp3.4 100->100', (100, 100, 1)), ('mlp3.6 100->1', (1, 100, 1))]


def ideal(nkr, waves=16):
    if nkr is None:
        return 0
    n, k, rt = nkr
    ctiles = (n + 15) // 16
    kpad = ((k + 3) // 4 + 3) // 4 * 4
    per_simd = max(len([c for c in range(ctiles) if (c % waves) % 4 == s]) for s in range(4))
    return per_simd * kpad * rt * 32


def main():
    B = 4096
    lib = _lib.load()
    probe = lib.cn_debug_sarl_cycles
    probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int]
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=0, robot_visible=1)
    eng.reset(2000 + np.arange(B))
    eng.step(np.zeros((B, 2)), update=True)
    torch.manual_seed(0)
    net = ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]))
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
    print('cn_sarl_select %.3f ms (instrumented); %d tiles per launch; %.0f ticks per tile (wave 0, barrier to barrier)'
          % (dt * 1e3, tiles // iters, per.sum()))
    tot_ideal = 0
    for (name, nkr), c in zip(STAGES, per):
        tot_ideal += ideal(nkr)
        print('  %-32s %8.0f  %5.1f %%   MFMA-bound %6d' % (name, c, 100 * c / per.sum(), ideal(nkr)))
    print('  MFMA-bound total %d ticks per tile = %.0f %% of measured' % (tot_ideal, 100 * tot_ideal / per.sum()))


if __name__ == '__main__':
    main()
