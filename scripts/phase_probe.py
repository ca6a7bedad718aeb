"""Per-phase shader-clock breakdown of the fused rollout (profiling build only).

    make -C crowdnav_amd/csrc exp NAME=timing DEFS=-DCN_PHASE_TIMING
    CROWDNAV_AMD_LIB=build/exp/lib_timing.so python scripts/phase_probe.py [--humans 5] [--envs 4096]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd  # noqa: E402
from crowdnav_amd import _lib  # noqa: E402

NAMES = ['stage', 'pairs-1 (distances)', 'pairs-2 (rank + half-plane)', 'solve: 2-D program', 'robot action publish',
         'collide (f64 swept distance)', 'reduce + integrate', 'episode bookkeeping / ring',
         'solve: 3-D fallback (+ wait for it)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--humans', type=int, default=5)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--circle-radius', type=float, default=4.0)
    a = ap.parse_args()
    lib = _lib.load()
    probe = lib.cn_debug_phase_cycles
    probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int]
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=a.envs, num_humans=a.humans, robot_policy=crowdnav_amd.ROBOT_ORCA,
                                       robot_visible=1, circle_radius=a.circle_radius)
    bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=4)
    eng.rollout(500)
    eng.sync()
    assert probe(None, 1) == 0
    lazy = getattr(lib, 'cn_debug_lazy_counts', None)
    if lazy is not None:
        lazy.restype, lazy.argtypes = C.c_int, [C.c_void_p, C.c_int]
        lazy(None, 1)
    t0 = time.perf_counter()
    done = 0
    while done < a.steps:
        eng.rollout(500)
        done += 500
    eng.sync()
    dt = time.perf_counter() - t0
    out = (C.c_ulonglong * 16)()
    assert probe(out, 0) == 0
    cyc = np.array(out[:9], dtype=np.float64)
    launches = a.steps // 500
    waves = out[15] / launches
    per = cyc / out[15] / 500  # cycles per wave-step
    print('envs %d humans %d: %.1f M env-steps/s (instrumented), %d waves, %.0f clock ticks per wave-step'
          % (a.envs, a.humans, a.envs * a.steps / dt / 1e6, waves, per.sum()))
    for n, c in zip(NAMES, per):
        print('  %-34s %8.0f  %5.1f %%' % (n, c, 100 * c / per.sum()))
    print('  agents in the 3-D fallback per wave-step: %.3f (of %d agent lanes)' % (out[9] / out[15] / 500, eng.A * max(1, a.envs // int(waves))))
    print('  transitions', int(bufs['transitions'].cpu()[0]))
    if lazy is not None:
        lz = (C.c_ulonglong * 16)()
        lazy(lz, 0)
        if lz[0]:
            print('  lazy fallback (10 half-planes): %d calls, %.2f agents, %.2f solve rounds + %.2f hand-out iterations per call, '
                  '%.0f ticks per call, %.0f per round' % (lz[0], lz[3] / lz[0], lz[1] / lz[0], lz[2] / lz[0], lz[4] / lz[0],
                                                           lz[4] / max(1, lz[1] + lz[2])))
        if lz[5]:
            print('  planar program on three lanes per agent: %d calls, %.2f rounds per call (the last one finds nobody active), '
                  '%.2f agents active per round' % (lz[5], lz[6] / lz[5], lz[7] / max(1, lz[6])))
            work = max(1, lz[8] + lz[9] + lz[10])
            print('    largest violated half-plane index of a working round: <= 3 in %.1f %%, <= 6 in %.1f %%, 7..9 in %.1f %% of them (mean %.2f)'
                  % (100.0 * lz[8] / work, 100.0 * lz[9] / work, 100.0 * lz[10] / work, lz[11] / work))


if __name__ == '__main__':
    main()
