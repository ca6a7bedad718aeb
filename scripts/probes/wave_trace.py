"""Where does a 20-step launch of the fused rollout spend its time?  (profiling build: make -C crowdnav_amd/csrc exp
NAME=trace DEFS=-DCN_WAVE_TRACE EXP_TU=env; CROWDNAV_AMD_LIB=build/exp/lib_trace.so python scripts/probes/wave_trace.py)
Every wave leaves 100 MHz timestamps of its kernel entry, step-loop entry / exit and kernel exit: dispatch spread, the
distribution of the waves' loop times, which waves finish last and what they did (fallback steps, episode ends)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd  # noqa: E402
from crowdnav_amd import _lib  # noqa: E402

lib = _lib.load()
trace = lib.cn_debug_wave_trace
trace.restype, trace.argtypes = C.c_int, [C.c_void_p, C.c_int]
eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=5, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1)
bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=1)
eng.rollout(200)
eng.sync()
W = 2048
for n in (20, 20, 20, 5, 100, 1000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.rollout(n)
    e1.record()
    eng.sync()
    out = (C.c_ulonglong * (W * 6))()
    assert trace(out, W) == 0
    t = np.array(out[:], dtype=np.uint64).reshape(W, 6)
    t0 = t[:, 0].min()
    entry, loop, exit_, end = [(t[:, k] - t0).astype(np.float64) / 100.0 for k in range(4)]  # us
    fb, ends = (t[:, 4] & np.uint64(0xffffffff)).astype(np.int64), (t[:, 4] >> np.uint64(32)).astype(np.int64)
    dur = exit_ - loop
    last = np.argsort(exit_)[-8:]
    print('%4d steps: events %7.1f us | kernel first entry -> last exit %7.1f us | entries spread over %5.1f us, prologue %4.1f us '
          '(mean), epilogue after the last loop exit %4.1f us' % (n, e0.elapsed_time(e1) * 1e3, end.max(), entry.max(),
                                                                  (loop - entry).mean(), end.max() - exit_.max()))
    print('      step loop per wave: mean %6.1f  p50 %6.1f  p90 %6.1f  p99 %6.1f  max %6.1f us  = %.2f / %.2f us per step (mean / max)'
          % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.mean() / n, dur.max() / n))
    print('      loop time vs fallback steps of the wave: ' + '  '.join(
        '%d-%d: %.1f us (%d waves)' % (a, b, dur[(fb >= a) & (fb <= b)].mean(), ((fb >= a) & (fb <= b)).sum())
        for a, b in ((0, 0), (1, n // 4), (n // 4 + 1, n // 2), (n // 2 + 1, n)) if ((fb >= a) & (fb <= b)).any()))
    print('      loop time vs episode ends of the wave: ' + '  '.join(
        '%d: %.1f us (%d waves)' % (k, dur[ends == k].mean(), (ends == k).sum()) for k in range(0, 4) if (ends == k).any()))
    print('      the 8 waves that left the loop last: ' + '  '.join(
        'entry %.1f loop %.1f us fb %d ends %d' % (entry[w], dur[w], fb[w], ends[w]) for w in last))
    simd = (t[:, 5] >> np.uint64(4)) & np.uint64(3)
    cu = (t[:, 5] >> np.uint64(8)) & np.uint64(15)
    se = (t[:, 5] >> np.uint64(13)) & np.uint64(7)
    print('      HW_ID sample (wave slot, simd, cu, se) of the first 6 workgroups:',
          [(int(t[w, 5] & np.uint64(15)), int(simd[w]), int(cu[w]), int(se[w])) for w in range(6)])
