"""Where the N = 1 shard boundary's wall time goes (bench.py: boundary_ms): the summary kernel enqueued on an idle stream until
the host sees its completion, against (a) the same with a trivial torch kernel, (b) the kernel's own device time (HIP events)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import crowdnav_amd  # noqa: E402

B = 4096
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1)
bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=1, per_env_transitions=True)
eng.rollout(200)
out = torch.zeros(8, dtype=torch.float64, device=eng.device)
ev = torch.cuda.Event(enable_timing=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def wall(fn, reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ev.record()
        while not ev.query():
            pass
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 10], ts[len(ts) // 2], ts[-len(ts) // 10]


def device(fn, reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 10], ts[len(ts) // 2], ts[-len(ts) // 10]


for name, fn in (('rollout_summary(out=)', lambda: eng.rollout_summary(out=out)), ('rollout_summary()', eng.rollout_summary),
                 ('torch zero_ [8]', out.zero_), ('nothing (event only)', lambda: None),
                 ('rollout(1)', lambda: eng.rollout(1)), ('rollout(20)', lambda: eng.rollout(20))):
    print('%-24s wall us p10/p50/p90 %6.1f %6.1f %6.1f   device us %6.1f %6.1f %6.1f' % ((name,) + wall(fn) + device(fn)))
