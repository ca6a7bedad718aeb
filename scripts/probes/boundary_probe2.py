"""Why the shard boundary costs 32-37 us in bench.py's driver shape when the same call reads 18.6 us in a loop
(boundary_probe.py): the call behind a 20-step rollout launch, a second call right behind the first, and the first call with the
device kept busy / the host pausing in between."""
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
ev.record()


def timed(fn):
    t0 = time.perf_counter()
    fn()
    ev.record()
    while not ev.query():
        pass
    return (time.perf_counter() - t0) * 1e6


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


rows = {k: [] for k in ('first behind rollout(20)', 'second, at once', 'behind rollout(20), host pause 200 us', 'behind rollout(20) + torch zero_',
                        'behind rollout(20), launched BEFORE the rollout has drained')}
summ = lambda: eng.rollout_summary(out=out)  # noqa: E731
for rep in range(60):
    eng.rollout(20); torch.cuda.synchronize()
    rows['first behind rollout(20)'].append(timed(summ))
    rows['second, at once'].append(timed(summ))
    eng.rollout(20); torch.cuda.synchronize(); time.sleep(0.0002)
    rows['behind rollout(20), host pause 200 us'].append(timed(summ))
    eng.rollout(20); torch.cuda.synchronize(); out.zero_(); torch.cuda.synchronize()
    rows['behind rollout(20) + torch zero_'].append(timed(summ))
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.rollout(20); summ(); ev.record()
    while not ev.query():
        pass
    both = (time.perf_counter() - t0) * 1e6
    t0 = time.perf_counter(); eng.rollout(20); ev.record()
    while not ev.query():
        pass
    alone = (time.perf_counter() - t0) * 1e6
    rows['behind rollout(20), launched BEFORE the rollout has drained'].append(both - alone)
for k, v in rows.items():
    print('%-62s median %6.1f us  min %6.1f' % (k, med(v), min(v)))
