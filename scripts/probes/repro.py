"""Two runs of the same fused rollout (fresh engines): end state and episode records bitwise identical?
usage: repro.py H steps chunk [radius [plain]]   — `plain`: the second run has the 20-human shard's 3-of-4 env schedule switched off
(CROWDNAV_AMD_SCHED_MIN_STEPS, read by cn_create): scheduled and plain launches must leave the same bits"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, crowdnav_amd
H, steps, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
radius = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
def run():
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=H, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1, circle_radius=radius)
    bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=4)
    done = 0
    while done < steps:
        n = min(chunk, steps - done); eng.rollout(n); done += n
    eng.sync(); s, g = eng.get_state()
    out = [s.clone(), g.clone()] + [bufs[k].clone() for k in ('ep_outcome', 'ep_steps', 'ep_return', 'ep_time', 'ep_count', 'cur_return')]
    tr, ep = int(bufs['transitions'].item()), int(bufs['ep_count'].sum().item())
    eng.close()
    return out, tr, ep
plain = len(sys.argv) > 5 and sys.argv[5] == 'plain'
a, tr, ep = run()
if plain:
    os.environ['CROWDNAV_AMD_SCHED_MIN_STEPS'] = '1000000000'
b, _, _ = run()
same = all(torch.equal(x.view(torch.uint8) if x.dtype != torch.uint8 else x, y.view(torch.uint8) if y.dtype != torch.uint8 else y) for x, y in zip(a, b))
print('humans', H, 'steps', steps, 'chunk', chunk, 'radius', radius, 'scheduled vs plain launches' if plain else 'two runs', 'bitwise identical', same, 'transitions', tr, 'paused', 4096 * steps - tr, 'episodes', ep,
      'finite', bool(torch.isfinite(a[0]).all().item()))
