# soak: 3 M batched steps of the fused 5-human rollout (12.3 G transitions) run twice (bitwise identical end states?), and
# 100 k steps of the 20-human kernel; bookkeeping must add up (transitions == envs x steps - paused, finite state)
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02w; mkdir -p $OUT; cd $REPO
timeout 120 python scripts/soak.py 3000000 2>&1 | grep -v amdgpu | tee $OUT/soak_h5.txt
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/soak_determinism.txt
import numpy as np, torch, crowdnav_amd
def run(humans, radius, steps, chunk):
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=4096, num_humans=humans, circle_radius=radius, robot_policy=crowdnav_amd.ROBOT_ORCA, robot_visible=1)
    bufs = eng.rollout_begin(seed_base=2000, seed_mod=2 ** 32 - 2000, record_capacity=4)
    for _ in range(steps // chunk): eng.rollout(chunk)
    eng.sync(); s, g = eng.get_state()
    return s.clone(), g.clone(), {k: v.clone() for k, v in bufs.items()}
for humans, radius, steps, chunk in ((5, 4.0, 200000, 1000), (5, 4.0, 200000, 37), (20, 12.0, 40000, 500)):
    a, b = run(humans, radius, steps, chunk), run(humans, radius, steps, chunk)
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    tr = int(a[2]['transitions'].item())
    print('humans', humans, 'steps', steps, 'chunk', chunk, 'two runs bitwise identical', same, 'transitions', tr, 'paused', 4096 * (steps // chunk) * chunk - tr,
          'episodes', int(a[2]['ep_count'].sum().item()), 'finite', bool(torch.isfinite(a[0]).all().item()))
PY
