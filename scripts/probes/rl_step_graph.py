"""Would a hipGraph of the single-env RL sampling step help?  The ten kernels of one step (select, explore, transform, step,
mask) with fixed output addresses, issued eagerly and replayed from a captured graph: microseconds per step, nothing else."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import crowdnav_amd
from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
B, H, D, N = 1, 5, 13, 400
eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=H, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
eng.reset(2000 + np.arange(B))
space, _, _ = build_action_space(1.0)
eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]))
eng.sarl_set_weights(ValueNetwork(13, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4).state_dict())
dev = eng.device
traj = torch.zeros(B, 1, H, D, dtype=torch.float32, device=dev)
rew = torch.zeros(B, dtype=torch.float64, device=dev); dmn = torch.zeros(B, dtype=torch.float64, device=dev)
inf = torch.zeros(B, dtype=torch.uint8, device=dev); done = torch.zeros(B, dtype=torch.uint8, device=dev)
alive = torch.ones(B, dtype=torch.uint8, device=dev); best = torch.zeros(B, dtype=torch.int32, device=dev)
action = torch.zeros(B, 2, dtype=torch.float64, device=dev)

def step():
    sel = eng.sarl_select(want_values=False, best=best, action=action)
    eng.sarl_explore(sel, 0.1, mask=alive, want_explored=False)
    eng.sarl_transform(out=traj[:, 0], env_stride=H * D)
    eng.step_into(action, rew, done, inf, dmn)
    alive.masked_fill_(done.view(torch.bool), 0)

for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / N
print('eager: %.1f us per step' % (eager * 1e6))
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.use_current_stream()
        step()
    eng.use_current_stream()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / N
    print('graph replay: %.1f us per step' % (gr * 1e6))
except Exception as e:  # noqa: BLE001
    print('capture failed:', repr(e)[:300])
