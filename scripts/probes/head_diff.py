"""V of the in-tree library against build/exp/lib_ab_headbase.so on the same decision (4096 x 81 x 5): where do they differ?"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch, crowdnav_amd
    from crowdnav_amd.compat.sarl import ValueNetwork, build_action_space
    om = sys.argv[1] == '1'
    torch.manual_seed(7)
    d = 61 if om else 13
    net = ValueNetwork(d, 6, [150, 100], [100, 50], [150, 100, 100, 1], [100, 100, 1], True, 1.0, 4)
    B = 4096
    eng = crowdnav_amd.BatchedCrowdSim(num_envs=B, num_humans=5, robot_policy=crowdnav_amd.ROBOT_EXTERNAL, robot_visible=1)
    eng.reset(2000 + np.arange(B)); eng.step(np.zeros((B, 2)), update=True)
    space, _, _ = build_action_space(1.0)
    eng.sarl_configure(actions=np.array([[a.vx, a.vy] for a in space]), with_om=om)
    eng.sarl_set_weights(net.state_dict())
    eng.sarl_select()
    np.save(sys.argv[2], eng.sarl_export('V').cpu().numpy())
    sys.exit(0)
for om in ('0', '1'):
    out = []
    for lib, f in (('', '/tmp/v_new.npy'), (os.path.join(ROOT, 'build/exp/lib_ab_headbase.so'), '/tmp/v_base.npy')):
        subprocess.run([sys.executable, __file__, om, f], env=dict(os.environ, CROWDNAV_AMD_LIB=lib), check=True)
        out.append(np.load(f).reshape(-1))
    diff = np.abs(out[0] - out[1])
    idx = np.argsort(-diff)[:12]
    print('om', om, 'max diff', diff.max(), 'count > 1e-6:', int((diff > 1e-6).sum()), 'of', diff.size)
    for i in idx:
        print('  group %7d (tile %5d, row %2d)  new %.7f base %.7f diff %.2e' % (i, i // 16, i % 16, out[0][i], out[1][i], diff[i]))
