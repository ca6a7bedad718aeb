"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/mixed_sarl.npz with the UNMODIFIED reference (/root/reference): value-network policies (SARL,
CADRL, LSTM-RL; random-init weights, torch.manual_seed(0)) acting under test_sim = 'mixed'
(crowd_sim.py:103-151: a different number of humans — 0..5 static obstacles or 1..5 moving humans — per episode), on top of
oracle/shims + the float32 rvo2 restatement.  Per robot decision: the agent states padded to 5 human slots (NaN = absent),
the number of humans, the 81 action values and the chosen action.  A FRESH env per case (the reference only survives one
mixed reset per env, see gen_golden_mixed.py).

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_mixed_sarl.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
SLOTS = 5


def snapshot(env):
    rows = np.full((1 + SLOTS, 8), np.nan)
    for i, a in enumerate([env.robot] + env.humans):
        rows[i] = [a.px, a.py, a.vx, a.vy, a.gx, a.gy, a.radius, a.v_pref]
    return rows


def main():
    assert rh.available()
    out = {}
    for policy_name, with_om in (('sarl', False), ('cadrl', False), ('lstm_rl', False)):  # (occupancy maps need >= 2 humans: multi_human_rl.py:117 raises on a one-human episode)
        rec = dict(states=[], count=[], gtime=[], values=[], best=[], action=[])
        params = None
        for case in range(12):
            rh.activate()
            torch.manual_seed(0)
            pcfg = rh.read_config('policy.config', {('sarl', 'with_om'): 'true' if with_om else 'false'})
            env, robot, policy = rh.make_env(robot_visible=True, policy_name=policy_name, policy_config=pcfg,
                                             overrides={('sim', 'test_sim'): 'mixed'})
            policy.set_device(torch.device('cpu'))
            policy.set_phase('test')
            policy.set_env(env)
            if params is None:
                params = {k: v.numpy().copy() for k, v in policy.get_model().state_dict().items()}
            ob = env.reset('test', case)
            done, t = False, 0
            while not done and t < 4:
                state8, gt = snapshot(env), env.global_time
                action = robot.act(ob)
                values = np.array(policy.action_values, dtype=np.float64)
                chosen = [i for i, a in enumerate(policy.action_space) if a == action]
                rec['states'].append(state8)
                rec['count'].append(len(env.humans))
                rec['gtime'].append(gt)
                rec['values'].append(values)
                rec['best'].append(chosen[0])
                rec['action'].append(list(action))
                ob, _, done, _ = env.step(action)
                t += 1
        for k, v in rec.items():
            out['%s_%s' % (policy_name, k)] = np.array(v)
        for k, v in params.items():
            out['%s_param_%s' % (policy_name, k)] = v
        out['%s_action_space' % policy_name] = np.array([list(a) for a in policy.action_space], dtype=np.float64)
        print(policy_name, 'decisions', len(rec['best']), 'human counts', sorted(set(rec['count'])))
    np.savez_compressed(os.path.join(OUT, 'mixed_sarl.npz'), **out)


if __name__ == '__main__':
    main()
