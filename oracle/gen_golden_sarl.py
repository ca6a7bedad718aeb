"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/sarl_*.npz with the UNMODIFIED reference (/root/reference): crowd_nav.policy.sarl.SARL
(random-init weights, torch.manual_seed(0)) driving crowd_sim CrowdSim on top of oracle/shims + the float32 rvo2
restatement.  For every robot decision of a few greedy ('test' phase) episodes it records the state before the
decision and what MultiHumanRL.predict computed: the 81 action values, the chosen action, the lookahead rewards,
the rotated network inputs (+ occupancy maps) and the raw network outputs.

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_sarl.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def snapshot(env):
    return np.array([[a.px, a.py, a.vx, a.vy, a.gx, a.gy, a.radius, a.v_pref] for a in [env.robot] + env.humans])


def generate(name, with_om, robot_visible, cases, max_steps, policy_name='sarl', kinematics='holonomic', extra=None,
             human_num=5):
    rh.activate()
    torch.manual_seed(0)
    pcfg = rh.read_config('policy.config', {('sarl', 'with_om'): 'true' if with_om else 'false',
                                            ('lstm_rl', 'with_om'): 'true' if with_om else 'false',
                                            ('action_space', 'kinematics'): kinematics, **(extra or {})})
    env, robot, policy = rh.make_env(robot_visible=robot_visible, policy_name=policy_name, policy_config=pcfg,
                                     human_num=human_num)
    policy.set_device(torch.device('cpu'))
    policy.set_phase('test')
    policy.set_env(env)
    model = policy.get_model()
    rec = dict(states=[], gtime=[], values=[], best=[], action=[], rewards=[], inputs=[], net_out=[], next_obs=[],
               theta=[], step_reward=[], step_done=[], step_info=[], next_states=[], next_theta=[])
    info_code = {'Nothing': 0, 'Danger': 1, 'ReachGoal': 2, 'Collision': 3, 'Timeout': 4}
    for case in cases:
        ob = env.reset('test', case)
        done, t = False, 0
        while not done and t < max_steps:
            state8, gt, th = snapshot(env), env.global_time, float(robot.theta)
            action = robot.act(ob)  # MultiHumanRL.predict: fills policy.action_values
            if policy.reach_destination(robot.policy.last_state) if False else False:
                pass
            values = np.array(policy.action_values, dtype=np.float64)
            # re-derive the intermediates with the reference's own functions (nothing is patched)
            from crowd_sim.envs.utils.state import JointState
            js = JointState(robot.get_full_state(), ob)
            if policy_name == 'lstm_rl':  # LstmRL.predict sorts the joint state it hands to MultiHumanRL.predict
                me = np.array(js.self_state.position)
                js.human_states = sorted(js.human_states, key=lambda h: np.linalg.norm(np.array(h.position) - me),
                                         reverse=True)
            rewards, inputs, outs, nobs = [], [], [], None
            om = None
            for a in policy.action_space:
                nself = policy.propagate(js.self_state, a)
                if policy.query_env:
                    nh, reward, _, _ = env.onestep_lookahead(a)
                else:  # multi_human_rl.py:39-42
                    from crowd_sim.envs.utils.action import ActionXY
                    nh = [policy.propagate(h, ActionXY(h.vx, h.vy)) for h in js.human_states]
                    reward = policy.compute_reward(nself, nh)
                batch = torch.cat([torch.Tensor([nself + h]) for h in nh], dim=0)
                x = policy.rotate(batch).unsqueeze(0)
                if with_om:
                    if om is None:
                        om = policy.build_occupancy_maps(nh).unsqueeze(0)
                    x = torch.cat([x, om], dim=2)
                rewards.append(float(reward))
                inputs.append(x[0].numpy().copy())
                outs.append(float(model(x).data.item()) if policy_name != 'cadrl' else
                            float(torch.min(model(x[0]), 0)[0].data.item()))  # cadrl.py:162-163
                if nobs is None:
                    nobs = np.array([[h.px, h.py, h.vx, h.vy, h.radius] for h in nh], dtype=np.float64)
            chosen = [i for i, a in enumerate(policy.action_space) if a == action]
            rec['states'].append(state8)
            rec['gtime'].append(gt)
            rec['values'].append(values)
            rec['best'].append(chosen[0] if len(values) else -1)
            rec['action'].append(list(action))  # (vx, vy) or (v, r)
            rec['theta'].append(th)
            rec['rewards'].append(rewards)
            rec['inputs'].append(np.array(inputs, dtype=np.float32))
            rec['net_out'].append(np.array(outs, dtype=np.float32))
            rec['next_obs'].append(nobs)
            ob, sr, done, sinfo = env.step(action)
            rec['step_reward'].append(float(sr))
            rec['step_done'].append(bool(done))
            rec['step_info'].append(info_code[type(sinfo).__name__])
            rec['next_states'].append(snapshot(env))
            rec['next_theta'].append(float(robot.theta))
            t += 1
    out = {k: np.array(v) for k, v in rec.items()}
    out['action_space'] = np.array([list(a) for a in policy.action_space], dtype=np.float64)
    for k, v in model.state_dict().items():
        out['param_' + k] = v.numpy()
    out['with_om'] = np.array(int(with_om))
    out['robot_visible'] = np.array(int(robot_visible))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, 'decisions', len(rec['best']), 'best', rec['best'][:8], 'value range',
          float(out['values'].min()), float(out['values'].max()))


if __name__ == '__main__':
    assert rh.available()
    generate('sarl_plain.npz', with_om=False, robot_visible=False, cases=[0, 1, 2], max_steps=8)
    generate('sarl_om.npz', with_om=True, robot_visible=True, cases=[3, 4, 5], max_steps=8)
    generate('cadrl_plain.npz', with_om=False, robot_visible=True, cases=[6, 7], max_steps=8, policy_name='cadrl')
    generate('lstm_rl_om.npz', with_om=True, robot_visible=True, cases=[8, 9], max_steps=8, policy_name='lstm_rl')
    generate('sarl_unicycle.npz', with_om=False, robot_visible=True, cases=[10, 11, 12], max_steps=10, kinematics='unicycle')
    generate('lstm_rl2_om.npz', with_om=True, robot_visible=True, cases=[13, 14], max_steps=8, policy_name='lstm_rl',
             extra={('lstm_rl', 'with_interaction_module'): 'true'})  # lstm_rl.ValueNetwork2
    generate('sarl_h12.npz', with_om=False, robot_visible=True, cases=[15, 16], max_steps=6, human_num=12)  # streamed humans
    # crowds beyond the one-tile kernels (more than 8 humans): occupancy maps, LSTM-RL ordering, CADRL minimum
    generate('sarl_om_h12.npz', with_om=True, robot_visible=True, cases=[24], max_steps=6, human_num=12)
    generate('lstm_rl_om_h12.npz', with_om=True, robot_visible=True, cases=[25], max_steps=6, policy_name='lstm_rl', human_num=12)
    generate('cadrl_h12.npz', with_om=False, robot_visible=True, cases=[26], max_steps=6, policy_name='cadrl', human_num=12)
    # [action_space] query_env = false: constant-velocity human model + MultiHumanRL.compute_reward
    noq = {('action_space', 'query_env'): 'false'}
    generate('sarl_noquery_om.npz', with_om=True, robot_visible=True, cases=[17, 18, 19], max_steps=12, extra=noq)
    generate('lstm_rl_noquery_om.npz', with_om=True, robot_visible=True, cases=[20, 21], max_steps=12,
             policy_name='lstm_rl', extra=noq)
    generate('sarl_noquery_unicycle.npz', with_om=False, robot_visible=False, cases=[22, 23], max_steps=10,
             kinematics='unicycle', extra=noq)
