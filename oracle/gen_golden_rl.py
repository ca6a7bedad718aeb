"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/rl_*.npz with the UNMODIFIED reference (/root/reference): the RL-phase sampling of
crowd_nav/train.py:147-157 — `explorer.run_k_episodes(k, 'train', update_memory=True)` with an epsilon-greedy SARL
robot (random-init weights, torch.manual_seed(0)), the reference's own Explorer, ReplayMemory and target model — on
top of oracle/shims + the float32 rvo2 restatement.  Recorded: the replay memory the reference filled (states and TD
targets, in push order) and, from a second identical pass driven step by step, every episode's action indices,
rewards and outcome.  The device pipeline (cn_reset -> cn_sarl_select -> cn_sarl_explore -> cn_sarl_transform ->
cn_step, Explorer._run_batched_rl) has to reproduce all of it.

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_rl.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
INFO = {'Nothing': 0, 'Danger': 1, 'ReachGoal': 2, 'Collision': 3, 'Timeout': 4}


def generate(name, with_om, robot_visible, k, epsilon, first_case=0, policy_name='sarl', extra=None):
    rh.activate()
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    torch.manual_seed(0)
    pcfg = rh.read_config('policy.config', {('sarl', 'with_om'): 'true' if with_om else 'false',
                                            ('lstm_rl', 'with_om'): 'true' if with_om else 'false', **(extra or {})})
    env, robot, policy = rh.make_env(robot_visible=robot_visible, policy_name=policy_name, policy_config=pcfg)
    device = torch.device('cpu')
    policy.set_device(device)
    policy.set_env(env)
    model = policy.get_model()
    memory = ReplayMemory(100000)
    explorer = Explorer(env, robot, device, memory, policy.gamma, target_policy=policy)
    explorer.update_target_model(model)
    robot.policy.set_epsilon(epsilon)

    # pass 1: the reference's own loop fills the memory
    env.case_counter['train'] = first_case
    explorer.run_k_episodes(k, 'train', update_memory=True, episode=0)
    states = np.stack([s.numpy() for s, _ in memory.memory]) if len(memory) else np.zeros((0, 5, policy.input_dim()))
    if states.ndim == 2:  # CADRL.transform: one human, [13]
        states = states[:, None, :]
    values = np.array([float(v.item()) for _, v in memory.memory], dtype=np.float32)

    # pass 2: the same episodes step by step (every reset reseeds numpy, the network is deterministic on the CPU)
    env.case_counter['train'] = first_case
    policy.set_phase('train')
    ep_actions, ep_rewards, ep_outcome, ep_steps, ep_time = [], [], [], [], []
    max_t = int(round(env.time_limit / env.time_step)) + 2
    for _ in range(k):
        ob = env.reset('train')
        done, acts, rews = False, [], []
        while not done:
            action = robot.act(ob)
            acts.append([i for i, a in enumerate(policy.action_space) if a == action][0])
            ob, reward, done, info = env.step(action)
            rews.append(float(reward))
        ep_actions.append(acts + [-1] * (max_t - len(acts)))
        ep_rewards.append(rews + [0.0] * (max_t - len(rews)))
        ep_outcome.append(INFO[type(info).__name__])
        ep_steps.append(len(acts))
        ep_time.append(float(env.global_time))
    kept = sum(n for n, o in zip(ep_steps, ep_outcome) if o in (INFO['ReachGoal'], INFO['Collision']))
    assert kept == len(values), (kept, len(values))

    out = dict(memory_states=states.astype(np.float32), memory_values=values, ep_actions=np.array(ep_actions),
               ep_rewards=np.array(ep_rewards), ep_outcome=np.array(ep_outcome), ep_steps=np.array(ep_steps),
               ep_time=np.array(ep_time), epsilon=np.array(epsilon), k=np.array(k), first_case=np.array(first_case),
               with_om=np.array(int(with_om)), robot_visible=np.array(int(robot_visible)),
               gamma=np.array(policy.gamma), policy=np.array(policy_name),
               pairwise=np.array(int(bool(extra))),
               action_space=np.array([list(a) for a in policy.action_space], dtype=np.float64))
    for key, v in model.state_dict().items():
        out['param_' + key] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, 'episodes', k, 'outcomes', ep_outcome, 'steps', ep_steps, 'memory', len(values),
          'value range', float(values.min()) if len(values) else None, float(values.max()) if len(values) else None)


def generate_il(name, with_om, robot_visible, k, first_case=0, policy_name='sarl'):
    """The imitation-learning collection of train.py:115-129: ORCA demonstrator, the target policy only transforms."""
    rh.activate()
    from crowd_nav.policy.policy_factory import policy_factory
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    torch.manual_seed(0)
    pcfg = rh.read_config('policy.config', {('sarl', 'with_om'): 'true' if with_om else 'false',
                                            ('lstm_rl', 'with_om'): 'true' if with_om else 'false'})
    env, robot, policy = rh.make_env(robot_visible=robot_visible, policy_name=policy_name, policy_config=pcfg)
    device = torch.device('cpu')
    policy.set_device(device)
    memory = ReplayMemory(100000)
    explorer = Explorer(env, robot, device, memory, policy.gamma, target_policy=policy)
    il_policy = policy_factory['orca']()
    il_policy.multiagent_training = policy.multiagent_training
    il_policy.safety_space = 0 if robot_visible else 0.15
    robot.set_policy(il_policy)
    env.case_counter['train'] = first_case
    explorer.run_k_episodes(k, 'train', update_memory=True, imitation_learning=True)
    states = np.stack([s.numpy() for s, _ in memory.memory])
    if states.ndim == 2:
        states = states[:, None, :]
    values = np.array([float(v.item()) for _, v in memory.memory], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, name), memory_states=states.astype(np.float32), memory_values=values,
                        k=np.array(k), first_case=np.array(first_case), with_om=np.array(int(with_om)),
                        robot_visible=np.array(int(robot_visible)), gamma=np.array(policy.gamma),
                        policy=np.array(policy_name))
    print(name, 'episodes', k, 'memory', len(values), 'value range', float(values.min()), float(values.max()))


if __name__ == '__main__':
    assert rh.available()
    which = sys.argv[1:] or ['sarl', 'cadrl', 'lstm_rl', 'lstm_rl2', 'il']
    if 'il' in which:
        generate_il('il_sarl_om.npz', with_om=True, robot_visible=False, k=8, first_case=600)
        generate_il('il_lstm_rl.npz', with_om=False, robot_visible=True, k=6, first_case=700, policy_name='lstm_rl')
    if 'sarl' in which:
        generate('rl_sarl_plain.npz', with_om=False, robot_visible=False, k=12, epsilon=0.5)
        generate('rl_sarl_om.npz', with_om=True, robot_visible=True, k=8, epsilon=0.3, first_case=100)
    if 'cadrl' in which:  # one human (multiagent_training = false), circle crossing
        generate('rl_cadrl.npz', with_om=False, robot_visible=True, k=10, epsilon=0.4, first_case=200, policy_name='cadrl')
    if 'lstm_rl' in which:  # replay states hold the humans by decreasing distance (lstm_rl.py:96-103)
        generate('rl_lstm_rl.npz', with_om=False, robot_visible=True, k=8, epsilon=0.3, first_case=300, policy_name='lstm_rl')
        generate('rl_lstm_rl_om.npz', with_om=True, robot_visible=False, k=6, epsilon=0.5, first_case=400, policy_name='lstm_rl')
    if 'lstm_rl2' in which:  # lstm_rl.ValueNetwork2 (with_interaction_module = true)
        generate('rl_lstm_rl2.npz', with_om=False, robot_visible=True, k=6, epsilon=0.3, first_case=500, policy_name='lstm_rl',
                 extra={('lstm_rl', 'with_interaction_module'): 'true'})
