"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/mixed.npz (robot visible) and mixed_invisible.npz with the UNMODIFIED reference (/root/reference): CrowdSim with
test_sim = train_val_sim = 'mixed' (crowd_sim/envs/crowd_sim.py:103-151), ORCA humans and the ORCA robot of
`test.py --policy orca`, on top of oracle/shims + the float32 rvo2 restatement.

  reset_*   the scenario of test cases 0..N-1 straight from the reference's generator: number of humans in env.humans,
            env.human_num as the rule leaves it (0 = the placeholder human at (0, -10)), agent rows padded to 5 humans
  ep_*      full episodes of the first cases: per-step agent states, robot action, reward, done, info

A FRESH env per case: the reference sizes `human_times` with the PREVIOUS episode's human_num before the rule draws the
new one (crowd_sim.py:262-265 vs :115), so a second `mixed` reset that draws more humans than the first raises
IndexError at :406 — the rule only works from a freshly configured env (human_num = 5).

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_mixed.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
INFO_CODE = {'Nothing': 0, 'Danger': 1, 'ReachGoal': 2, 'Collision': 3, 'Timeout': 4}
SLOTS = 5


def snapshot(env):
    rows = np.full((1 + SLOTS, 8), np.nan)
    for i, a in enumerate([env.robot] + env.humans):
        rows[i] = [a.px, a.py, a.vx, a.vy, a.gx, a.gy, a.radius, a.v_pref]
    return rows


def main(name='mixed.npz', n_reset=200, n_episodes=24, robot_visible=True):
    assert rh.available()
    def fresh():
        return rh.make_env(robot_visible=robot_visible,
                           overrides={('sim', 'test_sim'): 'mixed', ('sim', 'train_val_sim'): 'mixed'})[:2]

    reset_states, reset_count, reset_human_num = [], [], []
    for case in range(n_reset):
        env, robot = fresh()
        env.reset('test', case)
        reset_states.append(snapshot(env))
        reset_count.append(len(env.humans))
        reset_human_num.append(env.human_num)
    ep_states, ep_actions, ep_rewards, ep_dones, ep_infos, ep_steps, ep_count = [], [], [], [], [], [], []
    for case in range(n_episodes):
        env, robot = fresh()
        ob = env.reset('test', case)
        states, done = [snapshot(env)], False
        n = 0
        while not done:
            action = robot.act(ob)
            ob, reward, done, info = env.step(action)
            states.append(snapshot(env))
            ep_actions.append([action.vx, action.vy])
            ep_rewards.append(float(reward))
            ep_dones.append(bool(done))
            ep_infos.append(INFO_CODE[type(info).__name__])
            n += 1
        ep_states.append(np.array(states))
        ep_steps.append(n)
        ep_count.append(len(env.humans))
    np.savez_compressed(
        os.path.join(OUT, name), reset_states=np.array(reset_states), reset_count=np.array(reset_count),
        reset_human_num=np.array(reset_human_num), ep_states=np.concatenate(ep_states, axis=0),
        ep_actions=np.array(ep_actions), ep_rewards=np.array(ep_rewards), ep_dones=np.array(ep_dones, dtype=np.uint8),
        ep_infos=np.array(ep_infos, dtype=np.uint8), ep_steps=np.array(ep_steps), ep_count=np.array(ep_count),
        robot_visible=np.array(int(robot_visible)))
    print(name, 'humans per scenario', np.bincount(reset_count, minlength=6).tolist(), 'placeholders',
          int(sum(1 for c, h in zip(reset_count, reset_human_num) if h == 0 and c == 1)),
          'episode steps', ep_steps, 'infos', [ep_infos[sum(ep_steps[:i + 1]) - 1] for i in range(n_episodes)])


if __name__ == '__main__':
    main()
    main('mixed_invisible.npz', n_reset=40, n_episodes=16, robot_visible=False)
