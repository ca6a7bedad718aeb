"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/persistent_orca.npz with the UNMODIFIED reference (/root/reference): `test.py --policy orca` with
[env] randomize_attributes = true.  ONE ORCA policy object drives the robot through all cases; its rvo2 simulator is
built at the first predict and keeps the radii of the FIRST episode's humans for every later one (orca.py:95-110 only
overwrites positions and velocities) — the behaviour a batched run has to share across its envs (cn_set_robot_sim).
Recorded per case: outcome, steps, discounted return (explorer.py:71-72).

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_persistent.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
INFO = {'Nothing': 0, 'Danger': 1, 'ReachGoal': 2, 'Collision': 3, 'Timeout': 4}


def main(name='persistent_orca.npz', k=40, robot_visible=False):
    assert rh.available()
    env, robot, _ = rh.make_env(robot_visible=robot_visible, overrides={('env', 'randomize_attributes'): 'true'})
    outcome, steps, returns, first_radii = [], [], [], None
    for case in range(k):
        ob = env.reset('test')
        if first_radii is None:
            first_radii = [robot.radius] + [h.radius for h in env.humans]
        done, rewards = False, []
        while not done:
            ob, reward, done, info = env.step(robot.act(ob))
            rewards.append(reward)
        outcome.append(INFO[type(info).__name__])
        steps.append(len(rewards))
        returns.append(sum(pow(0.9, t * robot.time_step * robot.v_pref) * r for t, r in enumerate(rewards)))
    np.savez_compressed(os.path.join(OUT, name), outcome=np.array(outcome), steps=np.array(steps),
                        returns=np.array(returns), first_radii=np.array(first_radii), k=np.array(k),
                        robot_visible=np.array(int(robot_visible)))
    print(name, 'outcomes', np.bincount(outcome, minlength=5).tolist(), 'steps', steps[:10])


if __name__ == '__main__':
    main()
    main('persistent_orca_visible.npz', k=40, robot_visible=True)
