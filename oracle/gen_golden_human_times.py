"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates tests/golden/human_times.npz with the UNMODIFIED reference (/root/reference): a visible ORCA robot runs test
cases to ReachGoal, then CrowdSim.get_human_times() (crowd_sim/envs/crowd_sim.py:209-249) continues ONE centralised rvo2
simulation of all agents until every human has arrived.  Recorded per case: the state when the episode ended, the
human_times the reference returned, every agent's final position and the number of simulated steps.

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_human_times.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def snapshot(env):
    return np.array([[a.px, a.py, a.vx, a.vy, a.gx, a.gy, a.radius, a.v_pref] for a in [env.robot] + env.humans])


def main(cases=(0, 1, 2, 3, 4, 5, 6, 7)):
    assert rh.available()
    env, robot, _ = rh.make_env(robot_visible=True)
    rec = dict(case=[], end_state=[], end_time=[], times_before=[], human_times=[], final_pos=[], extra_steps=[])
    for case in cases:
        ob = env.reset('test', case)
        done = False
        while not done:
            ob, reward, done, info = env.step(robot.act(ob))
        if type(info).__name__ != 'ReachGoal':
            continue
        rec['case'].append(case)
        rec['end_state'].append(snapshot(env))
        rec['end_time'].append(env.global_time)
        rec['times_before'].append(list(env.human_times))
        n0 = len(env.states)
        times = env.get_human_times()
        rec['human_times'].append(list(times))
        rec['final_pos'].append(snapshot(env)[:, :2])
        rec['extra_steps'].append(len(env.states) - n0)
    np.savez_compressed(os.path.join(OUT, 'human_times.npz'), **{k: np.array(v) for k, v in rec.items()})
    print('human_times.npz cases', rec['case'], 'extra steps', rec['extra_steps'], 'times[0]', rec['human_times'][0])


if __name__ == '__main__':
    main()
