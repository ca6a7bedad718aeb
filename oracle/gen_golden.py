"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates the committed fixtures under tests/golden/ by running the UNMODIFIED reference Python
(/root/reference: CrowdSim.reset/step, Human/Robot.act, ORCA.predict, Explorer's return formula) on
top of oracle/shims (gym, git) and the float32 `rvo2` restatement (oracle/_build/rvo2*.so).

    make -C oracle && PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What is genuine and what is not:
  * everything float64 (scenario generation, collision / reward / done / info, kinematics, the numpy
    MT19937 stream) comes from the real reference code + numpy on this image;
  * the float32 ORCA velocities come from our restatement of the un-vendored RVO2 library
    ("parity unpinned" w.r.t. upstream binaries; anchored by the 213/284/3 aggregate).
The fixtures cannot be regenerated on the GPU box (no /root/reference there); they travel in-tree.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
INFO_CODE = {'Nothing': 0, 'Danger': 1, 'ReachGoal': 2, 'Collision': 3, 'Timeout': 4}


def snapshot(env):
    rows = []
    for a in [env.robot] + env.humans:
        rows.append([a.px, a.py, a.vx, a.vy, a.gx, a.gy, a.radius, a.v_pref])
    return np.array(rows, dtype=np.float64)


def run_episode(env, robot, phase, case, max_steps=None):
    ob = env.reset(phase, case)
    states = [snapshot(env)]
    actions, rewards, dones, infos, dmins = [], [], [], [], []
    done = False
    while not done:
        action = robot.act(ob)
        ob, reward, done, info = env.step(action)
        states.append(snapshot(env))
        actions.append([action.vx, action.vy])
        rewards.append(float(reward))
        dones.append(bool(done))
        name = type(info).__name__
        infos.append(INFO_CODE[name])
        dmins.append(float(info.min_dist) if name == 'Danger' else np.nan)
        if max_steps is not None and len(actions) >= max_steps:
            break
    return dict(states=np.array(states), actions=np.array(actions, dtype=np.float64),
                rewards=np.array(rewards, dtype=np.float64), dones=np.array(dones, dtype=np.uint8),
                infos=np.array(infos, dtype=np.uint8), dmins=np.array(dmins, dtype=np.float64),
                global_time=float(env.global_time))


def pack(episodes):
    """Ragged list of episodes -> flat arrays + offsets."""
    T = np.array([len(e['actions']) for e in episodes], dtype=np.int64)
    return dict(
        steps=T,
        states=np.concatenate([e['states'] for e in episodes], axis=0),  # sum(T+1) x A x 8
        actions=np.concatenate([e['actions'] for e in episodes], axis=0),
        rewards=np.concatenate([e['rewards'] for e in episodes]),
        dones=np.concatenate([e['dones'] for e in episodes]),
        infos=np.concatenate([e['infos'] for e in episodes]),
        dmins=np.concatenate([e['dmins'] for e in episodes]),
        global_time=np.array([e['global_time'] for e in episodes]),
    )


def trajectories(name, cases, phase='test', fresh_env_per_case=False, **env_kw):
    env, robot, _ = rh.make_env(**env_kw)
    eps = []
    for c in cases:
        if fresh_env_per_case:
            # a persistent robot ORCA policy keeps the radii of its first episode (orca.py:95-110);
            # with randomised attributes that staleness is avoided by a fresh policy per case
            env, robot, _ = rh.make_env(**env_kw)
        eps.append(run_episode(env, robot, phase, c))
    d = pack(eps)
    d['cases'] = np.array(cases, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, 'episodes', len(eps), 'steps', int(d['steps'].sum()),
          'outcomes', np.bincount(d['infos'][np.cumsum(d['steps']) - 1], minlength=5).tolist())


def outcomes_500():
    """Explorer.run_k_episodes bookkeeping (explorer.py:35-72) over the 500 test cases."""
    res = {}
    for tag, vis in (('invisible', False), ('visible', True)):
        env, robot, _ = rh.make_env(robot_visible=vis)
        info_c, steps, ret, rsum = [], [], [], []
        for case in range(500):
            e = run_episode(env, robot, 'test', case)
            info_c.append(e['infos'][-1])
            steps.append(len(e['actions']))
            # explorer.py:71-72 — pow(gamma, t * time_step * v_pref) * reward, python sum()
            ret.append(sum([pow(0.9, t * robot.time_step * robot.v_pref) * r
                            for t, r in enumerate(e['rewards'].tolist())]))
            rsum.append(sum(e['rewards'].tolist()))
        res[tag + '_info'] = np.array(info_c, dtype=np.uint8)
        res[tag + '_steps'] = np.array(steps, dtype=np.int32)
        res[tag + '_return'] = np.array(ret, dtype=np.float64)
        res[tag + '_reward_sum'] = np.array(rsum, dtype=np.float64)
        print('outcomes', tag, np.bincount(info_c, minlength=5).tolist(), 'steps', sum(steps))
    np.savez_compressed(os.path.join(OUT, 'outcomes_500.npz'), **res)


def resets():
    """Initial states produced by the reference's own generator (crowd_sim.py:251-312)."""
    out = {}
    specs = [
        ('test_h5', dict(human_num=5), 'test', list(range(64))),
        ('train_h5', dict(human_num=5), 'train', list(range(32))),
        ('val_h5', dict(human_num=5), 'val', list(range(16))),
        ('test_h5_random', dict(human_num=5, overrides={('env', 'randomize_attributes'): 'true'}),
         'test', list(range(16))),
        ('test_h5_square', dict(human_num=5, overrides={('sim', 'test_sim'): 'square_crossing'}),
         'test', list(range(16))),
        ('test_h10', dict(human_num=10), 'test', list(range(8))),
        ('test_h20', dict(human_num=20), 'test', list(range(4))),
    ]
    offset = {'train': 2000, 'val': 0, 'test': 1000}
    for name, kw, phase, cases in specs:
        env, robot, policy = rh.make_env(**kw)
        policy.multiagent_training = True  # train/val would otherwise collapse to 1 human (:265-267,278)
        states, probes, seeds = [], [], []
        for c in cases:
            env.reset(phase, c)
            states.append(snapshot(env))
            probes.append(np.random.random())  # next value of the stream = position check
            seeds.append(offset[phase] + c)
        out[name + '_states'] = np.array(states)
        out[name + '_probe'] = np.array(probes)
        out[name + '_seeds'] = np.array(seeds, dtype=np.uint32)
        print('reset', name, out[name + '_states'].shape)
    # raw RNG known-answer vectors
    for s in (0, 1000, 2000, 4294965295):
        np.random.seed(s)
        out['mt_random_%d' % s] = np.array([np.random.random() for _ in range(700)])
    np.savez_compressed(os.path.join(OUT, 'resets.npz'), **out)


def debug_case():
    """case -1: three humans on a symmetric layout (crowd_sim.py:286-292): exact neighbour ties."""
    env, robot, _ = rh.make_env(robot_visible=False)
    e = run_episode(env, robot, 'test', -1, max_steps=12)
    d = pack([e])
    np.savez_compressed(os.path.join(OUT, 'traj_debug_case.npz'), **d)
    print('debug case steps', len(e['actions']))


def main():
    assert rh.available(), 'needs /root/reference and oracle/_build (make -C oracle)'
    os.makedirs(OUT, exist_ok=True)
    trajectories('traj_invisible_h5.npz', list(range(20)), robot_visible=False)
    trajectories('traj_visible_h5.npz', list(range(10)), robot_visible=True)
    trajectories('traj_invisible_h5_random.npz', list(range(6)), robot_visible=False, fresh_env_per_case=True,
                 overrides={('env', 'randomize_attributes'): 'true'})
    trajectories('traj_visible_h5_square.npz', list(range(6)), robot_visible=True,
                 overrides={('sim', 'test_sim'): 'square_crossing'})
    trajectories('traj_visible_h10.npz', list(range(4)), robot_visible=True, human_num=10)
    trajectories('traj_visible_h20.npz', list(range(2)), robot_visible=True, human_num=20)
    debug_case()
    outcomes_500()
    resets()


if __name__ == '__main__':
    main()
