"""ORACLE — TEST INFRASTRUCTURE ONLY.

Times the UNMODIFIED reference's own CPU paths on ONE core of the machine this runs on — Python CrowdSim + the `rvo2` module
(here: the float32 restatement, oracle/rvo2_pymodule.cpp — upstream Python-RVO2 is not installable offline), torch on the CPU
with one thread.  Three bounded legs, one per number bench.py reports:

  orca      the loop of crowd_nav/test.py:86-92 / explorer.py:41-48 (`ob = env.reset('test', i); while not done: action =
            robot.act(ob); ob, _, done, info = env.step(action)`) with the ORCA robot policy             -> env-steps/s
            (beside bench.py's headline, BASELINE configs[1])
  decision  MultiHumanRL.predict (crowd_nav/policy/multi_human_rl.py:11-63: 81 x onestep_lookahead + 81 batch-1 forwards of
            the value network, random-init weights) behind robot.act, SARL with and without occupancy maps, CADRL (its own
            predict, cadrl.py:130-176) and LSTM-RL (lstm_rl.py:69-104)                                    -> decisions/s
            (beside secondary.sarl / om_sarl / cadrl / lstm_rl, BASELINE configs[2])
  sampling  the train-phase sampling loop of crowd_nav/train.py:156-170: single-episode calls of
            Explorer.run_k_episodes(1, 'train', update_memory=True) with the epsilon-greedy SARL robot    -> env-steps/s
            (beside secondary.sample_step and, scaled by the schedule's env-step counts, the per-phase estimate of
            BASELINE configs[4])

The reference is found by ref_harness.find_reference(): /root/reference in the build container, the byte-for-byte copy
under the git-ignored oracle/_ref/ on the GPU box (`make -C oracle ref`, run by __graft_entry__.build()).  bench.py runs
this script as a subprocess on the bench host (`--json --legs ...`: one JSON line on stdout, nothing written); run by hand it
also writes profiles/r06_reference_python.json, the labelled fallback bench.py embeds when no reference copy is present.

    make -C oracle all ref && PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_python.py
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def run(robot_visible, cases, human_num=5, overrides=None):
    env, robot, _ = ref_harness.make_env(robot_visible=robot_visible, human_num=human_num, overrides=overrides)
    steps = 0
    t0 = time.perf_counter()
    for i in range(cases):
        ob = env.reset('test', i)
        done = False
        while not done:
            action = robot.act(ob)
            ob, _, done, info = env.step(action)
            steps += 1
    return steps, time.perf_counter() - t0


def measure(cases):
    out = {'what': "unmodified reference loop (env.reset('test', i); robot.act; env.step), ORCA robot, 5 humans, "
                   "circle_crossing, one process, one core",
           'host_cpu': cpu_model(), 'cores': 1, 'unit': 'env-steps/s', 'runs': [],
           'reference_from': ref_harness.REFERENCE}
    for visible in (False, True):
        run(visible, 3)  # imports, first-call costs
        steps, dt = run(visible, cases)
        out['runs'].append({'robot_visible': visible, 'test_cases': cases, 'env_steps': steps, 'seconds': dt,
                            'env_steps_per_s': steps / dt})
    out['value'] = min(r['env_steps_per_s'] for r in out['runs'])
    out['value_visible_robot'] = out['runs'][1]['env_steps_per_s']
    return out


def measure_crowd20(cases, radius=12.0):
    """the ORCA loop at BASELINE configs[3]'s crowd: 20 humans, visible robot, circle of `radius` m (the 12 m circle bench.py's
    secondary.h20.r12 runs on: at env.config's 4 m the reference's rejection sampling does not terminate for some seeds)"""
    ov = {('sim', 'circle_radius'): radius}
    run(True, 1, human_num=20, overrides=ov)
    steps, dt = run(True, cases, human_num=20, overrides=ov)
    return {'humans': 20, 'circle_radius': radius, 'robot_visible': True, 'test_cases': cases, 'env_steps': steps, 'seconds': dt,
            'env_steps_per_s': steps / dt, 'unit': 'env-steps/s',
            'what': 'unmodified reference loop, ORCA robot, 20 humans, circle_crossing on a %g m circle, one core' % radius}


def rl_policy(policy_name, with_om, robot_visible=False, human_num=5, interaction=False):
    """(env, robot, policy) with a random-init value network exactly as crowd_nav/train.py:52-80 builds them (CPU device)"""
    import torch
    torch.set_num_threads(1)
    torch.manual_seed(0)
    pcfg = ref_harness.read_config('policy.config', {('sarl', 'with_om'): 'true' if with_om else 'false',
                                                     ('lstm_rl', 'with_om'): 'true' if with_om else 'false',
                                                     ('lstm_rl', 'with_interaction_module'): 'true' if interaction else 'false'})
    env, robot, policy = ref_harness.make_env(robot_visible=robot_visible, policy_name=policy_name, policy_config=pcfg,
                                              human_num=human_num)
    policy.set_device(torch.device('cpu'))
    policy.set_env(env)
    return env, robot, policy


def measure_decisions(policy_name, with_om, n, human_num=5, interaction=False):
    """n calls of the UNMODIFIED robot.act -> <policy>.predict in the 'test' phase (greedy: every call evaluates all 81
    actions), along the episodes the decisions themselves drive from env.reset('test', 0) on.  Only robot.act is timed."""
    env, robot, policy = rl_policy(policy_name, with_om, human_num=human_num, interaction=interaction)
    policy.set_phase('test')
    ob, case, spent, done_n = env.reset('test', 0), 0, 0.0, 0
    action = robot.act(ob)  # first call: action space, lazy imports
    while done_n < n:
        t0 = time.perf_counter()
        action = robot.act(ob)
        spent += time.perf_counter() - t0
        done_n += 1
        ob, _, done, _ = env.step(action)
        if done:
            case += 1
            ob = env.reset('test', case)
    return {'policy': policy_name + ('+om' if with_om else '') + ('+pairwise' if interaction else ''), 'decisions': n, 'seconds': spent, 'decisions_per_s': n / spent,
            'ms_per_decision': spent / n * 1e3, 'humans': human_num, 'actions': len(policy.action_space)}


def measure_sampling(budget_s, max_episodes=50, epsilon=0.5):
    """crowd_nav/train.py:156-170 (sample_episodes = 1): explorer.run_k_episodes(1, 'train', update_memory=True, episode=e)
    with the epsilon-greedy SARL robot, the reference's own Explorer / ReplayMemory / target model — repeated until
    `budget_s` of wall time is spent (at least one episode).  Env-steps of an episode = env.global_time / env.time_step."""
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    env, robot, policy = rl_policy('sarl', False)
    memory = ReplayMemory(100000)
    import torch
    explorer = Explorer(env, robot, torch.device('cpu'), memory, policy.gamma, target_policy=policy)
    explorer.update_target_model(policy.get_model())
    policy.set_phase('train')
    robot.policy.set_epsilon(epsilon)
    steps, episodes = 0, 0
    t0 = time.perf_counter()
    while episodes < max_episodes and (episodes == 0 or time.perf_counter() - t0 < budget_s):
        explorer.run_k_episodes(1, 'train', update_memory=True, episode=episodes)
        steps += int(round(env.global_time / env.time_step))
        episodes += 1
    dt = time.perf_counter() - t0
    return {'policy': 'sarl', 'epsilon': epsilon, 'episodes': episodes, 'env_steps': steps, 'seconds': dt,
            'env_steps_per_s': steps / dt, 'ms_per_env_step': dt / steps * 1e3, 'memory_rows': len(memory)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=60, help="orca leg: test cases per run (two runs: robot invisible / visible)")
    ap.add_argument('--decisions', type=int, default=30, help='decision leg: timed robot.act calls per policy')
    ap.add_argument('--sampling-seconds', type=float, default=8.0, help='sampling leg: wall-time budget')
    ap.add_argument('--legs', default='orca,decision,sampling')
    ap.add_argument('--json', action='store_true', help='print one JSON line, write nothing (bench.py)')
    args = ap.parse_args()
    if not ref_harness.available():
        raise SystemExit('needs the reference (/root/reference or oracle/_ref) and `make -C oracle`')
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})  # ONE core, as the line says
    except (AttributeError, OSError):
        pass
    legs = [s for s in args.legs.split(',') if s]
    out = measure(args.cases) if 'orca' in legs else {'host_cpu': cpu_model(), 'cores': 1, 'reference_from': ref_harness.REFERENCE}
    if 'orca' in legs:
        out['crowd20'] = measure_crowd20(max(2, args.cases // 25))
    if 'decision' in legs:
        out['decision'] = {
            'what': 'unmodified robot.act -> predict (81 onestep_lookahead + 81 batch-1 value-network forwards, random-init '
                    'weights, torch CPU, 1 thread), 5 humans, greedy phase; only robot.act is timed',
            'unit': 'decisions/s',
            'runs': [measure_decisions('sarl', False, args.decisions), measure_decisions('sarl', True, args.decisions),
                     measure_decisions('cadrl', False, args.decisions), measure_decisions('lstm_rl', False, args.decisions),
                     measure_decisions('lstm_rl', False, args.decisions, interaction=True)]}  # lstm_rl.ValueNetwork2
    if 'sampling' in legs:
        out['sampling'] = dict(measure_sampling(args.sampling_seconds), unit='env-steps/s',
                               what="unmodified explorer.run_k_episodes(1, 'train', update_memory=True) calls, epsilon-greedy "
                                    "SARL robot (train.py:156-170), one core")
    if not args.json:
        path = os.path.join(os.path.dirname(HERE), 'profiles', 'r06_reference_python.json')
        json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
