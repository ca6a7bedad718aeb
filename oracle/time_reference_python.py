"""ORACLE — TEST INFRASTRUCTURE ONLY.

Times the UNMODIFIED reference's own CPU path — the loop of crowd_nav/test.py:86-92 / explorer.py:41-48
(`ob = env.reset('test', i); while not done: action = robot.act(ob); ob, _, done, info = env.step(action)`) with the ORCA
robot policy, Python CrowdSim + the `rvo2` module (here: the float32 restatement, oracle/rvo2_pymodule.cpp — upstream
Python-RVO2 is not installable offline) — on ONE core of the machine this runs on.

The reference is found by ref_harness.find_reference(): /root/reference in the build container, the byte-for-byte copy
under the git-ignored oracle/_ref/ on the GPU box (`make -C oracle ref`, run by __graft_entry__.build()).  bench.py runs
this script as a subprocess on the bench host (`--json --cases N`: one JSON line on stdout, nothing written) for
`cpu_baseline.reference_python`; run by hand it also writes profiles/r05_reference_python.json, the labelled fallback
bench.py embeds when no reference copy is present.

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


def run(robot_visible, cases, human_num=5):
    env, robot, _ = ref_harness.make_env(robot_visible=robot_visible, human_num=human_num)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=60, help="test cases per run (two runs: robot invisible / visible)")
    ap.add_argument('--json', action='store_true', help='print one JSON line, write nothing (bench.py)')
    args = ap.parse_args()
    if not ref_harness.available():
        raise SystemExit('needs the reference (/root/reference or oracle/_ref) and `make -C oracle`')
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})  # ONE core, as the line says
    except (AttributeError, OSError):
        pass
    out = measure(args.cases)
    if not args.json:
        path = os.path.join(os.path.dirname(HERE), 'profiles', 'r05_reference_python.json')
        json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
