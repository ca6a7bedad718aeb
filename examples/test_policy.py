"""The reference's `test.py` on the MI355X engine (crowd_nav/test.py:14-110): evaluate a policy over a whole phase
(all cases as one device batch) or run ONE case step by step through the gym surface.

    python examples/test_policy.py --policy orca                          # BASELINE configs[0]: 500 test cases
    python examples/test_policy.py --policy sarl --weights rl_model.pth   # a trained value network (state_dict)
    python examples/test_policy.py --policy orca --test-case 3 --visible  # one episode, then get_human_times()

--visualize / --traj (matplotlib rendering) are outside the accelerated path.
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd.compat as cn  # noqa: E402
from crowdnav_amd.compat.policy import ORCA  # noqa: E402
from crowdnav_amd.compat.sarl import default_policy_config  # noqa: E402


def read_ini(path):
    import configparser
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise SystemExit('cannot read %s' % path)
    return cfg


def run(args):
    device = torch.device('cuda:0' if args.gpu and torch.cuda.is_available() else 'cpu')
    if args.env_config:  # the reference's own INI files are accepted as they are (test.py:16-17, 42-55)
        env_cfg = read_ini(args.env_config)
    else:
        env_cfg = cn.default_env_config({('robot', 'visible'): 'true' if args.visible else 'false'})
    policy = cn.policy_factory[args.policy]()
    overrides = {(args.policy, 'with_om'): 'true'} if args.with_om and args.policy in ('sarl', 'lstm_rl') else {}
    policy.configure(read_ini(args.policy_config) if args.policy_config else default_policy_config(overrides))
    if policy.trainable:
        if args.weights:
            policy.get_model().load_state_dict(torch.load(args.weights, map_location='cpu'))
        else:
            logging.warning('no --weights: evaluating a randomly initialised %s value network', args.policy)
    env = cn.CrowdSim()
    env.configure(env_cfg)
    if args.square:
        env.test_sim = 'square_crossing'
    if args.circle:
        env.test_sim = 'circle_crossing'
    robot = cn.Robot(env_cfg, 'robot')
    robot.set_policy(policy)
    env.set_robot(robot)
    explorer = cn.Explorer(env, robot, device, gamma=0.9)
    policy.set_phase(args.phase)
    policy.set_device(device)
    if isinstance(robot.policy, ORCA):  # test.py:78-85: the invisible robot has to keep clear of humans on its own
        robot.policy.safety_space = 0
        logging.info('ORCA agent buffer: %f', robot.policy.safety_space)
    policy.set_env(env)
    robot.print_info()
    if args.test_case is not None:
        ob = env.reset(args.phase, args.test_case)
        done, last = False, np.array(robot.get_position())
        while not done:
            ob, _, done, info = env.step(robot.act(ob))
            now = np.array(robot.get_position())
            logging.debug('Speed: %.2f', np.linalg.norm(now - last) / robot.time_step)
            last = now
        logging.info('It takes %.2f seconds to finish. Final status is %s', env.global_time, info)
        out = dict(time=env.global_time, info=str(info))
        if robot.visible and isinstance(info, cn.ReachGoal):
            out['human_times'] = env.get_human_times()
            logging.info('Average time for humans to reach goal: %.2f', sum(out['human_times']) / len(out['human_times']))
        return out
    explorer.run_k_episodes(env.case_size[args.phase], args.phase, print_failure=True)
    return dict(explorer.last_stats)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--policy', default='orca', choices=['orca', 'linear', 'sarl', 'cadrl', 'lstm_rl'])
    ap.add_argument('--weights', default=None, help='state_dict of the value network (the reference\'s rl_model.pth)')
    ap.add_argument('--with-om', action='store_true')
    ap.add_argument('--env-config', default=None, help="the reference's crowd_nav/configs/env.config")
    ap.add_argument('--policy-config', default=None, help="the reference's crowd_nav/configs/policy.config")
    ap.add_argument('--gpu', action='store_true', help='keep the torch model on cuda:0 (rollouts always are)')
    ap.add_argument('--phase', default='test', choices=['train', 'val', 'test'])
    ap.add_argument('--test-case', type=int, default=None)
    ap.add_argument('--visible', action='store_true', help='[robot] visible = true')
    ap.add_argument('--square', action='store_true')
    ap.add_argument('--circle', action='store_true')
    return ap


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO, format='%(asctime)s, %(levelname)s: %(message)s', datefmt='%Y-%m-%d %H:%M:%S')
    print(run(parser().parse_args()))
