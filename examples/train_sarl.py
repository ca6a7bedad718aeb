"""BASELINE configs[4]: the reference's `train.py --policy sarl` schedule (crowd_nav/train.py:96-173) on the
MI355X engine: imitation learning from ORCA demonstrations, then epsilon-greedy RL with a target network; rollouts
and the SARL decision run in libcrowdnav_amd, replay memory and the SGD trainer are plain PyTorch-ROCm.

    python examples/train_sarl.py --il-episodes 3000 --train-episodes 10000        # the shipped train.config
    python examples/train_sarl.py --il-episodes 20 --il-epochs 2 --train-episodes 5 --val-size 5 --test-size 5
"""
import argparse
import copy
import json
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crowdnav_amd.compat as cn  # noqa: E402
from crowdnav_amd.compat.sarl import default_policy_config  # noqa: E402
from crowdnav_amd.compat.trainer import DeviceReplayMemory, ReplayMemory, Trainer  # noqa: E402


def read_ini(path):
    import configparser
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise SystemExit('cannot read %s' % path)
    return cfg


def apply_train_config(args, path):
    """crowd_nav/configs/train.config -> the flags of this script (same names, train.py:86-96, 116-121)."""
    cfg = read_ini(path)
    for sec, keys, cast in (('trainer', ('batch_size',), int),
                            ('imitation_learning', ('il_episodes', 'il_epochs'), int),
                            ('imitation_learning', ('il_learning_rate', 'safety_space'), float),
                            ('train', ('train_batches', 'train_episodes', 'sample_episodes', 'target_update_interval',
                                       'evaluation_interval', 'capacity', 'epsilon_decay', 'checkpoint_interval'), int),
                            ('train', ('rl_learning_rate', 'epsilon_start', 'epsilon_end'), float)):
        for key in keys:
            if cfg.has_option(sec, key):
                setattr(args, key, cast(cfg.get(sec, key)))
    return args


def run(args):
    if args.train_config:
        args = apply_train_config(args, args.train_config)
    if args.seed is not None:  # the reference does not seed torch; runs then differ in initial weights and batch order
        torch.manual_seed(args.seed)
    device = torch.device('cuda:0' if args.gpu and torch.cuda.is_available() else 'cpu')
    if args.env_config:  # the reference's own INI files are accepted as they are (train.py:28-30, 60-75)
        env_cfg = read_ini(args.env_config)
    else:
        env_cfg = cn.default_env_config({('env', 'val_size'): args.val_size, ('env', 'test_size'): args.test_size})
    policy = cn.policy_factory[args.policy]()
    if args.policy_config:
        policy.configure(read_ini(args.policy_config))
    else:
        policy.configure(default_policy_config({(args.policy, 'with_om'): 'true' if args.with_om else 'false'}
                                               if args.policy != 'cadrl' else None))
    policy.set_device(device)
    env = cn.CrowdSim()
    env.configure(env_cfg)
    robot = cn.Robot(env_cfg, 'robot')
    env.set_robot(robot)

    # --gpu: the replay ring lives on the device next to the model (one index_copy per batched rollout, one gather per
    # SGD batch); otherwise the reference-style list memory + DataLoader
    memory = DeviceReplayMemory(args.capacity, device) if device.type == 'cuda' else ReplayMemory(args.capacity)
    model = policy.get_model()
    trainer = Trainer(model, memory, device, args.batch_size)
    explorer = cn.Explorer(env, robot, device, memory, policy.gamma, target_policy=policy)

    timing = dict(il_collect_s=0.0, il_sgd_s=0.0, rl_sample_s=0.0, rl_sgd_s=0.0, eval_s=0.0, il_env_steps=0,
                  rl_env_steps=0, eval_episodes=0)

    def timed(key, fn, *a, **kw):
        if device.type == 'cuda':
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        if device.type == 'cuda':
            torch.cuda.synchronize()
        timing[key] += time.perf_counter() - t0
        return out

    if args.output_dir:
        os.makedirs(args.output_dir, exist_ok=True)
    il_weights = os.path.join(args.output_dir, 'il_model.pth') if args.output_dir else None
    rl_weights = os.path.join(args.output_dir, 'rl_model.pth') if args.output_dir else None
    il_loss = None
    if args.resume:  # train.py:106-111: continue from the RL weights, write to resumed_rl_model.pth
        if not (rl_weights and os.path.exists(rl_weights)):
            raise SystemExit('--resume needs --output-dir with rl_model.pth')
        model.load_state_dict(torch.load(rl_weights, map_location=device))
        rl_weights = os.path.join(args.output_dir, 'resumed_rl_model.pth')
        logging.info('Load reinforcement learning trained weights. Resume training')
    elif il_weights and os.path.exists(il_weights):  # :112-114
        model.load_state_dict(torch.load(il_weights, map_location=device))
        logging.info('Load imitation learning trained weights.')
    else:
        # imitation learning from ORCA demonstrations (train.py:115-132)
        trainer.set_learning_rate(args.il_learning_rate)
        il_policy = cn.policy_factory['orca']()
        il_policy.multiagent_training = policy.multiagent_training
        il_policy.safety_space = 0 if robot.visible else args.safety_space
        robot.set_policy(il_policy)
        env.set_robot(robot)
        timed('il_collect_s', explorer.run_k_episodes, args.il_episodes, 'train', update_memory=True,
              imitation_learning=True)
        timing['il_env_steps'] = int((explorer.last_batch or {}).get('env_steps', 0))
        il_loss = timed('il_sgd_s', trainer.optimize_epoch, args.il_epochs)
        if il_weights:
            torch.save(model.state_dict(), il_weights)
        logging.info('Finish imitation learning. Experience set size: %d/%d', len(memory), memory.capacity)
    explorer.update_target_model(model)

    # reinforcement learning (train.py:134-170)
    policy.set_env(env)
    robot.set_policy(policy)
    env.set_robot(robot)
    trainer.set_learning_rate(args.rl_learning_rate)
    if args.resume:  # :141-145: fill the memory pool with some RL experience first
        robot.policy.set_epsilon(args.epsilon_end)
        timed('rl_sample_s', explorer.run_k_episodes, 100, 'train', update_memory=True, episode=0)
        logging.info('Experience set size: %d/%d', len(memory), memory.capacity)
    episode, rl_loss = 0, None
    while episode < args.train_episodes:
        if args.resume:
            epsilon = args.epsilon_end
        elif episode < args.epsilon_decay:
            epsilon = args.epsilon_start + (args.epsilon_end - args.epsilon_start) / args.epsilon_decay * episode
        else:
            epsilon = args.epsilon_end
        robot.policy.set_epsilon(epsilon)
        if episode % args.evaluation_interval == 0:
            timed('eval_s', explorer.run_k_episodes, env.case_size['val'], 'val', episode=episode)
            timing['eval_episodes'] += env.case_size['val']
        timed('rl_sample_s', explorer.run_k_episodes, args.sample_episodes, 'train', update_memory=True, episode=episode)
        timing['rl_env_steps'] += int((explorer.last_batch or {}).get('env_steps', 0))
        if len(memory):  # (the reference's DataLoader raises on an empty memory: nothing reached a goal or collided yet)
            rl_loss = timed('rl_sgd_s', trainer.optimize_batch, args.train_batches)
        episode += 1
        if episode % args.target_update_interval == 0:
            explorer.update_target_model(model)
        if rl_weights and episode % args.checkpoint_interval == 0:
            torch.save(model.state_dict(), rl_weights)
    timed('eval_s', explorer.run_k_episodes, env.case_size['test'], 'test', episode=episode)
    timing['eval_episodes'] += env.case_size['test']
    if timing['rl_sample_s'] > 0:
        timing['rl_sample_env_steps_per_s'] = timing['rl_env_steps'] / timing['rl_sample_s']
    if timing['il_collect_s'] > 0:
        timing['il_collect_env_steps_per_s'] = timing['il_env_steps'] / timing['il_collect_s']
    out = dict(il_loss=il_loss, rl_loss=rl_loss, memory=len(memory), stats=copy.deepcopy(explorer.last_stats),
               timing=timing, schedule={k: v for k, v in vars(args).items()})
    if args.timing_json:
        with open(args.timing_json, 'w') as f:
            json.dump(out, f, indent=1, default=str)
    return out


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpu', action='store_true', help='keep the torch model / trainer on cuda:0 (rollouts always are)')
    ap.add_argument('--policy', choices=['sarl', 'cadrl', 'lstm_rl'], default='sarl')
    ap.add_argument('--with-om', action='store_true')
    for name in ('env-config', 'policy-config', 'train-config'):
        ap.add_argument('--' + name, default=None, help='the reference\'s crowd_nav/configs/%s file' % name.replace('-', '.'))
    ap.add_argument('--output-dir', default=None, help='il_model.pth / rl_model.pth as train.py writes them')
    ap.add_argument('--resume', action='store_true', help='continue from <output-dir>/rl_model.pth (train.py:106-111)')
    ap.add_argument('--seed', type=int, default=None, help='torch.manual_seed (weights, batch order); default: unseeded')
    ap.add_argument('--timing-json', default=None, help='write losses, final stats and per-phase wall-clock here')
    for name, default in (('il-episodes', 3000), ('il-epochs', 50), ('train-episodes', 10000), ('train-batches', 100),
                          ('sample-episodes', 1), ('target-update-interval', 50), ('evaluation-interval', 1000),
                          ('checkpoint-interval', 1000), ('capacity', 100000), ('batch-size', 100),
                          ('epsilon-decay', 4000), ('val-size', 100), ('test-size', 500)):
        ap.add_argument('--' + name, type=int, default=default)
    for name, default in (('il-learning-rate', 0.01), ('rl-learning-rate', 0.001), ('safety-space', 0.15),
                          ('epsilon-start', 0.5), ('epsilon-end', 0.1)):
        ap.add_argument('--' + name, type=float, default=default)
    return ap


if __name__ == '__main__':
    cli = parser().parse_args()
    handlers = [logging.StreamHandler(sys.stdout)]
    if cli.output_dir:  # train.py:51-56: output.log next to the weights — what crowd_nav/utils/plot.py parses
        os.makedirs(cli.output_dir, exist_ok=True)
        handlers.append(logging.FileHandler(os.path.join(cli.output_dir, 'output.log'), mode='a' if cli.resume else 'w'))
    logging.basicConfig(level=logging.INFO, handlers=handlers, format='%(asctime)s, %(levelname)s: %(message)s',
                        datefmt='%Y-%m-%d %H:%M:%S')
    print(run(cli))
