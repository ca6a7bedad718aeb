"""ORACLE — TEST INFRASTRUCTURE ONLY.

Imports the UNMODIFIED reference on top of the shims and the float32 `rvo2` restatement, and offers
small helpers to drive it.  The reference is /root/reference in the build container; on the GPU box —
which has none — it is the byte-for-byte copy `make -C oracle ref` left under oracle/_ref/ (git-ignored,
travels with the snapshot).  Used to pin the C oracle, to generate the committed fixtures under
tests/golden/ (gen_golden.py), to run the reference's own scripts on crowdnav_amd.compat
(tests/test_dropin_surface.py) and to time the reference's Python loop (time_reference_python.py).
"""
import configparser
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def find_reference():
    """$CROWDNAV_REFERENCE, else /root/reference (build container), else oracle/_ref (the copy that travels); None if none"""
    for cand in (os.environ.get('CROWDNAV_REFERENCE'), '/root/reference', os.path.join(HERE, '_ref')):
        if cand and os.path.isfile(os.path.join(cand, 'crowd_nav', 'train.py')):
            return cand
    return None


REFERENCE = find_reference() or '/root/reference'


def available():
    build = os.path.join(HERE, '_build')
    return (os.path.isdir(os.path.join(REFERENCE, 'crowd_sim')) and os.path.isdir(build)
            and any(f.startswith('rvo2.') for f in os.listdir(build)))


def activate():
    """Put shims + oracle rvo2 + reference on sys.path (never writes bytecode into the reference)."""
    sys.dont_write_bytecode = True
    for p in (REFERENCE, os.path.join(HERE, 'shims'), os.path.join(HERE, '_build')):
        if p not in sys.path:
            sys.path.insert(0, p)


def read_config(name, overrides=None):
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(REFERENCE, 'crowd_nav', 'configs', name))
    for (sec, key), val in (overrides or {}).items():
        cfg.set(sec, key, str(val))
    return cfg


def make_env(robot_visible=False, human_num=5, overrides=None, policy_name='orca', policy_config=None):
    """Build (env, robot, policy) exactly as crowd_nav/test.py:51-92 does."""
    activate()
    import gym
    import crowd_sim  # noqa: F401  (registers CrowdSim-v0)
    from crowd_nav.policy.policy_factory import policy_factory
    from crowd_sim.envs.utils.robot import Robot

    ov = {('robot', 'visible'): 'true' if robot_visible else 'false',
          ('sim', 'human_num'): human_num}
    ov.update(overrides or {})
    env_cfg = read_config('env.config', ov)
    policy = policy_factory[policy_name]()
    policy.configure(policy_config if policy_config is not None else env_cfg)
    env = gym.make('CrowdSim-v0')
    logging.disable(logging.INFO)
    env.configure(env_cfg)
    robot = Robot(env_cfg, 'robot')
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_phase('test')
    policy.set_env(env)
    if policy_name == 'orca':
        policy.safety_space = 0
    return env, robot, policy
