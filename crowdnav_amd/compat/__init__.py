"""Drop-in surface of the reference for the accelerated path: gym-style CrowdSim, Robot/Human agents, the
device-backed ORCA policy, Explorer.  `register()` installs CrowdSim under the gym id the reference uses."""
from .agents import Human, Robot
from .crowd_sim import CrowdSim, default_env_config
from .explorer import Explorer
from .policy import ORCA, Policy, policy_factory, _register_trainable
from .sarl import SARL, ValueNetwork, build_action_space
from .cadrl import CADRL
from .lstm_rl import LstmRL
from .types import (ActionRot, ActionXY, Collision, Danger, FullState, JointState, Nothing, ObservableState,
                    ReachGoal, Timeout)

_register_trainable()

from .reference import install  # noqa: E402  (run the reference's own train.py / test.py on these classes)


def register():
    """gym.make('CrowdSim-v0') -> crowdnav_amd.compat.CrowdSim (reference: crowd_sim/__init__.py:3-6)."""
    from gym.envs.registration import register as gym_register
    gym_register(id='CrowdSim-v0', entry_point='crowdnav_amd.compat:CrowdSim')


__all__ = ['install', 'CrowdSim', 'Explorer', 'Robot', 'Human', 'ORCA', 'SARL', 'CADRL', 'LstmRL', 'ValueNetwork', 'build_action_space', 'Policy', 'policy_factory', 'default_env_config',
           'register', 'ActionXY', 'ActionRot', 'ObservableState', 'FullState', 'JointState', 'Timeout',
           'ReachGoal', 'Danger', 'Collision', 'Nothing']
