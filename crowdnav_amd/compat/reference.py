"""Zero-line integration seam: run the reference's OWN scripts (crowd_nav/train.py, crowd_nav/test.py) on this package.

    import crowdnav_amd.compat as cn
    cn.install()                                   # before the script's imports run
    runpy.run_path('crowd_nav/test.py', run_name='__main__')

or from a shell, in the reference's crowd_nav/ directory:

    python -m crowdnav_amd.compat.reference test.py --policy orca --phase test
    python -m crowdnav_amd.compat.reference train.py --policy sarl --gpu

install() registers module aliases under the names those scripts import (train.py:8-14, test.py:7-11), so that

    crowd_sim.envs.utils.robot.Robot          -> compat.Robot
    crowd_sim.envs.policy.orca.ORCA           -> compat.ORCA          (test.py:11,78: isinstance check)
    crowd_nav.policy.policy_factory           -> compat.policy_factory (orca, linear, none, cadrl, lstm_rl, sarl)
    crowd_nav.utils.explorer.Explorer         -> compat.Explorer
    crowd_nav.utils.trainer.Trainer           -> compat.trainer.Trainer
    crowd_nav.utils.memory.ReplayMemory       -> compat.trainer.ReplayMemory
    gym.make('CrowdSim-v0')                   -> compat.CrowdSim()

resolve to the device-backed classes without touching a line of the reference.  A real `gym` is used when importable (its
registry gets the id); otherwise a three-name stand-in (Env, make, envs.registration.register) is installed — the reference
uses nothing else of it (crowd_sim/__init__.py:1-6, train.py:76, test.py:64).  `git` (gitpython, train.py:9,57: one log
line with the commit hash) is replaced by a stand-in ONLY if it cannot be imported.
"""
import importlib
import sys
import types

ENV_ID = 'CrowdSim-v0'


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__crowdnav_amd_alias__ = True
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], leaf, m)
    if not hasattr(m, '__path__'):
        m.__path__ = []  # a package: `from a.b.c import d` walks through it
    return m


def _gym():
    from .crowd_sim import CrowdSim
    try:
        gym = importlib.import_module('gym')
        if not getattr(gym, '__crowdnav_amd_alias__', False):
            from gym.envs.registration import register
            try:
                register(id=ENV_ID, entry_point='crowdnav_amd.compat:CrowdSim')
            except Exception:  # already registered by an earlier install()
                pass
            return gym
    except ImportError:
        pass
    registry = {}

    def register(id, entry_point=None, **_):  # noqa: A002 (gym's own keyword)
        registry.setdefault(id, entry_point)

    def make(env_id, **_):
        if env_id != ENV_ID:
            raise KeyError('crowdnav_amd.compat.reference: only %r is registered (asked for %r)' % (ENV_ID, env_id))
        return CrowdSim()

    class Env(object):
        metadata = {}

    gym = _module('gym', Env=Env, make=make)
    _module('gym.envs')
    _module('gym.envs.registration', register=register, registry=registry)
    return gym


def install():
    """Idempotent.  Returns the dict alias name -> object for inspection."""
    from . import (CADRL, ORCA, SARL, CrowdSim, Explorer, Human, LstmRL, Robot, policy_factory, types as _types)
    from .agents import Agent
    from .policy import Linear, Policy
    from .trainer import ReplayMemory, Trainer
    aliases = {
        'crowd_sim.envs.crowd_sim': dict(CrowdSim=CrowdSim),
        'crowd_sim.envs': dict(CrowdSim=CrowdSim),
        'crowd_sim.envs.utils.agent': dict(Agent=Agent),
        'crowd_sim.envs.utils.robot': dict(Robot=Robot),
        'crowd_sim.envs.utils.human': dict(Human=Human),
        'crowd_sim.envs.utils.action': dict(ActionXY=_types.ActionXY, ActionRot=_types.ActionRot),
        'crowd_sim.envs.utils.state': dict(ObservableState=_types.ObservableState, FullState=_types.FullState,
                                           JointState=_types.JointState),
        'crowd_sim.envs.utils.info': dict(Timeout=_types.Timeout, ReachGoal=_types.ReachGoal, Danger=_types.Danger,
                                          Collision=_types.Collision, Nothing=_types.Nothing),
        'crowd_sim.envs.policy.policy': dict(Policy=Policy),
        'crowd_sim.envs.policy.orca': dict(ORCA=ORCA),
        'crowd_sim.envs.policy.linear': dict(Linear=Linear),
        'crowd_sim.envs.policy.policy_factory': dict(policy_factory=policy_factory),
        'crowd_nav.policy.policy_factory': dict(policy_factory=policy_factory),
        'crowd_nav.policy.sarl': dict(SARL=SARL),
        'crowd_nav.policy.cadrl': dict(CADRL=CADRL),
        'crowd_nav.policy.lstm_rl': dict(LstmRL=LstmRL),
        'crowd_nav.policy.multi_human_rl': dict(MultiHumanRL=SARL),
        'crowd_nav.utils.explorer': dict(Explorer=Explorer),
        'crowd_nav.utils.trainer': dict(Trainer=Trainer),
        'crowd_nav.utils.memory': dict(ReplayMemory=ReplayMemory),
    }
    _gym()
    for name, attrs in aliases.items():
        _module(name, **attrs)
    try:
        importlib.import_module('git')
    except ImportError:
        class _Repo(object):
            def __init__(self, *a, **k):
                self.head = types.SimpleNamespace(object=types.SimpleNamespace(hexsha='0' * 40))
        _module('git', Repo=_Repo)
    return aliases


def main(argv=None):
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit('usage: python -m crowdnav_amd.compat.reference <reference script> [its arguments]')
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name='__main__')


if __name__ == '__main__':
    main()
