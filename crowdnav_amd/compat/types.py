"""Value types on the gym boundary.  When the reference packages are importable (a user's CrowdNav checkout on
PYTHONPATH) their own classes are re-exported, so isinstance() checks in reference code (Explorer, policies)
keep working; otherwise equivalent stand-ins are defined here.
Reference: crowd_sim/envs/utils/state.py:1-50, action.py:3-4, info.py:1-38."""
from collections import namedtuple

try:  # pragma: no cover - depends on the user's environment
    from crowd_sim.envs.utils.state import ObservableState, FullState, JointState
    from crowd_sim.envs.utils.action import ActionXY, ActionRot
    from crowd_sim.envs.utils.info import Timeout, ReachGoal, Danger, Collision, Nothing
    USING_REFERENCE_TYPES = True
except Exception:  # ImportError, or gym missing inside crowd_sim/__init__
    USING_REFERENCE_TYPES = False

    ActionXY = namedtuple('ActionXY', ['vx', 'vy'])
    ActionRot = namedtuple('ActionRot', ['v', 'r'])

    class _Row(object):
        """A state row: named fields plus tuple concatenation (`row + tuple`, used when flattening joint states)."""
        _fields = ()

        def __init__(self, *values):
            if len(values) != len(self._fields):
                raise TypeError('%s takes %d values' % (type(self).__name__, len(self._fields)))
            for name, value in zip(self._fields, values):
                setattr(self, name, value)
            self.position = (self.px, self.py)
            self.velocity = (self.vx, self.vy)

        def _astuple(self):
            return tuple(getattr(self, f) for f in self._fields)

        def __add__(self, other):
            return other + self._astuple()

        def __str__(self):
            return ' '.join(str(x) for x in self._astuple())

    class ObservableState(_Row):
        _fields = ('px', 'py', 'vx', 'vy', 'radius')

    class FullState(_Row):
        _fields = ('px', 'py', 'vx', 'vy', 'radius', 'gx', 'gy', 'v_pref', 'theta')

        def __init__(self, *values):
            super().__init__(*values)
            self.goal_position = (self.gx, self.gy)

    class JointState(object):
        def __init__(self, self_state, human_states):
            assert isinstance(self_state, FullState)
            assert all(isinstance(h, ObservableState) for h in human_states)
            self.self_state = self_state
            self.human_states = human_states

    def _info(name, text, fields=()):
        def __init__(self, *args):
            for f, a in zip(fields, args):
                setattr(self, f, a)
        return type(name, (object,), {'__init__': __init__, '__str__': lambda self: text})

    Timeout = _info('Timeout', 'Timeout')
    ReachGoal = _info('ReachGoal', 'Reaching goal')
    Danger = _info('Danger', 'Too close', ('min_dist',))
    Collision = _info('Collision', 'Collision')
    Nothing = _info('Nothing', '')

# engine info code (include/crowdnav_amd.h CN_*) -> info object
def info_from_code(code, dmin=None):
    code = int(code)
    if code == 0:
        return Nothing()
    if code == 1:
        return Danger(float(dmin))
    return (ReachGoal, Collision, Timeout)[code - 2]()
