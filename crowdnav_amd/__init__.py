"""crowdnav_amd — MI355X (gfx950) batched crowd-navigation rollout engine.

Drop-in for the hot path of vita-epfl/CrowdNav (CrowdSim.step + ORCA humans + Explorer rollouts): Python
host classes mirroring the reference's interface on top of the C-ABI HIP library libcrowdnav_amd.so.
"""
from ._lib import (CrowdNavAmdError, CnConfig, INFO_NAMES, NOTHING, DANGER, REACH_GOAL, COLLISION, TIMEOUT,
                   ROBOT_EXTERNAL, ROBOT_ORCA, CIRCLE_CROSSING, SQUARE_CROSSING, MIXED, HOLONOMIC, UNICYCLE,
                   FLAG_ASYNC_SCENARIO_FILL)
from .engine import BatchedCrowdSim, default_config

__all__ = ['BatchedCrowdSim', 'default_config', 'CrowdNavAmdError', 'CnConfig', 'INFO_NAMES', 'NOTHING',
           'DANGER', 'REACH_GOAL', 'COLLISION', 'TIMEOUT', 'ROBOT_EXTERNAL', 'ROBOT_ORCA', 'CIRCLE_CROSSING',
           'SQUARE_CROSSING', 'MIXED', 'HOLONOMIC', 'UNICYCLE', 'FLAG_ASYNC_SCENARIO_FILL']
