"""Robot / human agent objects on the gym boundary (reference: crowd_sim/envs/utils/agent.py:10-138,
robot.py:5-14, human.py:5-17).  The physics lives on the device; these objects carry the attributes the
reference's callers read (px, py, vx, vy, gx, gy, radius, v_pref, theta, visible, policy, kinematics)."""
import logging

import numpy as np

from .policy import policy_factory
from .types import ActionRot, ActionXY, FullState, JointState, ObservableState


class Agent(object):
    def __init__(self, config, section):
        self.visible = config.getboolean(section, 'visible')
        self.v_pref = config.getfloat(section, 'v_pref')
        self.radius = config.getfloat(section, 'radius')
        name = config.get(section, 'policy')
        if name not in policy_factory:
            raise NotImplementedError('policy %r is outside the accelerated path (orca | none)' % name)
        self.policy = policy_factory[name]()
        self.sensor = config.get(section, 'sensor')
        self.kinematics = self.policy.kinematics if self.policy is not None else None
        self.px = self.py = self.gx = self.gy = self.vx = self.vy = self.theta = None
        self.time_step = None

    def print_info(self):
        logging.info('Agent is {} and has {} kinematic constraint'.format(
            'visible' if self.visible else 'invisible', self.kinematics))

    def set_policy(self, policy):
        self.policy = policy
        self.kinematics = policy.kinematics

    def set(self, px, py, gx, gy, vx, vy, theta, radius=None, v_pref=None):
        self.px, self.py, self.gx, self.gy, self.vx, self.vy, self.theta = px, py, gx, gy, vx, vy, theta
        if radius is not None:
            self.radius = radius
        if v_pref is not None:
            self.v_pref = v_pref

    def sample_random_attributes(self):
        """agent.py:39-45 on the HOST numpy stream (a device reset draws them on the env's own seeded stream instead,
        scenario_device.h, in the same order)."""
        self.v_pref = np.random.uniform(0.5, 1.5)
        self.radius = np.random.uniform(0.3, 0.5)

    def get_observable_state(self):
        return ObservableState(self.px, self.py, self.vx, self.vy, self.radius)

    def get_next_observable_state(self, action):
        """agent.py:63-74: where this agent would be after `action` (what onestep_lookahead reports for the humans)."""
        self.check_validity(action)
        nx, ny = self.compute_position(action, self.time_step)
        if self.kinematics == 'holonomic':
            return ObservableState(nx, ny, action.vx, action.vy, self.radius)
        heading = self.theta + action.r
        return ObservableState(nx, ny, action.v * np.cos(heading), action.v * np.sin(heading), self.radius)

    def get_full_state(self):
        return FullState(self.px, self.py, self.vx, self.vy, self.radius, self.gx, self.gy, self.v_pref, self.theta)

    def get_position(self):
        return self.px, self.py

    def set_position(self, position):
        self.px, self.py = position[0], position[1]

    def get_goal_position(self):
        return self.gx, self.gy

    def get_velocity(self):
        return self.vx, self.vy

    def set_velocity(self, velocity):
        self.vx, self.vy = velocity[0], velocity[1]

    def act(self, ob):
        raise NotImplementedError  # abstract in the reference too (agent.py:95-101)

    def check_validity(self, action):
        assert isinstance(action, ActionXY if self.kinematics == 'holonomic' else ActionRot)

    def compute_position(self, action, delta_t):
        if self.kinematics == 'holonomic':
            return self.px + action.vx * delta_t, self.py + action.vy * delta_t
        theta = self.theta + action.r
        return self.px + np.cos(theta) * action.v * delta_t, self.py + np.sin(theta) * action.v * delta_t

    def step(self, action):
        self.px, self.py = self.compute_position(action, self.time_step)
        if self.kinematics == 'holonomic':
            self.vx, self.vy = action.vx, action.vy
        else:
            self.theta = (self.theta + action.r) % (2 * np.pi)
            self.vx, self.vy = action.v * np.cos(self.theta), action.v * np.sin(self.theta)

    def reached_destination(self):
        d = np.array(self.get_position()) - np.array(self.get_goal_position())
        return np.linalg.norm(d) < self.radius


class Robot(Agent):
    def __init__(self, config, section):
        super().__init__(config, section)

    def act(self, ob):
        if self.policy is None:
            raise AttributeError('Policy attribute has to be set!')
        return self.policy.predict(JointState(self.get_full_state(), ob))


class Human(Agent):
    """Humans are ORCA agents simulated on the device; this object is a read-only view of one of them."""

    def __init__(self, config, section):
        super().__init__(config, section)

    def act(self, ob):
        raise RuntimeError('human actions are computed on the device inside CrowdSim.step')
