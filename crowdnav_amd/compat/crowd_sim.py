"""CrowdSim — the reference's gym.Env surface (crowd_sim/envs/crowd_sim.py:13-420) over a 1-env engine.

    env = CrowdSim(); env.configure(env_config); env.set_robot(robot)
    ob = env.reset(phase, test_case); ob, reward, done, info = env.step(action); env.onestep_lookahead(action)

Every transition (the humans' ORCA solves, collision / reward / done, integration) and every seeded reset runs
in libcrowdnav_amd on the GPU; this class only moves the results into the objects the reference's callers read.
All three scenario rules are covered (`mixed`: 5 agent slots, absent humans parked; unlike the reference — which sizes
human_times from the previous episode, crowd_sim.py:262-265 — consecutive mixed resets work).  get_human_times runs its centralised
float32 simulation through cn_orca.  Not covered (raise NotImplementedError): render, value-network policies under
`mixed`."""
import configparser
import logging

import numpy as np

from .. import _lib
from ..engine import BatchedCrowdSim
from .agents import Human
from .policy import is_device_orca
from .types import ActionXY, ObservableState, info_from_code

try:  # pragma: no cover
    import gym
    _Base = gym.Env
except Exception:
    _Base = object

_RULES = {'circle_crossing': _lib.CIRCLE_CROSSING, 'square_crossing': _lib.SQUARE_CROSSING, 'mixed': _lib.MIXED}
_MIXED_SLOTS = 5  # `mixed` draws 0..5 humans per episode (crowd_sim.py:105-106); the engine holds 5 slots


def default_env_config(overrides=None):
    """The shipped crowd_nav/configs/env.config as a RawConfigParser; overrides: {(section, key): value}."""
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        'env': dict(time_limit=25, time_step=0.25, val_size=100, test_size=500, randomize_attributes='false'),
        'reward': dict(success_reward=1, collision_penalty=-0.25, discomfort_dist=0.2, discomfort_penalty_factor=0.5),
        'sim': dict(train_val_sim='circle_crossing', test_sim='circle_crossing', square_width=10, circle_radius=4,
                    human_num=5),
        'humans': dict(visible='true', policy='orca', radius=0.3, v_pref=1, sensor='coordinates'),
        'robot': dict(visible='false', policy='none', radius=0.3, v_pref=1, sensor='coordinates'),
    })
    for (sec, key), val in (overrides or {}).items():
        cfg.set(sec, key, str(val))
    return cfg


class CrowdSim(_Base):
    metadata = {'render.modes': ['human']}

    def __init__(self):
        self.time_limit = self.time_step = None
        self.robot = self.humans = None
        self.global_time = None
        self.human_times = None
        self.success_reward = self.collision_penalty = self.discomfort_dist = self.discomfort_penalty_factor = None
        self.config = None
        self.case_capacity = self.case_size = self.case_counter = None
        self.randomize_attributes = None
        self.train_val_sim = self.test_sim = None
        self.square_width = self.circle_radius = self.human_num = None
        self.states = None
        self.action_values = self.attention_weights = None
        self._engines = {}
        self._eng = None
        self.device = 0

    # ------------------------------------------------------------------ crowd_sim.py:51-82
    def configure(self, config):
        self.config = config
        self.time_limit = config.getint('env', 'time_limit')
        self.time_step = config.getfloat('env', 'time_step')
        self.randomize_attributes = config.getboolean('env', 'randomize_attributes')
        self.success_reward = config.getfloat('reward', 'success_reward')
        self.collision_penalty = config.getfloat('reward', 'collision_penalty')
        self.discomfort_dist = config.getfloat('reward', 'discomfort_dist')
        self.discomfort_penalty_factor = config.getfloat('reward', 'discomfort_penalty_factor')
        if config.get('humans', 'policy') != 'orca':
            raise NotImplementedError
        umax = int(np.iinfo(np.uint32).max)
        self.case_capacity = {'train': umax - 2000, 'val': 1000, 'test': 1000}
        self.case_size = {'train': umax - 2000, 'val': config.getint('env', 'val_size'),
                          'test': config.getint('env', 'test_size')}
        self.train_val_sim = config.get('sim', 'train_val_sim')
        self.test_sim = config.get('sim', 'test_sim')
        self.square_width = config.getfloat('sim', 'square_width')
        self.circle_radius = config.getfloat('sim', 'circle_radius')
        self.human_num = config.getint('sim', 'human_num')
        self.case_counter = {'train': 0, 'test': 0, 'val': 0}
        logging.info('human number: {}'.format(self.human_num))
        logging.info('Training simulation: {}, test simulation: {}'.format(self.train_val_sim, self.test_sim))
        logging.info('Square width: {}, circle width: {}'.format(self.square_width, self.circle_radius))

    def set_robot(self, robot):
        self.robot = robot
        if is_device_orca(getattr(robot, 'policy', None)):
            robot.policy._sim_env = self
        for eng in self._engines.values():  # a new robot (policy object): nothing captured yet (orca.py:95-104)
            eng.drop_robot_sim()

    # ------------------------------------------------------------------ engine plumbing
    def engine_config(self, num_envs, human_num, rule, robot_policy):
        """cn_config for this env's settings (used for the 1-env engine and by Explorer's batched rollouts)."""
        if rule not in _RULES:
            raise NotImplementedError("scenario rule %r is outside the accelerated path" % rule)
        kin = getattr(self.robot, 'kinematics', 'holonomic')
        if kin not in ('holonomic', 'unicycle'):
            raise NotImplementedError('robot kinematics %r' % (kin,))
        if kin == 'unicycle' and robot_policy == _lib.ROBOT_ORCA:
            raise NotImplementedError('the device ORCA robot is holonomic')
        pol = self.robot.policy
        if rule == 'mixed':  # the rule ignores the configured number and draws its own (crowd_sim.py:103-115)
            human_num = _MIXED_SLOTS
        return dict(
            num_envs=num_envs, num_humans=human_num, time_step=self.time_step, time_limit=float(self.time_limit),
            success_reward=self.success_reward, collision_penalty=self.collision_penalty,
            discomfort_dist=self.discomfort_dist, discomfort_penalty_factor=self.discomfort_penalty_factor,
            robot_visible=int(bool(self.robot.visible)), robot_policy=robot_policy,
            robot_safety_space=float(getattr(pol, 'safety_space', 0) or 0), human_safety_space=0.0,
            scenario_rule=_RULES[rule], circle_radius=self.circle_radius, square_width=self.square_width,
            human_radius=self.config.getfloat('humans', 'radius'), human_v_pref=self.config.getfloat('humans', 'v_pref'),
            robot_radius=float(self.robot.radius), robot_v_pref=float(self.robot.v_pref),
            randomize_attributes=int(bool(self.randomize_attributes)), device=self.device,
            robot_kinematics=_lib.UNICYCLE if kin == 'unicycle' else _lib.HOLONOMIC)

    def _engine(self, human_num, rule):
        cfg = self.engine_config(1, human_num, rule, _lib.ROBOT_EXTERNAL)
        key = tuple(sorted(cfg.items()))
        if key not in self._engines:
            self._engines[key] = BatchedCrowdSim(**cfg)
        return self._engines[key]

    def _pull(self):
        """device state -> robot / human objects"""
        s, g = self._eng.get_state()
        s = s.cpu().numpy()[0]
        self.global_time = float(g.cpu()[0])
        r = s[0]
        self.robot.px, self.robot.py, self.robot.vx, self.robot.vy = (float(x) for x in r[:4])
        self.robot.theta = float(self._eng.get_theta().cpu()[0])
        for h, row in zip(self.humans, s[1:]):
            h.set(*(float(row[i]) for i in (0, 1, 4, 5, 2, 3)), theta=0.0, radius=float(row[6]), v_pref=float(row[7]))

    # ------------------------------------------------------------------ crowd_sim.py:251-312
    def reset(self, phase='test', test_case=None):
        if self.robot is None:
            raise AttributeError('robot has to be set!')
        assert phase in ['train', 'val', 'test']
        if test_case is not None:
            self.case_counter[phase] = test_case
        multi = getattr(self.robot.policy, 'multiagent_training', None)
        if not multi:
            self.train_val_sim = 'circle_crossing'
        offset = {'train': self.case_capacity['val'] + self.case_capacity['test'], 'val': 0,
                  'test': self.case_capacity['val']}
        self.robot.set(0, -self.circle_radius, 0, self.circle_radius, 0, 0, np.pi / 2)
        case = self.case_counter[phase]
        if case >= 0:
            if phase in ('train', 'val'):
                human_num, rule = (self.human_num if multi else 1), self.train_val_sim
            else:
                human_num, rule = self.human_num, self.test_sim
            self._rule = rule
            self._eng = self._engine(human_num, rule)
            draws = int(self._eng.reset([offset[phase] + case])[0].item())
            if rule == 'mixed':  # len(self.humans) of this episode; self.human_num as the reference leaves it (:115)
                human_num = int(self._eng.human_count().cpu()[0])
                st = self._eng.get_state()[0].cpu().numpy()[0]
                placeholder = human_num == 1 and tuple(st[1, [0, 1, 4, 5]]) == (0.0, -10.0, 0.0, -10.0)
                self.human_num = 0 if placeholder else human_num
            # the reference seeds numpy's GLOBAL generator here (crowd_sim.py:274) and its scenario draws advance it;
            # a train-phase policy then takes its epsilon-greedy draws from the same stream (multi_human_rl.py:28-30):
            # leave the host generator exactly where the reference leaves it
            np.random.seed(offset[phase] + case)
            if draws:
                np.random.random(draws)
            self.case_counter[phase] = (case + 1) % self.case_size[phase]
        else:
            assert phase == 'test'
            if case != -1:
                raise NotImplementedError
            self.human_num = human_num = 3  # the reference's debug layout (crowd_sim.py:286-292)
            self._rule = self.test_sim
            self._eng = self._engine(3, self.test_sim)
            hr, hv = self.config.getfloat('humans', 'radius'), self.config.getfloat('humans', 'v_pref')
            R = self.circle_radius
            st = np.array([[0, -R, 0, 0, 0, R, self.robot.radius, self.robot.v_pref],
                           [0, -6, 0, 0, 0, 5, hr, hv], [-5, -5, 0, 0, -5, 5, hr, hv], [5, -5, 0, 0, 5, 5, hr, hv]],
                          dtype=np.float64)[None]
            self._eng.set_state(st, np.zeros(1))
        self.humans = [Human(self.config, 'humans') for _ in range(human_num)]
        self.human_times = [0] * human_num
        self._pull()
        self.global_time = 0
        for agent in [self.robot] + self.humans:
            agent.time_step = self.time_step
            if agent.policy is not None:
                agent.policy.time_step = self.time_step
        self.states = list()
        if hasattr(self.robot.policy, 'action_values'):
            self.action_values = list()
        if hasattr(self.robot.policy, 'get_attention_weights'):
            self.attention_weights = list()
        if self.robot.sensor != 'coordinates':
            raise NotImplementedError
        return [h.get_observable_state() for h in self.humans]

    # ------------------------------------------------------------------ crowd_sim.py:314-420
    def onestep_lookahead(self, action):
        return self.step(action, update=False)

    def step(self, action, update=True):
        unicycle = getattr(self.robot, 'kinematics', 'holonomic') == 'unicycle'
        if not isinstance(action, tuple) or not hasattr(action, 'v' if unicycle else 'vx'):
            raise TypeError('expected %s for a %s robot' % ('ActionRot' if unicycle else 'ActionXY',
                                                           'unicycle' if unicycle else 'holonomic'))
        if update:
            self.states.append([self.robot.get_full_state(), [h.get_full_state() for h in self.humans]])
            if hasattr(self.robot.policy, 'action_values'):
                self.action_values.append(self.robot.policy.action_values)
            if hasattr(self.robot.policy, 'get_attention_weights'):
                self.attention_weights.append(self.robot.policy.get_attention_weights())
        out = self._eng.step(np.array([list(action)], dtype=np.float64), update=update)  # (vx, vy) or (v, r)
        reward = float(out['reward'].cpu()[0])
        done = bool(out['done'].cpu()[0])
        info = info_from_code(out['info'].cpu()[0], out['dmin'].cpu()[0])
        obs = out['obs'].cpu().numpy()[0][:len(self.humans)]  # absent humans of a `mixed` episode are parked behind
        if update:
            self._pull()
            for i, h in enumerate(self.humans):
                if self.human_times[i] == 0 and h.reached_destination():
                    self.human_times[i] = self.global_time
        ob = [ObservableState(*(float(x) for x in row)) for row in obs]
        return ob, reward, done, info

    def robot_sim_capture(self, radii=None):
        """(radii float32 [A], maxSpeed) of the robot policy's persistent rvo2 simulator (orca.py:95-110): built at the
        policy's first predict — and again whenever the number of agents changes — from the radii it sees THEN
        (+ 0.01 + safety_space), never refreshed afterwards.  Kept on the policy object, like the reference's sim, so
        the gym surface and the batched Explorer agree; only observable with randomize_attributes."""
        pol = self.robot.policy
        if radii is None:
            radii = [a.radius for a in [self.robot] + self.humans]
        cap = getattr(pol, '_rsim', None)
        if cap is None or len(cap[0]) != len(radii):
            safety = float(getattr(pol, 'safety_space', 0) or 0)
            cap = (np.array([np.float32(r + 0.01 + safety) for r in radii], dtype=np.float32),
                   float(np.float32(self.robot.v_pref)))
            pol._rsim = cap
        return cap

    def robot_orca_action(self):
        """Agent 0's ORCA velocity from the current device state (used by the device-backed ORCA policy)."""
        want = float(getattr(self.robot.policy, 'safety_space', 0) or 0)
        if want != self._eng.config['robot_safety_space']:  # test.py / train.py set it after set_robot
            s, g = self._eng.get_state()
            self._eng = self._engine(self._eng.H, self._rule)
            self._eng.set_state(s, g)
        if self.randomize_attributes:
            cap = self.robot_sim_capture()
            if getattr(self._eng, '_rsim_token', None) is not cap:
                self._eng.set_robot_sim(*cap)
                self._eng._rsim_token = cap
        v = self._eng.orca().cpu().numpy()[0, 0]
        return float(v[0]), float(v[1])

    def sarl_action(self, policy):
        """Greedy SARL decision for the current device state: (best index | -1 stop | -2 invalid, 81 values)."""
        eng = self._eng
        if self._rule == 'mixed' and eng.H != 5:
            raise NotImplementedError('value networks under the mixed rule need the 5 human slots the rule can draw')
        if getattr(eng, 'sarl', None) is None:
            eng.sarl_configure(**policy.engine_kwargs())
        # re-upload the parameters only when the Trainer (or a load_state_dict) has changed them: torch bumps a
        # tensor's _version on every eager in-place update; a replayed hipGraph step does not, so the Trainer also counts
        # its steps on the module (_cn_weights_epoch)
        stamp = (id(policy.model), getattr(policy.model, '_cn_weights_epoch', 0)) + tuple(
            p._version for p in policy.model.parameters())
        if getattr(eng, '_sarl_weights_stamp', None) != stamp:
            eng.sarl_set_weights(policy.model.state_dict())
            eng._sarl_weights_stamp = stamp
        out = eng.sarl_select()
        return int(out['best'].cpu()[0]), out['values'].cpu().numpy()[0].tolist()

    def render(self, mode='human', output_file=None):
        raise NotImplementedError('rendering is outside the accelerated path; use the reference CrowdSim')

    def get_human_times(self):
        """crowd_sim.py:209-249: after the robot reached its goal, run ONE centralised rvo2 simulation of all agents
        until every human has reached its goal, and report when each did.  The centralised simulator holds the raw
        radii (no +0.01 padding: safety space -0.01 on this engine) and integrates float32 positions itself
        (position += velocity * timeStep in float32), so that is what happens here: cn_orca for every agent's new
        velocity, float32 integration on the host, positions written back.  (Difference from the per-agent simulators:
        none but the visit order among exact distance ties — the robot comes first in the centralised simulator.)"""
        if not self.robot.reached_destination():
            raise ValueError('Episode is not done yet')
        agents = [self.robot] + self.humans
        cfg = self.engine_config(1, len(self.humans), self._rule if self._rule != 'mixed' else 'circle_crossing',
                                 _lib.ROBOT_ORCA)
        cfg.update(robot_visible=1, robot_safety_space=-0.01, human_safety_space=-0.01)
        sim = BatchedCrowdSim(**cfg)
        f32 = np.float32
        pos = np.array([[a.px, a.py] for a in agents], dtype=f32)           # rvo2 stores float32
        vel = np.array([[a.vx, a.vy] for a in agents], dtype=f32)
        dt32 = f32(self.time_step)
        max_time = 1000
        while not all(self.human_times):
            state = np.array([[[float(pos[i, 0]), float(pos[i, 1]), float(vel[i, 0]), float(vel[i, 1]), a.gx, a.gy,
                                a.radius, a.v_pref] for i, a in enumerate(agents)]], dtype=np.float64)
            # the preferred velocity comes from the agents' positions as Python sees them (goal - position, :229-232)
            for i, a in enumerate(agents):
                state[0, i, 0], state[0, i, 1] = a.px, a.py
            sim.set_state(state, np.zeros(1))
            new_vel = sim.orca().cpu().numpy()[0].astype(f32)                # doStep: new velocities ...
            sim.drop_robot_sim()
            vel = new_vel
            pos = (pos + vel * dt32).astype(f32)                             # ... then float32 integration
            self.global_time += self.time_step
            if self.global_time > max_time:
                logging.warning('Simulation cannot terminate!')
            for i, human in enumerate(self.humans):
                if self.human_times[i] == 0 and human.reached_destination():
                    self.human_times[i] = self.global_time
            for i, a in enumerate(agents):                                   # for visualization (:243-246)
                a.px, a.py = float(pos[i, 0]), float(pos[i, 1])
            self.states.append([self.robot.get_full_state(), [h.get_full_state() for h in self.humans]])
        return self.human_times
