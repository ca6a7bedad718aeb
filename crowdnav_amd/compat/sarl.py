"""SARL on the reference's policy protocol (crowd_nav/policy/sarl.py:9-86, multi_human_rl.py:11-107,
cadrl.py:11-19, 57-102).  The value network stays a torch module with the reference's parameter names, so
checkpoints (il_model.pth / rl_model.pth) and the PyTorch-ROCm Trainer work unchanged; action selection runs on
the device through cn_sarl_select."""
import itertools
import logging

import numpy as np
import torch
import torch.nn as nn

from .policy import Policy
from .types import ActionRot, ActionXY, FullState, ObservableState


def mlp(input_dim, mlp_dims, last_relu=False):
    """Linear/ReLU stack with the reference's module indexing (Linear at even indices)."""
    dims = [input_dim] + list(mlp_dims)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i != len(dims) - 2 or last_relu:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class ValueNetwork(nn.Module):
    """Same architecture and state_dict keys as crowd_nav.policy.sarl.ValueNetwork (torch reference of the
    device kernel; also what the Trainer optimises)."""

    def __init__(self, input_dim, self_state_dim, mlp1_dims, mlp2_dims, mlp3_dims, attention_dims, with_global_state,
                 cell_size, cell_num):
        super().__init__()
        self.self_state_dim = self_state_dim
        self.global_state_dim = mlp1_dims[-1]
        self.mlp1 = mlp(input_dim, mlp1_dims, last_relu=True)
        self.mlp2 = mlp(mlp1_dims[-1], mlp2_dims)
        self.with_global_state = with_global_state
        self.attention = mlp(mlp1_dims[-1] * (2 if with_global_state else 1), attention_dims)
        self.cell_size = cell_size
        self.cell_num = cell_num
        self.mlp3 = mlp(mlp2_dims[-1] + self_state_dim, mlp3_dims)
        self._attention_row = None

    @property
    def attention_weights(self):
        """Attention weights of batch element 0 of the last forward (sarl.py:54), fetched from the device on demand —
        a host copy inside forward() would cost one device sync per SGD step and forbid graph capture."""
        return None if self._attention_row is None else self._attention_row.cpu().numpy()

    def forward(self, state):
        n, h, d = state.shape
        self_state = state[:, 0, :self.self_state_dim]
        hidden = self.mlp1(state.reshape(-1, d))
        feats = self.mlp2(hidden)
        if self.with_global_state:
            glob = hidden.view(n, h, -1).mean(1, keepdim=True).expand(n, h, self.global_state_dim)
            att_in = torch.cat([hidden, glob.reshape(-1, self.global_state_dim)], dim=1)
        else:
            att_in = hidden
        scores = self.attention(att_in).view(n, h)
        scores_exp = torch.exp(scores) * (scores != 0).float()  # masked, no max-subtraction (sarl.py:52-53)
        weights = (scores_exp / scores_exp.sum(dim=1, keepdim=True)).unsqueeze(2)
        self._attention_row = weights[0, :, 0].detach()
        weighted = (weights * feats.view(n, h, -1)).sum(dim=1)
        return self.mlp3(torch.cat([self_state, weighted], dim=1))


def build_action_space(v_pref, speed_samples=5, rotation_samples=16, kinematics='holonomic'):
    """Action table of CADRL.build_action_space (cadrl.py:82-102): stop + rotations x speeds; ActionXY for a holonomic
    robot, ActionRot(v, r) with r in [-pi/4, pi/4] for a unicycle one."""
    holonomic = kinematics == 'holonomic'
    speeds = [(np.exp((i + 1) / speed_samples) - 1) / (np.e - 1) * v_pref for i in range(speed_samples)]
    if holonomic:
        rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    else:
        rotations = np.linspace(-np.pi / 4, np.pi / 4, rotation_samples)
    space = [ActionXY(0, 0) if holonomic else ActionRot(0, 0)]
    for rotation, speed in itertools.product(rotations, speeds):
        if holonomic:
            space.append(ActionXY(speed * np.cos(rotation), speed * np.sin(rotation)))
        else:
            space.append(ActionRot(speed, rotation))
    return space, speeds, rotations


class SARL(Policy):
    """Drop-in for policy_factory['sarl']: predict() returns the action the reference's greedy loop would pick,
    computed for the env's current device state by the HIP kernels."""

    def __init__(self):
        super().__init__()
        self.name = 'SARL'
        self.trainable = True
        self.multiagent_training = None
        self.kinematics = None
        self.epsilon = None
        self.gamma = None
        self.query_env = None
        self.action_space = None
        self.speeds = self.rotations = None
        self.action_values = None
        self.with_om = None
        self.cell_num = self.cell_size = self.om_channel_size = None
        self.self_state_dim = 6
        self.human_state_dim = 7
        self.joint_state_dim = 13
        self.net_cfg = None

    def set_common_parameters(self, config):
        """[rl], [action_space], [om] of policy.config (cadrl.py:64-73) — shared by the three value-network policies."""
        self.gamma = config.getfloat('rl', 'gamma')
        for key, get in (('kinematics', config.get), ('sampling', config.get), ('speed_samples', config.getint),
                         ('rotation_samples', config.getint), ('query_env', config.getboolean)):
            setattr(self, key, get('action_space', key))
        self.cell_num = config.getint('om', 'cell_num')
        self.cell_size = config.getfloat('om', 'cell_size')
        self.om_channel_size = config.getint('om', 'om_channel_size')
        if self.kinematics not in ('holonomic', 'unicycle'):
            raise NotImplementedError('kinematics %r' % self.kinematics)

    def configure(self, config):
        self.set_common_parameters(config)
        dims = {k: [int(x) for x in config.get('sarl', k).split(', ')]
                for k in ('mlp1_dims', 'mlp2_dims', 'mlp3_dims', 'attention_dims')}
        self.with_om = config.getboolean('sarl', 'with_om')
        with_global_state = config.getboolean('sarl', 'with_global_state')
        self.model = ValueNetwork(self.input_dim(), self.self_state_dim, dims['mlp1_dims'], dims['mlp2_dims'],
                                  dims['mlp3_dims'], dims['attention_dims'], with_global_state, self.cell_size,
                                  self.cell_num)
        self.net_cfg = dict(gamma=self.gamma, with_om=self.with_om, cell_num=self.cell_num, cell_size=self.cell_size,
                            om_channel_size=self.om_channel_size, with_global_state=with_global_state,
                            query_env=self.query_env, **dims)
        self.multiagent_training = config.getboolean('sarl', 'multiagent_training')
        if self.with_om:
            self.name = 'OM-SARL'
        logging.info('Policy: {} {} global state'.format(self.name, 'w/' if with_global_state else 'w/o'))

    def input_dim(self):
        return self.joint_state_dim + (self.cell_num ** 2 * self.om_channel_size if self.with_om else 0)

    def set_device(self, device):
        self.device = device
        self.model.to(device)

    def set_epsilon(self, epsilon):
        self.epsilon = epsilon

    def get_attention_weights(self):
        return self.model.attention_weights

    def build_action_space(self, v_pref):
        self.action_space, self.speeds, self.rotations = build_action_space(v_pref, self.speed_samples,
                                                                          self.rotation_samples, self.kinematics)

    def action_table(self):
        return np.array([list(a) for a in self.action_space], dtype=np.float64)  # (vx, vy) or (v, r)

    def engine_kwargs(self):
        """Arguments of BatchedCrowdSim.sarl_configure for this policy."""
        return dict(actions=self.action_table(), **self.net_cfg)

    def predict(self, state):
        if self.phase is None or self.device is None:
            raise AttributeError('Phase, device attributes have to be set!')
        if self.phase == 'train' and self.epsilon is None:
            raise AttributeError('Epsilon attribute has to be set in training phase')
        if self.action_space is None:
            self.build_action_space(state.self_state.v_pref)
        env = self.env
        if env is None or not hasattr(env, 'sarl_action'):
            raise RuntimeError('crowdnav_amd SARL needs policy.set_env(<crowdnav_amd CrowdSim>)')
        if self.reach_destination(state):
            return ActionXY(0, 0) if self.kinematics == 'holonomic' else ActionRot(0, 0)
        probability = np.random.random()  # drawn in every phase (multi_human_rl.py:28)
        if self.phase == 'train' and probability < self.epsilon:
            action = self.action_space[np.random.choice(len(self.action_space))]
        else:
            best, values = env.sarl_action(self)
            if best < 0:
                raise ValueError('Value network is not well trained. ')
            self.action_values = values
            action = self.action_space[best]
        if self.phase == 'train':
            self.last_state = self.transform(state)
        return action

    # ---- host-side mirrors of the reference's helper methods.  The device path does not call them (cn_sarl_select computes
    # the same quantities for the whole batch); they exist so that code written against the reference's classes finds them.
    def propagate(self, state, action):
        """CADRL.propagate (cadrl.py:104-129): constant-velocity step of an observed human / kinematic step of the robot."""
        dt = self.time_step
        if isinstance(state, ObservableState):
            return ObservableState(state.px + action.vx * dt, state.py + action.vy * dt, action.vx, action.vy, state.radius)
        if not isinstance(state, FullState):
            raise ValueError('Type error')
        if self.kinematics == 'holonomic':
            vx, vy, theta = action.vx, action.vy, state.theta
        else:
            theta = state.theta + action.r
            vx, vy = action.v * np.cos(theta), action.v * np.sin(theta)
        return FullState(state.px + vx * dt, state.py + vy * dt, vx, vy, state.radius, state.gx, state.gy, state.v_pref, theta)

    def rotate(self, state):
        """CADRL.rotate (cadrl.py:187-222) on a batch of 14-float joint rows."""
        return rotate(state, self.kinematics)

    def build_occupancy_maps(self, human_states):
        """MultiHumanRL.build_occupancy_maps (multi_human_rl.py:109-163): tensor [H, cell_num^2 * om_channel_size]."""
        return occupancy_maps(human_states, self.cell_num, self.cell_size, self.om_channel_size)

    def compute_reward(self, nav, humans):
        """MultiHumanRL.compute_reward (multi_human_rl.py:65-88): the reward model of query_env = false, with the reference's
        literal constants (-0.25, 1, 0.2, 0.5) and its distance at the robot's NEXT position against the humans' next ones."""
        dmin = float('inf')
        for human in humans:
            gap = np.linalg.norm((nav.px - human.px, nav.py - human.py)) - nav.radius - human.radius
            if gap < 0:
                return -0.25
            dmin = min(dmin, gap)
        if np.linalg.norm((nav.px - nav.gx, nav.py - nav.gy)) < nav.radius:
            return 1
        return (dmin - 0.2) * 0.5 * self.time_step if dmin < 0.2 else 0


MultiHumanRL = SARL  # the reference's common base of SARL and LSTM-RL (multi_human_rl.py:8); here SARL carries that logic


def default_policy_config(overrides=None):
    """The shipped crowd_nav/configs/policy.config sections SARL reads, as a RawConfigParser."""
    import configparser
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        'rl': dict(gamma=0.9),
        'om': dict(cell_num=4, cell_size=1, om_channel_size=3),
        'action_space': dict(kinematics='holonomic', speed_samples=5, rotation_samples=16, sampling='exponential',
                             query_env='true'),
        'cadrl': dict(mlp_dims='150, 100, 100, 1', multiagent_training='false'),
        'lstm_rl': dict(global_state_dim=50, mlp1_dims='150, 100, 100, 50', mlp2_dims='150, 100, 100, 1',
                        multiagent_training='true', with_om='false', with_interaction_module='false'),
        'sarl': dict(mlp1_dims='150, 100', mlp2_dims='100, 50', attention_dims='100, 100, 1',
                     mlp3_dims='150, 100, 100, 1', multiagent_training='true', with_om='false',
                     with_global_state='true'),
    })
    for (sec, key), val in (overrides or {}).items():
        cfg.set(sec, key, str(val))
    return cfg


def rotate(state, kinematics='holonomic'):
    """Agent-centric 13-vector rows from 14-float joint rows (CADRL.rotate, cadrl.py:187-222)."""
    dx, dy = state[:, 5] - state[:, 0], state[:, 6] - state[:, 1]
    rot = torch.atan2(dy, dx)
    c, s = torch.cos(rot), torch.sin(rot)
    dg = torch.norm(torch.stack([dx, dy], dim=1), 2, dim=1)
    ex, ey = state[:, 9] - state[:, 0], state[:, 10] - state[:, 1]
    theta = state[:, 8] - rot if kinematics == 'unicycle' else torch.zeros_like(dg)
    cols = [dg, state[:, 7], theta, state[:, 4],
            state[:, 2] * c + state[:, 3] * s, state[:, 3] * c - state[:, 2] * s,
            ex * c + ey * s, ey * c - ex * s,
            state[:, 11] * c + state[:, 12] * s, state[:, 12] * c - state[:, 11] * s,
            state[:, 13], torch.norm(torch.stack([-ex, -ey], dim=1), 2, dim=1), state[:, 4] + state[:, 13]]
    return torch.stack(cols, dim=1)


def occupancy_maps(human_states, cell_num, cell_size, channels):
    """(H, cell_num^2 * channels) float32 maps (MultiHumanRL.build_occupancy_maps, multi_human_rl.py:109-163)."""
    hs = np.array([[h.px, h.py, h.vx, h.vy] for h in human_states], dtype=np.float64)
    cells = cell_num ** 2
    out = np.zeros((len(hs), cells * channels), dtype=np.float64)
    for i, me in enumerate(hs):
        others = np.delete(hs, i, axis=0)
        ox, oy = others[:, 0] - me[0], others[:, 1] - me[1]
        mine = np.arctan2(me[3], me[2])
        rot = np.arctan2(oy, ox) - mine
        dist = np.linalg.norm([ox, oy], axis=0)
        xi = np.floor(np.cos(rot) * dist / cell_size + cell_num / 2)
        yi = np.floor(np.sin(rot) * dist / cell_size + cell_num / 2)
        inside = (xi >= 0) & (xi < cell_num) & (yi >= 0) & (yi < cell_num)
        vrot = np.arctan2(others[:, 3], others[:, 2]) - mine
        speed = np.linalg.norm(others[:, 2:4], axis=1)
        ovx, ovy = np.cos(vrot) * speed, np.sin(vrot) * speed
        sums = [[[] for _ in range(3)] for _ in range(cells)]
        for j in np.nonzero(inside)[0]:
            cell = int(cell_num * yi[j] + xi[j])
            sums[cell][0].append(1)
            sums[cell][1].append(ovx[j])
            sums[cell][2].append(ovy[j])
        for cell in range(cells):
            if not sums[cell][0]:
                continue
            mean = [sum(v) / len(v) for v in sums[cell]]
            if channels == 1:
                out[i, cell] = 1
            elif channels == 2:
                out[i, 2 * cell:2 * cell + 2] = mean[1:]
            else:
                out[i, 3 * cell:3 * cell + 3] = mean
    return torch.from_numpy(out).float()


def _transform(self, state):
    """Replay-memory view of a joint state (MultiHumanRL.transform, multi_human_rl.py:90-104)."""
    rows = torch.cat([torch.Tensor([state.self_state + h]).to(self.device) for h in state.human_states], dim=0)
    x = rotate(rows, self.kinematics)
    if self.with_om:
        om = occupancy_maps(state.human_states, self.cell_num, self.cell_size, self.om_channel_size)
        x = torch.cat([x, om.to(self.device)], dim=1)
    return x


SARL.transform = _transform
