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
from .types import ActionXY


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
        self.attention_weights = None

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
        self.attention_weights = weights[0, :, 0].data.cpu().numpy()
        weighted = (weights * feats.view(n, h, -1)).sum(dim=1)
        return self.mlp3(torch.cat([self_state, weighted], dim=1))


def build_action_space(v_pref, speed_samples=5, rotation_samples=16):
    """Holonomic action table of CADRL.build_action_space (cadrl.py:82-102): stop + rotations x speeds."""
    speeds = [(np.exp((i + 1) / speed_samples) - 1) / (np.e - 1) * v_pref for i in range(speed_samples)]
    rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    space = [ActionXY(0, 0)]
    for rotation, speed in itertools.product(rotations, speeds):
        space.append(ActionXY(speed * np.cos(rotation), speed * np.sin(rotation)))
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

    def configure(self, config):
        self.gamma = config.getfloat('rl', 'gamma')
        self.kinematics = config.get('action_space', 'kinematics')
        self.sampling = config.get('action_space', 'sampling')
        self.speed_samples = config.getint('action_space', 'speed_samples')
        self.rotation_samples = config.getint('action_space', 'rotation_samples')
        self.query_env = config.getboolean('action_space', 'query_env')
        self.cell_num = config.getint('om', 'cell_num')
        self.cell_size = config.getfloat('om', 'cell_size')
        self.om_channel_size = config.getint('om', 'om_channel_size')
        dims = {k: [int(x) for x in config.get('sarl', k).split(', ')]
                for k in ('mlp1_dims', 'mlp2_dims', 'mlp3_dims', 'attention_dims')}
        self.with_om = config.getboolean('sarl', 'with_om')
        with_global_state = config.getboolean('sarl', 'with_global_state')
        if self.kinematics != 'holonomic' or not self.query_env:
            raise NotImplementedError('only holonomic, query_env=true SARL is on the accelerated path')
        self.model = ValueNetwork(self.input_dim(), self.self_state_dim, dims['mlp1_dims'], dims['mlp2_dims'],
                                  dims['mlp3_dims'], dims['attention_dims'], with_global_state, self.cell_size,
                                  self.cell_num)
        self.net_cfg = dict(gamma=self.gamma, with_om=self.with_om, cell_num=self.cell_num, cell_size=self.cell_size,
                            om_channel_size=self.om_channel_size, with_global_state=with_global_state, **dims)
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
                                                                          self.rotation_samples)

    def action_table(self):
        return np.array([[a.vx, a.vy] for a in self.action_space], dtype=np.float64)

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
        probability = np.random.random()  # drawn in every phase (multi_human_rl.py:28)
        if self.phase == 'train' and probability < self.epsilon:
            if self.reach_destination(state):
                return ActionXY(0, 0)
            return self.action_space[np.random.choice(len(self.action_space))]
        best, values = env.sarl_action(self)
        if best == -1:
            return ActionXY(0, 0)
        if best < 0:
            raise ValueError('Value network is not well trained. ')
        self.action_values = values
        return self.action_space[best]

    @staticmethod
    def reach_destination(state):
        s = state.self_state
        return np.linalg.norm((s.py - s.gy, s.px - s.gx)) < s.radius
