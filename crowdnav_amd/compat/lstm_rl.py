"""LSTM-RL on the reference's policy protocol (crowd_nav/policy/lstm_rl.py:9-104): an LSTM over the humans followed by
the value head (ValueNetwork1, the shipped with_interaction_module = false), optionally with the pairwise interaction
module in front (ValueNetwork2).
Shares SARL's device pipeline (lookahead, rewards, rotate, occupancy maps) with the LSTM head
(cn_sarl_config.model = CN_MODEL_LSTM_RL)."""
import logging

import numpy as np
import torch
import torch.nn as nn

from .sarl import SARL, mlp


class ValueNetwork1(nn.Module):
    """state_dict keys mlp.{0,2,4,6}.*, lstm.{weight,bias}_{ih,hh}_l0, as crowd_nav.policy.lstm_rl.ValueNetwork1."""

    def __init__(self, input_dim, self_state_dim, mlp_dims, lstm_hidden_dim):
        super().__init__()
        self.self_state_dim = self_state_dim
        self.lstm_hidden_dim = lstm_hidden_dim
        self.mlp = mlp(self_state_dim + lstm_hidden_dim, mlp_dims)
        self.lstm = nn.LSTM(input_dim, lstm_hidden_dim, batch_first=True)

    def forward(self, state):
        n = state.shape[0]
        zeros = torch.zeros(1, n, self.lstm_hidden_dim, device=state.device)
        _, (hn, _) = self.lstm(state, (zeros, zeros.clone()))
        return self.mlp(torch.cat([state[:, 0, :self.self_state_dim], hn.squeeze(0)], dim=1))


class ValueNetwork2(nn.Module):
    """state_dict keys mlp1.{0,2,4,6}.*, mlp.{0,2,4,6}.*, lstm.*, as crowd_nav.policy.lstm_rl.ValueNetwork2 (the
    pairwise interaction module: every human's row passes mlp1 before the LSTM, lstm_rl.py:36-66)."""

    def __init__(self, input_dim, self_state_dim, mlp1_dims, mlp_dims, lstm_hidden_dim):
        super().__init__()
        self.self_state_dim = self_state_dim
        self.lstm_hidden_dim = lstm_hidden_dim
        self.mlp1 = mlp(input_dim, mlp1_dims)
        self.mlp = mlp(self_state_dim + lstm_hidden_dim, mlp_dims)
        self.lstm = nn.LSTM(mlp1_dims[-1], lstm_hidden_dim, batch_first=True)

    def forward(self, state):
        n, h, d = state.shape
        self_state = state[:, 0, :self.self_state_dim]
        pair = self.mlp1(state.reshape(-1, d)).reshape(n, h, -1)
        zeros = torch.zeros(1, n, self.lstm_hidden_dim, device=state.device)
        _, (hn, _) = self.lstm(pair, (zeros, zeros.clone()))
        return self.mlp(torch.cat([self_state, hn.squeeze(0)], dim=1))


class LstmRL(SARL):
    def __init__(self):
        super().__init__()
        self.name = 'LSTM-RL'

    def configure(self, config):
        self.set_common_parameters(config)
        pairwise = config.getboolean('lstm_rl', 'with_interaction_module')
        mlp_dims = [int(x) for x in config.get('lstm_rl', 'mlp2_dims').split(', ')]
        hidden = config.getint('lstm_rl', 'global_state_dim')
        self.with_om = config.getboolean('lstm_rl', 'with_om')
        self.net_cfg = dict(gamma=self.gamma, with_om=self.with_om, cell_num=self.cell_num, cell_size=self.cell_size,
                            om_channel_size=self.om_channel_size, mlp1_dims=(hidden, 1), mlp3_dims=mlp_dims,
                            model='lstm_rl', query_env=self.query_env)
        if pairwise:
            mlp1_dims = [int(x) for x in config.get('lstm_rl', 'mlp1_dims').split(', ')]
            if len(mlp1_dims) != 4:
                raise NotImplementedError('the device interaction module holds 4 layers ([lstm_rl] mlp1_dims)')
            self.model = ValueNetwork2(self.input_dim(), self.self_state_dim, mlp1_dims, mlp_dims, hidden)
            self.net_cfg['interaction_dims'] = tuple(mlp1_dims)
        else:
            self.model = ValueNetwork1(self.input_dim(), self.self_state_dim, mlp_dims, hidden)
        self.multiagent_training = config.getboolean('lstm_rl', 'multiagent_training')
        logging.info('Policy: {}LSTM-RL {} pairwise interaction module'.format('OM-' if self.with_om else '',
                                                                              'w/' if pairwise else 'w/o'))

    def predict(self, state):
        # humans sorted by decreasing distance to the robot (lstm_rl.py:96-103); with query_env the network input comes
        # from the env's lookahead (env order), so the sort only shapes the replay-memory state of the train phase;
        # without it the sorted states themselves are propagated (the device sorts the same way: cn_sarl_config.
        # constant_velocity_model)
        me = np.array(state.self_state.position)
        state.human_states = sorted(state.human_states, key=lambda h: np.linalg.norm(np.array(h.position) - me),
                                    reverse=True)
        return super().predict(state)


LstmRL.get_attention_weights = property(lambda self: (_ for _ in ()).throw(AttributeError('no attention')))
