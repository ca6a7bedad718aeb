"""CADRL on the reference's policy protocol (crowd_nav/policy/cadrl.py:22-185): one MLP value network applied to
every (robot, human) pair, action value = reward + gamma^(dt v_pref) * min over humans.  Shares the device pipeline
of SARL (lookahead, rewards, rotate) with the CADRL head (cn_sarl_config.model = CN_MODEL_CADRL)."""
import logging

import torch
import torch.nn as nn

from .sarl import SARL, mlp, rotate


class ValueNetwork(nn.Module):
    """state_dict keys value_network.{0,2,4,6}.{weight,bias}, as crowd_nav.policy.cadrl.ValueNetwork."""

    def __init__(self, input_dim, mlp_dims):
        super().__init__()
        self.value_network = mlp(input_dim, mlp_dims)

    def forward(self, state):
        return self.value_network(state)


class CADRL(SARL):
    def __init__(self):
        super().__init__()
        self.name = 'CADRL'
        self.with_om = False

    def configure(self, config):
        self.set_common_parameters(config)
        mlp_dims = [int(x) for x in config.get('cadrl', 'mlp_dims').split(', ')]
        self.model = ValueNetwork(self.joint_state_dim, mlp_dims)
        self.net_cfg = dict(gamma=self.gamma, mlp3_dims=mlp_dims, model='cadrl')
        self.multiagent_training = config.getboolean('cadrl', 'multiagent_training')
        logging.info('Policy: CADRL without occupancy map')

    def transform(self, state):
        """Single-human training input (cadrl.py:174-185)."""
        assert len(state.human_states) == 1
        row = torch.Tensor(state.self_state + state.human_states[0]).to(self.device)
        return rotate(row.unsqueeze(0), self.kinematics).squeeze(dim=0)


# CADRL exposes no attention weights: CrowdSim.reset/step probe for the attribute (crowd_sim.py:301-304)
CADRL.get_attention_weights = property(lambda self: (_ for _ in ()).throw(AttributeError('no attention')))
