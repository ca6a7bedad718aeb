"""Batched value-network rollouts: Explorer.run_k_episodes' episode loop (crowd_nav/utils/explorer.py:35-72) for a
robot whose policy is a value network (SARL / CADRL / LSTM-RL), B envs at a time.  Per batched step the host issues
two C-ABI calls — cn_sarl_select (greedy action of every env) and cn_rollout_step (the transition, the episode
bookkeeping and the seeded auto-reset from the scenario ring) — and nothing is synchronised until records are read."""


class SarlRollout(object):
    def __init__(self, eng, gamma, seed_base, seed_mod, episode_limit=-1, env_offset=0, env_stride=None,
                 record_capacity=8):
        self.eng = eng
        eng.set_gamma(gamma)
        # one bookkept step per launch: the transitions are counted per env (ABI v6), so that a launch ends without the
        # hand-off between its workgroups that a job-wide counter costs (~9 us behind a ~20 us cn_rollout_step launch)
        self.bufs = eng.rollout_begin(seed_base=seed_base, seed_mod=seed_mod, episode_limit=episode_limit,
                                      record_capacity=record_capacity, env_offset=env_offset, env_stride=env_stride,
                                      per_env_transitions=True)
        b = self.bufs
        self.rec = dict(outcome=b['ep_outcome'], steps=b['ep_steps'], ret=b['ep_return'], time=b['ep_time'],
                        danger=b['ep_danger'], dsum=b['ep_danger_dmin_sum'])

    @property
    def transitions(self):
        return self.bufs['env_transitions'].sum()

    @property
    def ep_count(self):
        return self.bufs['ep_count']

    def step(self):
        """One transition of every running env."""
        sel = self.eng.sarl_select(want_values=False)
        self.eng.rollout_step(sel['action'])

    def run(self, n_steps):
        for _ in range(int(n_steps)):
            self.step()

    def any_active(self):
        return bool((self.bufs['active'] != 0).any().item())
