"""Batched SARL rollouts: Explorer.run_k_episodes' episode loop (crowd_nav/utils/explorer.py:35-72) for a robot
whose policy is the SARL value network, B envs at a time.  Per batched step the host issues cn_sarl_select (greedy
action of every env), cn_step (the transition) and a masked cn_reset (seeded auto-reset of finished envs); all
bookkeeping is torch arithmetic on the device, nothing is synchronised until records are read."""
import torch

from . import _lib


class SarlRollout(object):
    def __init__(self, eng, gamma, seed_base, seed_mod, episode_limit=-1, env_offset=0, env_stride=None,
                 record_capacity=8):
        self.eng = eng
        B, dev = eng.B, eng.device
        self.seed_base, self.seed_mod, self.limit = int(seed_base), int(seed_mod), int(episode_limit)
        self.offset, self.stride = int(env_offset), int(B if env_stride is None else env_stride)
        self.K = int(record_capacity)
        steps = int(round(eng.config['time_limit'] / eng.config['time_step'])) + 8
        # pow(gamma, t * time_step * v_pref), python floats as explorer.py:71 computes them
        self.discount = torch.tensor([pow(gamma, t * eng.config['time_step'] * eng.config['robot_v_pref'])
                                      for t in range(steps)], dtype=torch.float64, device=dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.env_id = torch.arange(B, device=dev, dtype=torch.int64) + self.offset
        self.ep_count = z((B,), torch.int64)
        self.cur_steps = z((B,), torch.int64)
        self.cur_return = z((B,), torch.float64)
        self.cur_danger = z((B,), torch.int64)
        self.cur_dsum = z((B,), torch.float64)
        self.rec = dict(outcome=z((B, self.K), torch.uint8), steps=z((B, self.K), torch.int64),
                        ret=z((B, self.K), torch.float64), time=z((B, self.K), torch.float64),
                        danger=z((B, self.K), torch.int64), dsum=z((B, self.K), torch.float64))
        self.transitions = z((), torch.int64)
        self.active = self._within_limit(self.env_id)
        self._reset(torch.ones(B, dtype=torch.bool, device=dev) & self.active)

    def _within_limit(self, episode_id):
        if self.limit < 0:
            return torch.ones_like(episode_id, dtype=torch.bool)
        return episode_id < self.limit

    def _reset(self, mask):
        c = self.env_id + self.ep_count * self.stride
        seeds = (self.seed_base + c % self.seed_mod) & 0xffffffff
        seeds = torch.where(seeds >= 2 ** 31, seeds - 2 ** 32, seeds).to(torch.int32)  # uint32 bit pattern
        self.eng.reset_async(seeds.contiguous(), mask.to(torch.uint8).contiguous())

    def step(self):
        """One transition of every active env (inactive envs are stepped too but not counted or recorded)."""
        eng = self.eng
        sel = eng.sarl_select(want_values=False)
        out = eng.step(sel['action'], update=True, want_obs=False)
        act = self.active
        reward, info = out['reward'], out['info'].to(torch.int64)
        done = (out['done'] != 0) & act
        disc = self.discount[self.cur_steps.clamp(max=len(self.discount) - 1)]
        self.cur_return = torch.where(act, self.cur_return + disc * reward, self.cur_return)
        self.cur_steps = self.cur_steps + act.to(torch.int64)
        danger = act & (info == _lib.DANGER)
        self.cur_danger = self.cur_danger + danger.to(torch.int64)
        self.cur_dsum = torch.where(danger, self.cur_dsum + out['dmin'], self.cur_dsum)
        self.transitions = self.transitions + act.sum()
        # episode records
        _, gtime = eng.get_state()
        slot = (self.ep_count % self.K).unsqueeze(1)
        tl = torch.full_like(gtime, float(eng.config['time_limit']))
        vals = dict(outcome=out['info'], steps=self.cur_steps, ret=self.cur_return,
                    time=torch.where(info == _lib.TIMEOUT, tl, gtime), danger=self.cur_danger, dsum=self.cur_dsum)
        for k, v in vals.items():
            cur = self.rec[k].gather(1, slot).squeeze(1)
            self.rec[k].scatter_(1, slot, torch.where(done, v.to(self.rec[k].dtype), cur).unsqueeze(1))
        self.ep_count = self.ep_count + done.to(torch.int64)
        zero_i, zero_f = torch.zeros_like(self.cur_steps), torch.zeros_like(self.cur_return)
        self.cur_steps = torch.where(done, zero_i, self.cur_steps)
        self.cur_return = torch.where(done, zero_f, self.cur_return)
        self.cur_danger = torch.where(done, zero_i, self.cur_danger)
        self.cur_dsum = torch.where(done, zero_f, self.cur_dsum)
        self.active = act & self._within_limit(self.env_id + self.ep_count * self.stride)
        self._reset(done & self.active)
        return out

    def run(self, n_steps):
        for _ in range(int(n_steps)):
            self.step()

    def any_active(self):
        return bool(self.active.any().item())
