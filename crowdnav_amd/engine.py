"""BatchedCrowdSim — tensor-level host API over the C ABI (B independent CrowdSim envs on one GPU).

Mirrors, batched over B envs, the reference calls
    CrowdSim.configure / reset / step / onestep_lookahead   crowd_sim/envs/crowd_sim.py:51-79, 251-420
    ORCA.predict for every agent                            crowd_sim/envs/policy/orca.py:82-132
    Explorer.run_k_episodes' episode loop                   crowd_nav/utils/explorer.py:35-72
torch is used for device memory and streams only; all compute happens in libcrowdnav_amd.so.
Agent 0 of every env is the robot, agents 1..H the humans.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import CnConfig, CnRolloutIo, check

# shipped defaults of crowd_nav/configs/env.config + the hard-coded ORCA constants (orca.py:60-66)
_DEFAULTS = dict(
    num_envs=1, num_humans=5, time_step=0.25, time_limit=25.0, success_reward=1.0,
    collision_penalty=-0.25, discomfort_dist=0.2, discomfort_penalty_factor=0.5, robot_visible=0,
    robot_policy=_lib.ROBOT_ORCA, robot_safety_space=0.0, human_safety_space=0.0, neighbor_dist=10.0,
    max_neighbors=10, scenario_rule=_lib.CIRCLE_CROSSING, time_horizon=5.0, time_horizon_obst=5.0,
    circle_radius=4.0, square_width=10.0, human_radius=0.3, human_v_pref=1.0, robot_radius=0.3,
    robot_v_pref=1.0, randomize_attributes=0, device=0, robot_kinematics=_lib.HOLONOMIC, flags=0)


def default_config(**overrides):
    d = dict(_DEFAULTS)
    unknown = set(overrides) - set(d)
    if unknown:
        raise TypeError('unknown config keys: %s' % sorted(unknown))
    d.update(overrides)
    return d


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedCrowdSim(object):
    """B crowd-navigation envs resident in HBM.  All tensors returned live on the engine's device."""

    def __init__(self, **config):
        self.config = default_config(**config)
        self.B = int(self.config['num_envs'])
        self.H = int(self.config['num_humans'])
        self.A = self.H + 1
        self._lib = _lib.load()
        self._h = C.c_void_p()
        cfg = CnConfig(**self.config)
        check(self._lib.cn_create(C.byref(cfg), C.byref(self._h)))
        self.device = torch.device('cuda', int(self.config['device']))
        self._rollout = None
        self.use_current_stream()

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.cn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # ---------------------------------------------------------------- plumbing
    def use_current_stream(self):
        """Launch on torch's current HIP stream so torch ops and torch.cuda.Event order with the engine."""
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream()
            check(self._lib.cn_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)))

    def sync(self):
        check(self._lib.cn_sync(self._h))

    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _dev(self, x, dtype, shape):
        t = torch.as_tensor(x, dtype=dtype).to(self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError('expected shape %s, got %s' % (tuple(shape), tuple(t.shape)))
        return t

    # ---------------------------------------------------------------- state
    def set_state(self, state8=None, global_time=None):
        """state8: [B, A, 8] float64 (px,py,vx,vy,gx,gy,radius,v_pref); global_time: [B] float64."""
        s = None if state8 is None else self._dev(state8, torch.float64, (self.B, self.A, 8))
        g = None if global_time is None else self._dev(global_time, torch.float64, (self.B,))
        check(self._lib.cn_set_state(self._h, _ptr(s), _ptr(g)))
        self.sync()  # s/g may be temporaries

    def get_state(self):
        s = self._new((self.B, self.A, 8), torch.float64)
        g = self._new((self.B,), torch.float64)
        check(self._lib.cn_get_state(self._h, _ptr(s), _ptr(g)))
        return s, g

    def set_theta(self, theta):
        """Robot heading per env, [B] float64 (FullState.theta)."""
        t = self._dev(theta, torch.float64, (self.B,))
        check(self._lib.cn_set_theta(self._h, _ptr(t)))
        self.sync()

    def get_theta(self):
        t = self._new((self.B,), torch.float64)
        check(self._lib.cn_get_theta(self._h, _ptr(t)))
        return t

    def human_count(self):
        """len(env.humans) per env, [B] int32 (differs from num_humans only under the `mixed` rule)."""
        n = self._new((self.B,), torch.int32)
        check(self._lib.cn_get_human_count(self._h, _ptr(n)))
        return n

    def set_robot_sim(self, radii, max_speed):
        """Every env gets the same captured robot simulator: radii float32 [A] (radius + 0.01 + safety_space as the
        persistent ORCA policy first saw them), max_speed (cn_set_robot_sim)."""
        r = np.ascontiguousarray(radii, dtype=np.float32).reshape(self.A)
        check(self._lib.cn_set_robot_sim(self._h, r.ctypes.data_as(C.c_void_p), C.c_float(float(max_speed))))

    def drop_robot_sim(self):
        check(self._lib.cn_drop_robot_sim(self._h))

    def drop_sims(self):
        """Fresh rvo2 simulators for every agent (cn_drop_sims): the robot's captured radii and every simulator's kd-tree
        order are forgotten — after teleporting the agents with set_state, when freshly built policies are meant."""
        check(self._lib.cn_drop_sims(self._h))

    def reset(self, seeds, mask=None):
        """np.random.seed(seeds[b]) + scenario generation per env; returns the np.random.random() call counts."""
        host = seeds.cpu().numpy() if torch.is_tensor(seeds) else np.asarray(seeds)
        bits = np.ascontiguousarray(host.astype(np.uint32)).view(np.int32)  # torch has no uint32 arithmetic
        sd = self._dev(torch.from_numpy(bits), torch.int32, (self.B,))
        m = None if mask is None else self._dev(mask, torch.uint8, (self.B,))
        draws = torch.zeros(self.B, dtype=torch.int64, device=self.device)
        check(self._lib.cn_reset(self._h, _ptr(sd), _ptr(m), _ptr(draws)))
        self.sync()
        return draws

    def reset_async(self, seeds_i32, mask_u8):
        """cn_reset with device-resident seeds (int32 bit patterns of the uint32 seeds) and mask; no sync."""
        check(self._lib.cn_reset(self._h, _ptr(seeds_i32), _ptr(mask_u8), None))

    # ---------------------------------------------------------------- one transition
    def orca(self):
        out = self._new((self.B, self.A, 2), torch.float32)
        check(self._lib.cn_orca(self._h, _ptr(out)))
        return out

    def step(self, action=None, update=True, want_obs=True):
        """CrowdSim.step for every env.  action: [B, 2] float64 or None when the robot is ORCA on device."""
        a = None if action is None else self._dev(action, torch.float64, (self.B, 2))
        out = dict(
            reward=self._new((self.B,), torch.float64), done=self._new((self.B,), torch.uint8),
            info=self._new((self.B,), torch.uint8), dmin=self._new((self.B,), torch.float64),
            action=self._new((self.B, 2), torch.float64),
            orca_vel=self._new((self.B, self.A, 2), torch.float32),
            obs=self._new((self.B, self.H, 5), torch.float64) if want_obs else None)
        check(self._lib.cn_step(self._h, _ptr(a), int(bool(update)), _ptr(out['reward']), _ptr(out['done']),
                                _ptr(out['info']), _ptr(out['dmin']), _ptr(out['action']),
                                _ptr(out['orca_vel']), _ptr(out['obs'])))
        if a is not None and not (torch.is_tensor(action) and a.data_ptr() == action.data_ptr()):
            self.sync()  # `a` is a temporary copy
        return out

    def step_into(self, action, reward, done, info, dmin, update=True):
        """CrowdSim.step for every env, results written into the caller's device tensors (reward / dmin float64 [B], done / info
        uint8 [B]; e.g. row t of a [T, B] history) — nothing allocated, no optional outputs: the per-step form of the RL
        sampling loop, where a dozen small torch kernels per step were as expensive as the step itself."""
        for t, dt in ((action, torch.float64), (reward, torch.float64), (dmin, torch.float64), (done, torch.uint8), (info, torch.uint8)):
            assert t.dtype == dt and t.is_contiguous() and t.device.type == self.device.type
        check(self._lib.cn_step(self._h, _ptr(action), int(bool(update)), _ptr(reward), _ptr(done), _ptr(info), _ptr(dmin),
                                None, None, None))

    # ---------------------------------------------------------------- fused rollouts
    def set_gamma(self, gamma):
        check(self._lib.cn_set_gamma(self._h, float(gamma)))

    def rollout_begin(self, seed_base, seed_mod, episode_limit=-1, record_capacity=8, env_offset=0,
                      env_stride=None, boundary_records=0, per_env_transitions=False):
        """Start Explorer-style episode bookkeeping: env b runs global episodes b, b+B, b+2B, ... (< limit),
        episode c seeded with seed_base + c % seed_mod.  A shard of a larger job passes its first global env
        id as env_offset and the global env count as env_stride.
        boundary_records = K > 0: every rollout launch also leaves the shard-boundary outputs itself (its last workgroup):
        bufs['summary'] float64 [8] (what records_summary() returns for this engine) and bufs['blocks'] float64
        [B, 1 + 6 K] (what rollout_records(K) returns) — no boundary kernels on a single GPU, only the all-gather on many.
        per_env_transitions (ABI v6): instead of the ONE job-wide counter bufs['transitions'] every launch adds up (arrival
        tickets between the workgroups at the end of every launch), each env counts its own transitions in
        bufs['env_transitions'] int64 [B] and a launch ends without any hand-off — with boundary_records = 0 the statistics are
        then computed once, when the caller asks (rollout_records + records_summary), as explorer.py:74-90 does."""
        B, K = self.B, int(record_capacity)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)  # noqa: E731
        bufs = dict(
            ep_outcome=z((B, K), torch.uint8), ep_steps=z((B, K), torch.int32), ep_return=z((B, K), torch.float64),
            ep_time=z((B, K), torch.float64), ep_danger=z((B, K), torch.int32),
            ep_danger_dmin_sum=z((B, K), torch.float64),
            ep_count=z((B,), torch.int32), cur_steps=z((B,), torch.int32), cur_return=z((B,), torch.float64),
            cur_danger=z((B,), torch.int32), cur_danger_dmin_sum=z((B,), torch.float64),
            active=z((B,), torch.uint8))
        if per_env_transitions:
            bufs['env_transitions'] = z((B,), torch.int64)
        else:
            bufs['transitions'] = z((1,), torch.int64)
        Kb = int(boundary_records)
        if Kb > 0:
            bufs['summary'] = z((_lib.SUMMARY_FIELDS,), torch.float64)
            bufs['blocks'] = z((B, 1 + Kb * _lib.RECORD_FIELDS), torch.float64)
        io = CnRolloutIo(seed_base=int(seed_base), seed_mod=int(seed_mod), episode_limit=int(episode_limit),
                         env_offset=int(env_offset), env_stride=int(B if env_stride is None else env_stride),
                         record_capacity=K, blocks_records=Kb, **{k: v.data_ptr() for k, v in bufs.items()})
        # the previous rollout's buffers stay referenced until cn_rollout_begin has returned: with the asynchronous scenario
        # fill, kernels of the old rollout may still be reading them on the engine's side streams (the call waits for them)
        previous = self._rollout
        check(self._lib.cn_rollout_begin(self._h, C.byref(io)))
        self._rollout = (io, bufs)
        del previous
        return bufs

    def rollout(self, n_steps):
        """n_steps transitions per active env in one kernel launch (in-kernel auto-reset)."""
        if self._rollout is None:
            raise RuntimeError('call rollout_begin() first')
        check(self._lib.cn_rollout(self._h, C.byref(self._rollout[0]), int(n_steps)))
        return self._rollout[1]

    def rollout_step(self, action):
        """One bookkept transition of every running env with caller-supplied actions ([B, 2] float64 device tensor):
        the external-policy counterpart of rollout() (robot_policy == ROBOT_EXTERNAL)."""
        if self._rollout is None:
            raise RuntimeError('call rollout_begin() first')
        a = self._dev(action, torch.float64, (self.B, 2))
        check(self._lib.cn_rollout_step(self._h, C.byref(self._rollout[0]), _ptr(a)))
        if not (torch.is_tensor(action) and a.data_ptr() == action.data_ptr()):
            self.sync()
        return self._rollout[1]

    # ---------------------------------------------------------------- shard boundary (explorer.py:74-90)
    def rollout_records(self, max_records=None):
        """The finished episodes of this engine's rollout as ONE self-contained float64 block per env:
        [B, 1 + 6 K] = (episodes finished, K x (outcome, steps, discounted return, nav time, danger steps, danger dmin
        sum)); see distributed.split_blocks.  One kernel (cn_rollout_records)."""
        if self._rollout is None:
            raise RuntimeError('call rollout_begin() first')
        io, bufs = self._rollout
        K = int(bufs['ep_outcome'].shape[1] if max_records is None else max_records)
        blocks = self._new((self.B, 1 + K * _lib.RECORD_FIELDS), torch.float64)
        check(self._lib.cn_rollout_records(self._h, C.byref(io), K, _ptr(blocks)))
        return blocks

    def records_summary(self, blocks, record_capacity=None, out=None):
        """float64 [8]: episodes finished, records held, ReachGoal / Collision / Timeout among them, sum of successful nav
        times, sum of discounted returns, sum of Danger steps — of any [n, 1 + 6 K] record blocks, a shard's own or the
        gathered ones (cn_records_summary: one workgroup, fixed summation order)."""
        K = (int(blocks.shape[1]) - 1) // _lib.RECORD_FIELDS
        cap = K if record_capacity is None else int(record_capacity)
        out = self._summary_out(out)
        check(self._lib.cn_records_summary(self._h, int(blocks.shape[0]), K, cap, _ptr(blocks), _ptr(out)))
        return out

    def _summary_out(self, out):
        if out is None:
            return self._new((_lib.SUMMARY_FIELDS,), torch.float64)
        if out.dtype != torch.float64 or out.device != self.device or out.numel() != _lib.SUMMARY_FIELDS or not out.is_contiguous():
            raise ValueError('summary output: a contiguous float64 [%d] tensor on %s' % (_lib.SUMMARY_FIELDS, self.device))
        return out

    def rollout_summary(self, out=None):
        """records_summary(rollout_records()) of this engine in ONE kernel, straight from its record rings (cn_rollout_summary):
        the statistics of explorer.py:74-90 for a run on one engine.  out: the caller's float64 [8] device tensor (a run that
        asks once per boundary keeps one: no allocation between the last launch and the kernel)."""
        if self._rollout is None:
            raise RuntimeError('call rollout_begin() first')
        out = self._summary_out(out)
        check(self._lib.cn_rollout_summary(self._h, C.byref(self._rollout[0]), _ptr(out)))
        return out

    def gather_records_rccl(self, comm, n_ranks, blocks):
        """All-gather the record blocks of every rank over the caller's RCCL communicator (`comm`: the ncclComm_t as an
        integer, see crowdnav_amd.rccl) on the engine's stream (cn_gather_records): [n_ranks * B, 1 + 6 K]."""
        K = (int(blocks.shape[1]) - 1) // _lib.RECORD_FIELDS
        out = self._new((n_ranks * self.B, int(blocks.shape[1])), torch.float64)
        check(self._lib.cn_gather_records(self._h, C.c_void_p(int(comm)), int(n_ranks), K, _ptr(blocks), _ptr(out)))
        return out

    def launch_counts(self):
        """What the host has enqueued for this engine since it was created (cn_launch_counts): dict of _lib.LAUNCH_COUNTERS ->
        int.  Host-side, no device work: the difference of two calls says which launches a region contained."""
        out = (C.c_uint64 * len(_lib.LAUNCH_COUNTERS))()
        check(self._lib.cn_launch_counts(self._h, out))
        return dict(zip(_lib.LAUNCH_COUNTERS, (int(v) for v in out)))

    def mt_random(self, seed, n):
        out = self._new((n,), torch.float64)
        check(self._lib.cn_mt_random(self._h, int(seed), int(n), _ptr(out)))
        return out


# ---------------------------------------------------------------------------------------- SARL decision
SARL_PARAM_ORDER = tuple('%s.%d.%s' % (m, i, p) for m, idxs in (('mlp1', (0, 2)), ('mlp2', (0, 2)),
                                                               ('attention', (0, 2, 4)), ('mlp3', (0, 2, 4, 6)))
                         for i in idxs for p in ('weight', 'bias'))


LSTM_PARAM_ORDER = tuple('mlp.%d.%s' % (i, p) for i in (0, 2, 4, 6) for p in ('weight', 'bias')) + (
    'lstm.weight_ih_l0', 'lstm.weight_hh_l0', 'lstm.bias_ih_l0', 'lstm.bias_hh_l0')
LSTM_PAIRWISE_MLP1_ORDER = tuple('mlp1.%d.%s' % (i, p) for i in (0, 2, 4, 6) for p in ('weight', 'bias'))
CADRL_PARAM_ORDER = tuple('value_network.%d.%s' % (i, p) for i in (0, 2, 4, 6) for p in ('weight', 'bias'))


def _sarl_configure(self, actions, gamma=0.9, with_om=False, cell_num=4, cell_size=1.0, om_channel_size=3,
                    with_global_state=True, mlp1_dims=(150, 100), mlp2_dims=(100, 50), attention_dims=(100, 100, 1),
                    mlp3_dims=(150, 100, 100, 1), model='sarl', interaction_dims=None, query_env=True):
    """SARL.configure (or model='cadrl': mlp3_dims = [cadrl] mlp_dims; model='lstm_rl': mlp1_dims[0] = hidden width,
    mlp3_dims = [lstm_rl] mlp2_dims, interaction_dims = [lstm_rl] mlp1_dims when with_interaction_module) +
    build_action_space for
    this engine.  actions: [K, 2] float64 ActionXY table (host).  query_env=False: [action_space] query_env = false, the
    constant-velocity human model + MultiHumanRL.compute_reward instead of the env's lookahead."""
    acts = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(-1, 2))
    cfg = _lib.CnSarlConfig(n_actions=len(acts), with_om=int(bool(with_om)), cell_num=int(cell_num),
                            om_channel_size=int(om_channel_size), cell_size=float(cell_size), gamma=float(gamma),
                            with_global_state=int(bool(with_global_state)),
                            mlp1_dims=(C.c_int32 * 2)(*mlp1_dims), mlp2_dims=(C.c_int32 * 2)(*mlp2_dims),
                            attention_dims=(C.c_int32 * 3)(*attention_dims), mlp3_dims=(C.c_int32 * 4)(*mlp3_dims),
                            model={'sarl': 0, 'cadrl': 1, 'lstm_rl': 2}[model],
                            interaction_dims=(C.c_int32 * 4)(*(interaction_dims or (0, 0, 0, 0))),
                            constant_velocity_model=0 if query_env else 1, reserved=0)
    check(self._lib.cn_sarl_configure(self._h, C.byref(cfg), acts.ctypes.data_as(C.c_void_p)))
    self.sarl = dict(n_actions=len(acts), in_dim=13 + (cell_num ** 2 * om_channel_size if with_om else 0),
                     actions=acts, model=model, pairwise=bool(interaction_dims))


def _sarl_set_weights(self, state_dict):
    """Hand the value network's parameters (sarl.ValueNetwork.state_dict()) to the device kernels."""
    order = {'cadrl': CADRL_PARAM_ORDER, 'lstm_rl': LSTM_PARAM_ORDER}.get(self.sarl['model'], SARL_PARAM_ORDER)
    if self.sarl.get('pairwise'):  # lstm_rl.ValueNetwork2: mlp1 first, as in its state_dict
        order = LSTM_PAIRWISE_MLP1_ORDER + order
    # parameters that live on the engine's device as dense float32 are read in place (the RL phase uploads weights once per
    # sampled episode: 22 conversions and stream records were ~0.1 ms of host time each time); anything else goes through a copy
    tensors, temporaries = [], []
    for k in order:
        t = state_dict[k]
        if not (t.device == self.device and t.dtype == torch.float32 and t.is_contiguous()):
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            temporaries.append(t)
        tensors.append(t)
    ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    check(self._lib.cn_sarl_set_weights(self._h, ptrs))
    # the tensors above must outlive the repack kernel: kept until the next call — by then it is long behind on the engine's
    # stream; no synchronize per call.  The caching allocator is told that the engine's stream reads the TEMPORARIES: were the
    # caller inside another torch.cuda.stream(...) context, a freed temporary could otherwise be handed out again on THAT stream
    # while the repack kernel, on the stream captured at construction / use_current_stream(), still reads it.
    for t in temporaries:
        t.record_stream(self._stream)
    self._sarl_weights_keepalive = tensors


def _sarl_select(self, want_values=True, best=None, action=None):
    """Greedy SARL action of every env: dict(values [B,K] f64, best [B] i32, action [B,2] f64).  best / action: the caller's
    device tensors to write into (e.g. row t of an action history) instead of fresh ones."""
    K = self.sarl['n_actions']
    if best is None:
        best = self._new((self.B,), torch.int32)
    if action is None:
        action = self._new((self.B, 2), torch.float64)
    assert best.dtype == torch.int32 and best.is_contiguous() and action.dtype == torch.float64 and action.is_contiguous()
    out = dict(values=self._new((self.B, K), torch.float64) if want_values else None, best=best, action=action)
    check(self._lib.cn_sarl_select(self._h, _ptr(out['values']), _ptr(out['best']), _ptr(out['action'])))
    return out


def _sarl_export(self, name):
    """Internal buffers of the last sarl_select (tests): reward, V, next_obs, om, X (natural [B,K,H,ld] order)."""
    K, H = self.sarl['n_actions'], self.H
    if name == 'reward':
        t, which = self._new((self.B, K), torch.float64), 0
    elif name == 'V':
        t, which = self._new((self.B, K), torch.float32), 1
    elif name == 'next_obs':
        t, which = self._new((self.B, H, 5), torch.float64), 2
    elif name == 'om':
        t, which = self._new((self.B, H, self.sarl['in_dim'] - 13), torch.float32), 3
    elif name == 'X':
        d = self.sarl['in_dim']  # cn::sarl_ks: whole 16-column tiles, at least the k loop's 5-step chunks
        ks = max((d + 15) // 16 * 4, ((d + 3) // 4 + 4) // 5 * 5)
        tiles = (self.B * K + 15) // 16
        t, which = self._new((tiles, H, ks, 4, 16), torch.float32), 4  # MFMA A-fragment order
    else:
        raise KeyError(name)
    check(self._lib.cn_sarl_export(self._h, which, _ptr(t), t.numel() * t.element_size()))
    if name == 'X':  # [tile][h][k-step][k % 4][g] -> [B, K, H, in_dim]
        t = t.permute(0, 4, 1, 2, 3).reshape(-1, H, t.shape[2] * 4)[:self.B * K, :, :self.sarl['in_dim']]
        t = t.reshape(self.B, K, H, -1)
    return t


def _sarl_explore(self, sel, epsilon, mask=None, want_explored=True):
    """Epsilon-greedy on top of a sarl_select result (in place): each env draws np.random.random() — and, below
    epsilon, np.random.choice(K) — from the numpy stream its last reset() seeded (multi_human_rl.py:28-31)."""
    m = None if mask is None else self._dev(mask, torch.uint8, (self.B,))
    explored = self._new((self.B,), torch.uint8) if want_explored else None
    check(self._lib.cn_sarl_explore(self._h, float(epsilon), _ptr(m), _ptr(sel['best']), _ptr(sel['action']),
                                    _ptr(explored)))
    sel['explored'] = explored
    return sel


def _sarl_transform(self, out=None, env_stride=0, sort_humans=None):
    """MultiHumanRL.transform of every env's current joint state: [B, H, in_dim] float32 (replay-memory state).
    With out (a float32 device tensor) and env_stride (floats between consecutive envs) the rows are written in place,
    e.g. out = traj[:, t] of a [B, T, H, D] trajectory tensor with env_stride = T * H * D.  sort_humans: LSTM-RL's
    decreasing-distance order (default for that model: the RL phase); False = env order (imitation learning)."""
    if out is None:
        out = self._new((self.B, self.H, self.sarl['in_dim']), torch.float32)
    assert out.dtype == torch.float32 and out.device.type == self.device.type
    if sort_humans is None:  # what a train-phase predict() of this model leaves in last_state
        sort_humans = self.sarl['model'] == 'lstm_rl'
    check(self._lib.cn_sarl_transform(self._h, C.c_void_p(out.data_ptr()), int(env_stride), int(bool(sort_humans))))
    return out


def _sarl_sampler(self, traj, rew, info, dmin, act, alive, done, action):
    """The train-phase decision + step of every env as ONE call per step: step(t, epsilon) is cn_sarl_sample_step —
    alive &= ~done (the previous step's episode ends; zero `done` before an episode's first step), then cn_sarl_select ->
    cn_sarl_explore (mask = alive) -> cn_sarl_transform -> cn_step — with row t of the caller's histories (traj [B, T, H, D]
    float32; rew / dmin [T, B] float64; info [T, B] uint8; act [T, B] int32: the chosen action index) as outputs, addresses
    precomputed.  For a few envs a streamed loop of these calls is two launches per step (include/crowdnav_amd.h); on that
    route the kernels skip an env whose episode is over (its rows of the histories are not written any more)."""
    B, T, H, D = traj.shape
    if B != self.B or H != self.H:
        raise ValueError('traj is [%d, T, %d, D]; the engine holds %d envs x %d humans' % (B, H, self.B, self.H))
    specs = ((traj, torch.float32, (B, T, H, D)), (act, torch.int32, (T, B)), (rew, torch.float64, (T, B)),
             (dmin, torch.float64, (T, B)), (info, torch.uint8, (T, B)), (alive, torch.uint8, (B,)), (done, torch.uint8, (B,)),
             (action, torch.float64, (B, 2)))
    for t_, dt, shape in specs:  # raw addresses go to the C ABI below: every tensor on the engine's device, dense, as declared
        # (info may live in PINNED host memory instead: the kernels only write it — posted stores over the host link — and a
        # caller that streams steps can watch the episode-end codes arrive without a device synchronisation)
        # (round 6, later: so may the reward / min-distance / action histories — written before the step's info code on the
        # two-launch route; on the other routes a kernel READS the action row back: correct over the host link, only slower)
        on_device = t_.device == self.device or (any(t_ is w for w in (info, rew, dmin, act)) and t_.device.type == 'cpu'
                                                 and t_.is_pinned())
        if not on_device or t_.dtype != dt or tuple(t_.shape) != shape or not t_.is_contiguous():
            raise ValueError('sarl_sampler: expected a contiguous %s %s tensor on %s, got %s %s on %s (contiguous: %s)'
                             % (dt, shape, self.device, t_.dtype, tuple(t_.shape), t_.device, t_.is_contiguous()))
    # the closure owns the tensors its addresses point into, but only a WEAK reference to the engine: stored on the engine or on
    # a rollout object it would otherwise form a cycle, and cn_destroy (a stream synchronize + hipFree) would run whenever the
    # cyclic collector gets to it — e.g. in the middle of somebody's timed region (bench.py closes its engines explicitly)
    keep = (traj, rew, info, dmin, act, alive, done, action)
    alive_engine = weakref.ref(self)
    lib, h, V = self._lib, self._h, C.c_void_p
    p_traj, p_rew, p_inf, p_dmn, p_act = traj.data_ptr(), rew.data_ptr(), info.data_ptr(), dmin.data_ptr(), act.data_ptr()
    p_alive, p_done, p_action = V(alive.data_ptr()), V(done.data_ptr()), V(action.data_ptr())
    sort = int(self.sarl['model'] == 'lstm_rl')
    stride = T * H * D

    def step(t, epsilon):
        eng = alive_engine()
        if eng is None or eng._h is not h or not h.value:
            raise RuntimeError('sarl_sampler: the engine has been closed')
        check(lib.cn_sarl_sample_step(h, epsilon, p_alive, V(p_act + 4 * B * t), p_action, V(p_traj + 4 * H * D * t), stride, sort,
                                      V(p_rew + 8 * B * t), p_done, V(p_inf + B * t), V(p_dmn + 8 * B * t)))
    step.keep = keep  # the tensors live as long as the step function does
    return step


def _sarl_values(self, states, out=None):
    """V(state) of joint states the caller holds (cn_sarl_values): float32 [n, H, 13] rows as sarl_transform / the sampler
    write them -> float32 [n], under the weights of the last sarl_set_weights.  One launch on the narrow tiles; the engine's
    env state is not touched.  What target_model(next_states) is to Explorer.update_memory's TD targets."""
    n = int(states.shape[0])
    if (states.device != self.device or states.dtype != torch.float32 or tuple(states.shape[1:]) != (self.H, 13)
            or not states.is_contiguous()):
        raise ValueError('sarl_values: expected a contiguous float32 [n, %d, 13] tensor on %s' % (self.H, self.device))
    if out is None:
        out = self._new((n,), torch.float32)
    check(self._lib.cn_sarl_values(self._h, _ptr(states), n, _ptr(out)))
    return out


BatchedCrowdSim.sarl_values = _sarl_values
BatchedCrowdSim.sarl_sampler = _sarl_sampler
BatchedCrowdSim.sarl_configure = _sarl_configure
BatchedCrowdSim.sarl_set_weights = _sarl_set_weights
BatchedCrowdSim.sarl_select = _sarl_select
BatchedCrowdSim.sarl_export = _sarl_export
BatchedCrowdSim.sarl_explore = _sarl_explore
BatchedCrowdSim.sarl_transform = _sarl_transform
