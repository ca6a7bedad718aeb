"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/_build/libcrowd_oracle.so (crowd_oracle.cpp), the batched CPU restatement
of the reference hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (crowdnav_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libcrowd_oracle.so')

INFO_NAMES = ('Nothing', 'Danger', 'ReachGoal', 'Collision', 'Timeout')


class CoConfig(C.Structure):
    _fields_ = [
        ('num_envs', C.c_int32), ('num_humans', C.c_int32),
        ('time_step', C.c_double), ('time_limit', C.c_double),
        ('success_reward', C.c_double), ('collision_penalty', C.c_double),
        ('discomfort_dist', C.c_double), ('discomfort_penalty_factor', C.c_double),
        ('robot_visible', C.c_int32), ('robot_policy', C.c_int32),
        ('robot_safety_space', C.c_double), ('human_safety_space', C.c_double),
        ('neighbor_dist', C.c_double),
        ('max_neighbors', C.c_int32), ('scenario_rule', C.c_int32),
        ('time_horizon', C.c_double), ('time_horizon_obst', C.c_double),
        ('circle_radius', C.c_double), ('square_width', C.c_double),
        ('human_radius', C.c_double), ('human_v_pref', C.c_double),
        ('robot_radius', C.c_double), ('robot_v_pref', C.c_double),
        ('randomize_attributes', C.c_int32), ('robot_kinematics', C.c_int32),
    ]


DEFAULTS = dict(
    num_envs=1, num_humans=5, time_step=0.25, time_limit=25.0, success_reward=1.0,
    collision_penalty=-0.25, discomfort_dist=0.2, discomfort_penalty_factor=0.5, robot_visible=0,
    robot_policy=1, robot_safety_space=0.0, human_safety_space=0.0, neighbor_dist=10.0,
    max_neighbors=10, scenario_rule=0, time_horizon=5.0, time_horizon_obst=5.0, circle_radius=4.0,
    square_width=10.0, human_radius=0.3, human_v_pref=1.0, robot_radius=0.3, robot_v_pref=1.0,
    randomize_attributes=0, robot_kinematics=0)


def build(quiet=True):
    """Compile the oracle (g++) if the shared object is missing or stale."""
    srcs = [os.path.join(HERE, f) for f in ('crowd_oracle.cpp', 'rvo2_oracle.cpp', 'rvo2_oracle.hpp')]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(['make', '-C', HERE, '_build/libcrowd_oracle.so'],
                          stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.co_create.restype = C.c_void_p
        L.co_create.argtypes = [C.POINTER(CoConfig)]
        L.co_destroy.argtypes = [C.c_void_p]
        L.co_set_gamma.argtypes = [C.c_void_p, C.c_double]
        L.co_set_threads.argtypes = [C.c_int]
        L.co_max_threads.restype = C.c_int
        L.co_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.co_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.co_drop_sims.argtypes = [C.c_void_p]
        L.co_set_theta.argtypes = [C.c_void_p, C.c_void_p]
        L.co_get_theta.argtypes = [C.c_void_p, C.c_void_p]
        L.co_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.co_orca.argtypes = [C.c_void_p, C.c_void_p]
        L.co_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.co_rollout.restype = C.c_int64
        L.co_rollout.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 7
        L.co_rollout_full.restype = C.c_int64
        L.co_rollout_full.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 12
        L.co_mt_random.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def mt_random(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().co_mt_random(int(seed), n, _p(out))
    return out


class CrowdOracle(object):
    """Batched CPU oracle.  State fields are [B, A] float64 arrays, agent 0 = robot."""

    FIELDS = ('px', 'py', 'vx', 'vy', 'gx', 'gy', 'radius', 'v_pref')

    def __init__(self, **kw):
        d = dict(DEFAULTS)
        unknown = set(kw) - set(d)
        if unknown:
            raise TypeError('unknown config keys: %s' % sorted(unknown))
        d.update(kw)
        self.config = d
        self.cfg = CoConfig(**d)
        self.B, self.H = d['num_envs'], d['num_humans']
        self.A = self.H + 1
        self._h = C.c_void_p(lib().co_create(C.byref(self.cfg)))

    def __del__(self):
        if getattr(self, '_h', None):
            lib().co_destroy(self._h)
            self._h = None

    def set_state(self, state, global_time=None):
        """state: [B, A, 8] (px,py,vx,vy,gx,gy,radius,v_pref)."""
        st = np.ascontiguousarray(state, dtype=np.float64).reshape(self.B, self.A, 8)
        cols = [np.ascontiguousarray(st[:, :, k]) for k in range(8)]
        gt = None if global_time is None else np.ascontiguousarray(global_time, dtype=np.float64)
        lib().co_set_state(self._h, *[_p(c) for c in cols], _p(gt))

    def get_state(self):
        cols = [np.empty((self.B, self.A), dtype=np.float64) for _ in range(8)]
        gt = np.empty(self.B, dtype=np.float64)
        lib().co_get_state(self._h, *[_p(c) for c in cols], _p(gt))
        return np.stack(cols, axis=2), gt

    def human_count(self):
        """len(env.humans) per env: humans the `mixed` rule left out are parked at x >= 1e6 behind the present ones."""
        return (self.get_state()[0][:, 1:, 0] < 5.0e5).sum(axis=1).astype(np.int32)

    def set_theta(self, theta):
        t = np.ascontiguousarray(theta, dtype=np.float64).reshape(self.B)
        lib().co_set_theta(self._h, _p(t))

    def get_theta(self):
        t = np.empty(self.B, dtype=np.float64)
        lib().co_get_theta(self._h, _p(t))
        return t

    def drop_sims(self):
        lib().co_drop_sims(self._h)

    def reset(self, seeds, mask=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        draws = np.zeros(self.B, dtype=np.uint64)
        lib().co_reset(self._h, _p(seeds), _p(m), _p(draws))
        return draws

    def orca(self):
        out = np.empty((self.B, self.A, 2), dtype=np.float32)
        lib().co_orca(self._h, _p(out))
        return out

    def step(self, action=None, update=True):
        a = None if action is None else np.ascontiguousarray(action, dtype=np.float64).reshape(self.B, 2)
        reward = np.empty(self.B, dtype=np.float64)
        done = np.empty(self.B, dtype=np.uint8)
        info = np.empty(self.B, dtype=np.uint8)
        dmin = np.empty(self.B, dtype=np.float64)
        act = np.empty((self.B, 2), dtype=np.float64)
        vel = np.empty((self.B, self.A, 2), dtype=np.float32)
        lib().co_step(self._h, _p(a), int(bool(update)), _p(reward), _p(done), _p(info), _p(dmin),
                      _p(act), _p(vel))
        return dict(reward=reward, done=done, info=info, dmin=dmin, action=act, orca_vel=vel)

    def rollout(self, n_steps, seed_base, seed_mod, max_ep, ep_index, cur_steps, cur_return):
        B = self.B
        ep_count = np.zeros(B, dtype=np.int32)
        ep_outcome = np.zeros((B, max_ep), dtype=np.uint8)
        ep_steps = np.zeros((B, max_ep), dtype=np.int32)
        ep_return = np.zeros((B, max_ep), dtype=np.float64)
        total = lib().co_rollout(self._h, int(n_steps), int(seed_base), int(seed_mod), int(max_ep),
                                 _p(ep_count), _p(ep_outcome), _p(ep_steps), _p(ep_return),
                                 _p(ep_index), _p(cur_steps), _p(cur_return))
        return total, dict(count=ep_count, outcome=ep_outcome, steps=ep_steps, ret=ep_return)

    def rollout_full(self, n_steps, seed_base, seed_mod, max_ep, ep_index=None, cur=None):
        """rollout() with everything Explorer.run_k_episodes keeps per episode (explorer.py:46-62): + nav time, Danger steps,
        sum of their min_dist; `cur` (dict steps / ret / danger / dsum, in/out) carries the running episode between calls."""
        B = self.B
        z = lambda shape, dt: np.zeros(shape, dtype=dt)  # noqa: E731
        rec = dict(count=z(B, np.int32), outcome=z((B, max_ep), np.uint8), steps=z((B, max_ep), np.int32),
                   ret=z((B, max_ep), np.float64), time=z((B, max_ep), np.float64), danger=z((B, max_ep), np.int32),
                   dsum=z((B, max_ep), np.float64))
        ep_index = z(B, np.int32) if ep_index is None else ep_index
        cur = cur or dict(steps=z(B, np.int32), ret=z(B, np.float64), danger=z(B, np.int32), dsum=z(B, np.float64))
        total = lib().co_rollout_full(self._h, int(n_steps), int(seed_base), int(seed_mod), int(max_ep), _p(rec['count']),
                                      _p(rec['outcome']), _p(rec['steps']), _p(rec['ret']), _p(rec['time']), _p(rec['danger']),
                                      _p(rec['dsum']), _p(ep_index), _p(cur['steps']), _p(cur['ret']), _p(cur['danger']),
                                      _p(cur['dsum']))
        return total, rec, cur

    @staticmethod
    def set_threads(n):
        lib().co_set_threads(int(n))

    @staticmethod
    def max_threads():
        return lib().co_max_threads()
