"""ctypes binding of libcrowdnav_amd.so (include/crowdnav_amd.h).

The shared library is the product: there is no CPU fallback.  If it has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C crowdnav_amd/csrc`) importing this
module raises, and creating an engine without a visible gfx950 device raises CrowdNavAmdError.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CROWDNAV_AMD_LIB: another build of the same library (kernel A/B experiments, scripts/gpu_ab.sh); default in-tree
LIB_PATH = os.environ.get('CROWDNAV_AMD_LIB') or os.path.join(HERE, 'lib', 'libcrowdnav_amd.so')
ABI_VERSION = 11

CN_OK, CN_ERR_INVALID, CN_ERR_UNSUPPORTED, CN_ERR_HIP, CN_ERR_NO_DEVICE = 0, -1, -2, -3, -4
INFO_NAMES = ('Nothing', 'Danger', 'ReachGoal', 'Collision', 'Timeout')
NOTHING, DANGER, REACH_GOAL, COLLISION, TIMEOUT = range(5)
ROBOT_EXTERNAL, ROBOT_ORCA = 0, 1
CIRCLE_CROSSING, SQUARE_CROSSING, MIXED = 0, 1, 2
HOLONOMIC, UNICYCLE = 0, 1
RECORD_FIELDS, SUMMARY_FIELDS = 6, 8
LAUNCH_COUNTERS = ('rollout_kernels', 'scheduled_kernels', 'ring_fills', 'async_fills', 'sarl_narrow', 'sarl_decide_steps')  # CN_COUNT_*
FLAG_ASYNC_SCENARIO_FILL = 1


class CrowdNavAmdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__('libcrowdnav_amd: %s (status %d)' % (message, status))
        self.status = status


class CnConfig(C.Structure):
    """struct cn_config (include/crowdnav_amd.h)."""
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
        ('randomize_attributes', C.c_int32), ('device', C.c_int32),
        ('robot_kinematics', C.c_int32), ('flags', C.c_int32),
    ]


class CnRolloutIo(C.Structure):
    """struct cn_rollout_io (include/crowdnav_amd.h)."""
    _fields_ = [
        ('seed_base', C.c_uint32), ('seed_mod', C.c_uint32),
        ('episode_limit', C.c_int64), ('env_offset', C.c_int64), ('env_stride', C.c_int64),
        ('record_capacity', C.c_int32),
        ('ep_outcome', C.c_void_p), ('ep_steps', C.c_void_p), ('ep_return', C.c_void_p),
        ('ep_time', C.c_void_p), ('ep_danger', C.c_void_p), ('ep_danger_dmin_sum', C.c_void_p),
        ('ep_count', C.c_void_p), ('cur_steps', C.c_void_p), ('cur_return', C.c_void_p),
        ('cur_danger', C.c_void_p), ('cur_danger_dmin_sum', C.c_void_p),
        ('active', C.c_void_p), ('transitions', C.c_void_p),
        ('summary', C.c_void_p), ('blocks', C.c_void_p), ('blocks_records', C.c_int32),
        ('env_transitions', C.c_void_p),  # ABI v6
    ]


class CnSarlConfig(C.Structure):
    """struct cn_sarl_config (include/crowdnav_amd.h)."""
    _fields_ = [
        ('n_actions', C.c_int32), ('with_om', C.c_int32), ('cell_num', C.c_int32), ('om_channel_size', C.c_int32),
        ('cell_size', C.c_double), ('gamma', C.c_double), ('with_global_state', C.c_int32),
        ('mlp1_dims', C.c_int32 * 2), ('mlp2_dims', C.c_int32 * 2), ('attention_dims', C.c_int32 * 3),
        ('mlp3_dims', C.c_int32 * 4), ('model', C.c_int32), ('interaction_dims', C.c_int32 * 4),
        ('constant_velocity_model', C.c_int32), ('reserved', C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/crowdnav_amd.h declares
_P = C.c_void_p
SYMBOLS = {
    'cn_last_error': (C.c_char_p, []),
    'cn_abi_version': (C.c_int, []),
    'cn_create': (C.c_int, [C.POINTER(CnConfig), C.POINTER(_P)]),
    'cn_destroy': (C.c_int, [_P]),
    'cn_set_stream': (C.c_int, [_P, _P]),
    'cn_sync': (C.c_int, [_P]),
    'cn_set_state': (C.c_int, [_P, _P, _P]),
    'cn_get_state': (C.c_int, [_P, _P, _P]),
    'cn_set_theta': (C.c_int, [_P, _P]),
    'cn_get_theta': (C.c_int, [_P, _P]),
    'cn_get_human_count': (C.c_int, [_P, _P]),
    'cn_drop_robot_sim': (C.c_int, [_P]),
    'cn_drop_sims': (C.c_int, [_P]),
    'cn_set_robot_sim': (C.c_int, [_P, _P, C.c_float]),
    'cn_reset': (C.c_int, [_P, _P, _P, _P]),
    'cn_orca': (C.c_int, [_P, _P]),
    'cn_step': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    'cn_set_gamma': (C.c_int, [_P, C.c_double]),
    'cn_rollout_begin': (C.c_int, [_P, C.POINTER(CnRolloutIo)]),
    'cn_rollout': (C.c_int, [_P, C.POINTER(CnRolloutIo), C.c_int]),
    'cn_rollout_step': (C.c_int, [_P, C.POINTER(CnRolloutIo), _P]),
    'cn_sarl_configure': (C.c_int, [_P, C.POINTER(CnSarlConfig), _P]),
    'cn_sarl_set_weights': (C.c_int, [_P, C.POINTER(_P)]),
    'cn_sarl_select': (C.c_int, [_P, _P, _P, _P]),
    'cn_sarl_explore': (C.c_int, [_P, C.c_double, _P, _P, _P, _P]),
    'cn_sarl_transform': (C.c_int, [_P, _P, C.c_int64, C.c_int]),
    'cn_sarl_sample_step': (C.c_int, [_P, C.c_double, _P, _P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    'cn_sarl_values': (C.c_int, [_P, _P, C.c_int64, _P]),
    'cn_sarl_export': (C.c_int, [_P, C.c_int, _P, C.c_uint64]),
    'cn_rollout_records': (C.c_int, [_P, C.POINTER(CnRolloutIo), C.c_int, _P]),
    'cn_gather_records': (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    'cn_records_summary': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P]),
    'cn_rollout_summary': (C.c_int, [_P, C.POINTER(CnRolloutIo), _P]),
    'cn_launch_counts': (C.c_int, [_P, _P]),
    'cn_mt_random': (C.c_int, [_P, C.c_uint32, C.c_int, _P]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'crowdnav_amd: %s is missing. Build the HIP library first (make -C crowdnav_amd/csrc, or '
            '__graft_entry__.build()); there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.cn_abi_version()
    if got != ABI_VERSION:
        raise ImportError('crowdnav_amd: library ABI %d != binding ABI %d; rebuild' % (got, ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status != CN_OK:
        raise CrowdNavAmdError(status, load().cn_last_error().decode('utf-8', 'replace'))
