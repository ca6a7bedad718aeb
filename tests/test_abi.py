"""C-ABI surface checks that need no GPU: the in-tree library loads, exports every symbol
include/crowdnav_amd.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'crowdnav_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cn_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    ge.build()
    from crowdnav_amd import _lib
    return _lib


def test_header_and_binding_agree(built):
    declared = _declared_symbols()
    assert declared, 'no cn_* declarations found in the header'
    assert sorted(built.SYMBOLS) == declared


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(built.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert built.load().cn_abi_version() == built.ABI_VERSION


def test_struct_layouts_match_header(built):
    # cn_config: 2 i32, 6 f64, 2 i32, 3 f64, 2 i32, 8 f64, 4 i32 -> 25 x 8 bytes, no padding surprises
    assert C.sizeof(built.CnConfig) == 8 + 48 + 8 + 24 + 8 + 64 + 16
    assert built.CnConfig.time_step.offset == 8 and built.CnConfig.neighbor_dist.offset == 80
    assert built.CnConfig.device.offset == C.sizeof(built.CnConfig) - 12
    assert built.CnConfig.robot_kinematics.offset == C.sizeof(built.CnConfig) - 8
    assert C.sizeof(built.CnRolloutIo) == 8 + 8 + 8 + 8 + 8 + 13 * 8 + 2 * 8 + 8 + 8  # v5: + summary, blocks, blocks_records (+ pad); v6: + env_transitions
    assert built.CnRolloutIo.env_transitions.offset == 168
    assert built.CnRolloutIo.summary.offset == 144 and built.CnRolloutIo.blocks_records.offset == 160
    assert C.sizeof(built.CnSarlConfig) == 16 + 16 + 4 + 8 + 8 + 12 + 16 + 4 + 16 + 4 + 8  # 112: ints, 2 doubles, dims, pad, model flags


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    import crowdnav_amd
    with pytest.raises(crowdnav_amd.CrowdNavAmdError) as ei:
        crowdnav_amd.BatchedCrowdSim(num_envs=4)
    assert ei.value.status == built.CN_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, 'crowdnav_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'crowd_oracle' not in text and 'rvo2_oracle' not in text, os.path.join(dirpath, f)
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
                # ... nor the unmodified reference that build() copies to oracle/_ref/ as test infrastructure
                # (comments cite /root/reference file:line; no code may open, import or put on sys.path anything under it)
                assert 'oracle/_ref' not in text and "'_ref'" not in text and 'ref_harness' not in text, os.path.join(dirpath, f)
                assert not re.search(r'(sys\.path|import_module|run_path|open\(|os\.path|chdir)[^\n]*/root/reference', text), f


def test_create_rejects_bad_configs_before_touching_the_device(built):
    """Error convention (include/crowdnav_amd.h): negative cn_status + cn_last_error(), nothing thrown across the
    ABI.  Config validation happens before the device is probed, so it is testable without a GPU."""
    lib = built.load()
    from crowdnav_amd.engine import default_config

    def create(**kw):
        cfg = built.CnConfig(**default_config(**kw))
        h = C.c_void_p()
        rc = lib.cn_create(C.byref(cfg), C.byref(h))
        return rc, lib.cn_last_error().decode()

    rc, msg = create(num_envs=0)
    assert rc == built.CN_ERR_INVALID and 'num_envs' in msg
    rc, msg = create(num_humans=64)
    assert rc == built.CN_ERR_UNSUPPORTED and 'num_humans' in msg
    rc, msg = create(max_neighbors=11)
    assert rc == built.CN_ERR_UNSUPPORTED and 'max_neighbors' in msg
    rc, msg = create(time_step=0.0)
    assert rc == built.CN_ERR_INVALID and 'time_step' in msg
    rc, msg = create(scenario_rule=3)
    assert rc == built.CN_ERR_UNSUPPORTED and 'scenario_rule' in msg
    rc, msg = create(scenario_rule=2, num_humans=4)  # mixed draws up to 5 humans per episode
    assert rc == built.CN_ERR_UNSUPPORTED and 'mixed' in msg
    rc, msg = create(robot_policy=7)
    assert rc == built.CN_ERR_INVALID
    assert lib.cn_create(None, None) == built.CN_ERR_INVALID
    assert lib.cn_destroy(None) == built.CN_OK  # destroying NULL is a no-op
    assert lib.cn_sync(None) == built.CN_ERR_INVALID and 'NULL' in lib.cn_last_error().decode()


def test_unicycle_needs_external_robot_policy(built):
    lib = built.load()
    from crowdnav_amd.engine import default_config
    cfg = built.CnConfig(**default_config(robot_kinematics=built.UNICYCLE, robot_policy=built.ROBOT_ORCA))
    h = C.c_void_p()
    assert lib.cn_create(C.byref(cfg), C.byref(h)) == built.CN_ERR_INVALID
    assert 'holonomic' in lib.cn_last_error().decode()
