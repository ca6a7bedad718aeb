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
    # cn_config: 2 i32, 6 f64, 2 i32, 3 f64, 2 i32, 8 f64, 2 i32 -> 24 x 8 bytes, no padding surprises
    assert C.sizeof(built.CnConfig) == 8 + 48 + 8 + 24 + 8 + 64 + 8
    assert built.CnConfig.time_step.offset == 8 and built.CnConfig.neighbor_dist.offset == 80
    assert built.CnConfig.device.offset == C.sizeof(built.CnConfig) - 4
    assert C.sizeof(built.CnRolloutIo) == 8 + 8 + 8 + 8 + 8 + 13 * 8


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
