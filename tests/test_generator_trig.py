"""The scenario generators' float32 stages (crowdnav_amd/csrc/scenario_wave.h: blocked-cell table, prefilter) take cos / sin of
an attempt's angle from the hardware's V_COS_F32 / V_SIN_F32 and are exact only if those are within cn::kTrigAbsError of the
true values (the margins are ten times the resulting position error).  scripts/probes/trig_error.hip measures the error over
ALL 2^27 angle fractions the generators can feed them, on the device, against float64 cos / sin."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_probe_assume_the_same_bound():
    hdr = open(os.path.join(ROOT, 'crowdnav_amd', 'csrc', 'scenario_wave.h')).read()
    probe = open(os.path.join(ROOT, 'scripts', 'probes', 'trig_error.hip')).read()
    h = float(re.search(r'kTrigAbsError = ([0-9.e+-]+)f', hdr).group(1))
    p = float(re.search(r'const float bound = ([0-9.e+-]+)f', probe).group(1))
    assert h == p == 1.0e-5


@pytest.mark.gpu
def test_hardware_sin_cos_are_within_the_bound_the_generators_assume(tmp_path):
    hipcc = next((c for c in ('/opt/rocm/bin/hipcc', shutil.which('hipcc')) if c and os.path.exists(c)), None)
    if hipcc is None:
        pytest.skip('no hipcc on this machine')
    exe = str(tmp_path / 'trig_error')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', os.path.join(ROOT, 'scripts', 'probes', 'trig_error.hip'), '-o', exe],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    worst = float(re.search(r'fractions: ([0-9.e+-]+)', r.stdout).group(1))
    assert 0.0 < worst <= 1.0e-5
