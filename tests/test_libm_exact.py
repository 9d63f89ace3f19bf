"""cice4_amd/csrc/libm_exact.h restates glibc's exp (the third-party routine behind the reference's
Fortran `exp` calls, source/ice_mechred.F90:2001, source/ice_therm_vertical.F90:2393) so that the device
reproduces the host's bits.  Here: the restatement, compiled for the host, against the host libm."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def test_exp_restatement_equals_host_libm():
    if not _host_has_fma():
        pytest.skip("host CPU without FMA: glibc selects its non-FMA exp build here")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "probe")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma",
                               os.path.join(ROOT, "tests", "libm_probe.cpp"), "-o", exe])
        out = subprocess.run([exe, "10000000"], capture_output=True, text=True, check=True).stdout
    assert "mismatches=0" in out, out
