import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Several tests run the ranks of a multi-rank job as contexts of this process whose one-launch loops wait for each other: each
# needs a hardware queue of its own (on a node each has a GPU of its own).  The runtime multiplexes its streams on four
# queues unless told otherwise -- read when it starts, hence here, before anything touches the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU checker (plain-C restatement, oracle/)."""
    from oracle import oracle
    oracle.build()
    return oracle.Oracle()


def _ref(cfg, kind="ref"):
    from oracle import refapi
    if not refapi.available(cfg, kind):
        pytest.skip(f"compiled reference oracle/_ref/libcice_{kind}_{cfg}.so not built")
    return refapi.Ref(cfg, kind)


@pytest.fixture(scope="session")
def ref_gx3():
    return _ref("gx3")


@pytest.fixture(scope="session")
def ref_gx3b4():
    return _ref("gx3b4")


@pytest.fixture(scope="session")
def refaus_gx3b4():
    """the reference compiled -DAusCOM -Dcoupled with the access-om driver's constants (AUS=1 oracle/build_ref.sh)"""
    return _ref("gx3b4", "refaus")


@pytest.fixture(scope="session")
def orc_aus():
    from oracle import oracle
    oracle.build()
    return oracle.Oracle(aus=True)


@pytest.fixture(scope="session")
def ctx():
    """Device context of the product library.  No fallback: fails if there is no GPU."""
    from cice4_amd import lib
    c = lib.Context()
    c.sync()  # raises CiceError without a device
    return c


def host_libm_is_restated():
    """The device evaluates exp() with glibc's own algorithm in the form glibc uses on x86-64 CPUs with FMA
    (cice4_amd/csrc/libm_exact.h; tests/test_libm_exact.py checks it against the host libm).  On such a
    host the checker, the compiled reference and the device therefore agree BIT FOR BIT also where exp()
    is involved; on a host without FMA glibc runs its other build and only the 1e-10 bound holds."""
    try:
        with open("/proc/cpuinfo") as f:
            return any(line.startswith("flags") and " fma " in line + " " for line in f)
    except OSError:
        return False


# tolerance for results that pass through exp(): 0 = bit for bit
TOL_EXP = 0.0 if host_libm_is_restated() else 1e-10
# results that pass through pow() (frzmlt_bottom_lateral, `deltaT**m2`): glibc's pow is restated on the device too
TOL_POW = TOL_EXP


def single_block_domain(c, nxg, nyg, ew=1, ns=0):
    return c.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=ns)


def relerr(a, b):
    """max |a-b| / max|b| over the field (field-level relative error of BASELINE.json)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.abs(b).max()
    if den == 0.0:
        return np.abs(a).max()
    return np.abs(a - b).max() / den
