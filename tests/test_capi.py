"""The C-ABI library loads on a CPU-only host and exports every symbol include/cice4_amd.h
declares; device entry points fail LOUDLY (error code + message) when there is no GPU --
there is no CPU fallback in the product."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cice4_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cice4_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cice_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("flavour", ["standalone", "auscom"])
def test_header_symbols_exported(flavour):
    """both builds of the library: libcice4_amd.so and libcice4_amd_auscom.so (the one for a -DAusCOM -Dcoupled
    reference) export the whole header and say which one they are"""
    l = lib.load(flavour)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/cice4_amd.h but not exported"
    assert l.cice_build_flavour().decode() == flavour


def test_namelist_setters_belong_to_the_coupled_build():
    """no device needed to be told so: the stand-alone build has the turning angle, drag and chio compiled in"""
    c = lib.Context()
    with pytest.raises(lib.CiceError, match="libcice4_amd_auscom.so"):
        c.set_auscom(sinw=0.2)
    with pytest.raises(lib.CiceError, match="libcice4_amd_auscom.so"):
        c.set_chio(0.004)


def test_product_never_imports_the_oracle():
    """cice4_amd/ and bench.py's measured path must not reach into oracle/ (checker only)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cice4_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".F90", ".f90")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace(
                    "oracle/_ref", "").replace("the oracle", ""), f


def test_device_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = lib.Context()
    with pytest.raises(lib.CiceError):
        c.sync()
    c.domain_create(16, 12, 8, 6)
    with pytest.raises(lib.CiceError) as e:
        c.thermo_init()
        a = np.zeros((8, 10))
        c.frzmlt_bottom_lateral(2, 9, 2, 7, 3600.0, a, a, np.zeros((20, 8, 10)), np.zeros((5, 8, 10)), a, a, a, a)
    assert "hip" in str(e.value).lower() or "device" in str(e.value).lower()


def test_argument_errors_return_codes():
    c = lib.Context()
    with pytest.raises(lib.CiceError):
        c.domain_create(0, 10, 5, 5)
    with pytest.raises(lib.CiceError):
        c.domain_create(10, 10, 5, 5, ew=3)
    with pytest.raises(lib.CiceError):
        c.domain_create(10, 10, 5, 5, rank=4, npx=2, npy=2)
    with pytest.raises(lib.CiceError):
        c.thermo_init(heat_capacity=False)      # zero-layer thermodynamics: not implemented, says so


def test_host_model_sizes_are_checked():
    """A model built with another ncat / nilyr / nslyr / max_ntrcr must not get past its init calls: the library's
    strides of those dimensions are compile-time sizes (the drop-in modules call this from init_evp,
    init_thermo_vertical and init_transport)."""
    c = lib.Context()
    c.check_sizes(5, 4, 1, 5)
    for bad in ((6, 4, 1, 5), (5, 7, 1, 5), (5, 4, 3, 5), (5, 4, 1, 6)):
        with pytest.raises(lib.CiceError) as e:
            c.check_sizes(*bad)
        assert "ncat=5" in str(e.value) and f"ncat={bad[0]}" in str(e.value)
    assert c.comm_count() == 0          # no communicator yet
    with pytest.raises(lib.CiceError):
        c.halo_msgs(0)                  # no domain: an error, not an empty list
    c.domain_create(16, 12, 8, 6)
    with pytest.raises(lib.CiceError):
        c.halo_msgs(7)


def test_sweep_kernel_strip_layout_is_right_for_every_width():
    """cice_debug_skew_layout (no device): for every block width and every K (and both workgroup shapes of K = 4) the strip
    layout of the K-subcycle sweep kernel has a shift of the seam strip with which a lane-level restatement of the
    kernel's dependencies -- western velocity for the stress (two lanes for ilo), eastern stress for the momentum, G's
    velocity from the lane beside it, owner lanes of the shared columns -- leaves every owned column right; the shift is
    small, and without the shift exactly the widths with the seam in a rim (or on shared columns) fail."""
    import ctypes as C
    from cice4_amd import lib
    L = lib.load()
    for K in (2, 3, 4, 5, 6, 8):
        for S in ((1, 3) if K == 4 else (1,)):
            ownw = 62 * S + 2 - 2 * K
            shifted = 0
            for ncol in list(range(K, 420)) + [1440, 3600]:
                st = C.c_int(0)
                sh = L.cice_debug_skew_layout(K, S, ncol, 1, C.byref(st))
                assert 0 <= sh <= K + 1, (K, S, ncol, sh)
                assert (st.value - 1) * ownw < ncol + 1 + ownw, (K, S, ncol, st.value)      # no strip too many
                shifted += sh > 0
                # open east-west edge: no seam, no shift
                assert L.cice_debug_skew_layout(K, S, ncol, 0, C.byref(st)) == 0, (K, S, ncol)
            assert 0 < shifted < 60, (K, S, shifted)
    assert L.cice_debug_skew_layout(9, 1, 100, 1, None) == -2


@pytest.mark.parametrize("cover", ["full", "caps", "blobs"])
def test_sweep_segments_converge_on_a_model_of_the_kernel(cover):
    """cice_debug_balance_strip (no device): the per-strip step of the measured balancing, iterated against a model of the
    sweep kernel -- a workgroup takes (rows with ice in its runs + what each run costs to start) / (speed of its place),
    with 1.5 % of noise -- reaches tiles of equal time from equal segments within the sweeps one loop measures, keeps a
    partition of the strip's rows at every step, and stays there."""
    import ctypes as C
    from cice4_amd import lib
    L = lib.load()
    rng = np.random.default_rng(7)
    rows, n, K = 2400, 11, 4
    act = np.ones(rows, np.uint8)
    if cover == "caps":
        act[:] = 0; act[:250] = 1; act[2010:] = 1
    elif cover == "blobs":
        act[:] = 0
        for lo, hi in ((100, 380), (700, 760), (1200, 1800), (2250, 2390)):
            act[lo:hi] = 1
    speed = np.array([1.15, 1.0, 0.85, 1.15, 1.0, 0.85, 1.26, 1.26, 1.0, 0.85, 1.15])     # what the place really does
    w = np.array([1.15, 1.0, 0.85, 1.15, 1.0, 0.85, 1.26, 1.26, 1.0, 0.85, 1.15]) * rng.uniform(0.93, 1.07, n)   # what the table believes

    def kernel_model(ends):
        d = np.zeros(n)
        lo = 0
        for i, e in enumerate(ends):
            a = act[lo:e]
            if a.any():
                idx = np.flatnonzero(a)
                gaps = np.diff(idx) - 1
                runs = 1 + int((gaps > 3 * K).sum())
                bridged = int(gaps[gaps <= 3 * K].sum())
                d[i] = (len(idx) + bridged + runs * (4 * K - 1)) / speed[i] * rng.normal(1.0, 0.015)
            lo = e
        return d

    ends = np.array([(i + 1) * rows // n for i in range(n)], np.int32)
    ends[-1] = rows
    worst = []
    for it in range(36):
        d = kernel_model(ends)
        busy = d[d > 0]
        worst.append(d.max() / busy.mean())
        ne = np.zeros(n, np.int32)
        tot = C.c_double(0)
        rc = L.cice_debug_balance_strip(rows, n, ends.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                        w.ctypes.data_as(C.c_void_p), act.ctypes.data_as(C.c_void_p), ne.ctypes.data_as(C.c_void_p),
                                        C.byref(tot))
        assert rc == 0 and tot.value > 0
        assert ne[-1] == rows and np.all(np.diff(np.concatenate([[0], ne])) >= 2), ne.tolist()     # a partition, no tile below 2 rows
        ends = ne
    d = kernel_model(ends)
    assert (d > 0).sum() == n, "every tile should have got rows with ice"
    assert d.max() / d.mean() < 1.06, (cover, worst[:3], worst[-3:], d.round(1).tolist())
    assert worst[0] > 1.12, "the model should start out unbalanced"
    # nothing to go by: all durations zero (a strip of open water) -> the ends stay
    z = np.zeros(n)
    rc = L.cice_debug_balance_strip(rows, n, ends.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                    act.ctypes.data_as(C.c_void_p), ne.ctypes.data_as(C.c_void_p), C.byref(tot))
    assert rc == 0 and tot.value == 0 and np.array_equal(ne, ends)
    bad = ends.copy(); bad[3] = bad[2] - 1
    assert L.cice_debug_balance_strip(rows, n, bad.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                      None, ne.ctypes.data_as(C.c_void_p), None) == -2
