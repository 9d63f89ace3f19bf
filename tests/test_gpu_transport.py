"""Horizontal transport by incremental remapping: cice_transport_remap on the MI355X against
`call transport_remap(dt)` of the compiled reference (source/ice_transport_driver.F90:179,
source/ice_transport_remap.F90:328), one process per configuration (tests/transport_case.py): 2 x 2 and padded
3 x 3 blocks, cyclic / open edges, tripole north boundary, and the real gx3 grid and land mask cut into 120 blocks
with the all-land ones eliminated.  Every state array incl. ghost cells, three flow / ice-cover regimes: bit for bit.
The same configurations for advection = 'upwind' (cice_transport_upwind against `call transport_upwind(dt)`, :672)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("small", "cyclic", "open"), ("pad", "cyclic", "open"), ("pad", "open", "open"), ("small", "cyclic", "tripole"),
         ("small", "cyclic", "tripoleT"),
         ("gx3", "cyclic", "open", "gx3"), ("gx3e", "cyclic", "open", "gx3")]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(c))
def test_transport_remap_equals_reference(case):
    from oracle import refapi
    if not refapi.available(case[0]):
        pytest.skip(f"oracle/_ref/libcice_ref_{case[0]}.so not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "transport_case.py"), *case],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "TRANSPORT-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-4000:]


# (no open east-west edge here: the reference's transport_upwind keeps its edge velocities in automatic arrays that
#  ice_HaloUpdate leaves untouched on an open edge, :699-738 -- what it computes there is not defined)
@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c[1] != "open"], ids=lambda c: "-".join(c))
def test_transport_upwind_equals_reference(case):
    from oracle import refapi
    if not refapi.available(case[0]):
        pytest.skip(f"oracle/_ref/libcice_ref_{case[0]}.so not built")
    args = list(case) + ["-"] * (4 - len(case)) + ["upwind"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "transport_case.py"), *args],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "TRANSPORT-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("bsx,bsy", [(64, 48), (32, 24)], ids=["1block", "2x2blocks"])
def test_evp_then_transport_without_a_pcie_round_trip(bsx, bsy):
    """cice_transport_chain (source/ice_step_mod.F90:575-584: `call evp(dt)` is followed at once by `call transport_remap(dt)`):
    with the chain, evp prefetches aice0, trcrn, vsnon, eicen, esnon while it subcycles and the transport takes uvel, vvel,
    aicen, vicen from the dynamics' device buffers -- it uploads nothing.  Two steps chained = two steps unchained, bit for
    bit; that the device copies really were used is shown by breaking the contract on purpose (a host array changed after
    evp returned has no effect); a transport call that does not follow an evp call uploads as usual."""
    import numpy as np
    from cice4_amd import lib, synth
    DT, NDTE = 3600.0, 120
    nxg, nyg = 64, 48
    ctx = lib.Context()
    ctx.sync()
    dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.05), dom)
    s0 = synth.evp_state(grid, dom, cover="patchy")
    nb, ncat, ny, nx = s0["aicen"].shape
    rng = np.random.default_rng(5)

    def fresh():
        s = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s0.items()}
        a, v = s["aicen"], s["vicen"]
        ts = dict(aicen=a, vicen=v, uvel=s["uvel"], vvel=s["vvel"])
        ts["vsnon"] = np.ascontiguousarray(0.2 * a)
        tr = np.zeros((nb, ncat, 5, ny, nx))
        tr[:, :, 0] = -5.0 - 3.0 * np.arange(ncat)[None, :, None, None]
        tr[:, :, 1] = 10.0
        ts["trcrn"] = tr
        ts["eicen"] = np.ascontiguousarray(np.repeat(v, 4, axis=1) * (-7.5e7) * (1.0 + 0.1 * np.arange(ncat * 4) % 4)[None, :, None, None])
        ts["esnon"] = np.ascontiguousarray(ts["vsnon"] * (-1.1e8))
        ts["aice0"] = np.ascontiguousarray(1.0 - a.sum(axis=1))
        return s, ts

    def setup():
        ctx.evp_init(grid, ndte=NDTE)
        ctx.transport_init({k: grid[k] for k in ("HTN", "HTE", "dxt", "dyt", "dxu", "dyu", "tarear", "hm")}, ntrcr=2,
                           trcr_depend=(0, 1))

    keys = ("aice0", "aicen", "vicen", "vsnon", "trcrn", "eicen", "esnon")
    # unchained: two steps
    setup()
    sA, tsA = fresh()
    for _ in range(2):
        ctx.evp(DT, sA)
        assert ctx.transport_remap(DT, tsA) == (0, 0, 0)
    assert np.abs(sA["uvel"]).max() > 0.01 and not np.array_equal(tsA["aicen"], s0["aicen"])
    # chained: the same two steps
    setup()
    sB, tsB = fresh()
    for d in (sB, tsB):
        for v in d.values():
            if isinstance(v, np.ndarray):
                ctx.host_register(v)
    ctx.transport_chain(tsB)
    for _ in range(2):
        ctx.evp(DT, sB)
        assert ctx.transport_remap(DT, tsB) == (0, 0, 0)
    for k in keys:
        assert np.array_equal(tsB[k], tsA[k]), ("chained vs unchained", k)
    # the device copies are what the transport reads: a host array changed AFTER evp returned (against the contract) is not seen
    ctx.evp(DT, sB)
    kept = tsB["vsnon"].copy()
    tsB["vsnon"][...] = 0.0
    assert ctx.transport_remap(DT, tsB) == (0, 0, 0)
    ctx.evp(DT, sA)
    assert ctx.transport_remap(DT, tsA) == (0, 0, 0)
    for k in keys:
        assert np.array_equal(tsB[k], tsA[k]), ("third step: prefetched state used", k)
    assert np.abs(kept).max() > 0 and np.abs(tsB["vsnon"]).max() > 0
    # a transport call that does not follow an evp call uploads as usual: now the changed host array IS seen
    tsB["vsnon"][...] = 0.0
    tsA["vsnon"][...] = 0.0
    assert ctx.transport_remap(DT, tsB) == (0, 0, 0)
    ctx.transport_chain(None)
    assert ctx.transport_remap(DT, tsA) == (0, 0, 0)
    for k in keys:
        assert np.array_equal(tsB[k], tsA[k]), ("no evp before: uploaded as usual", k)
    assert np.abs(tsB["vsnon"]).max() == 0.0
    ctx.host_unregister_all()
