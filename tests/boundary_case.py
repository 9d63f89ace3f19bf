"""One boundary-module comparison in its own process (the compiled reference allows ONE
init_domain per process and library).  Started by tests/test_boundary.py:

    python tests/boundary_case.py lists <cfg> <ew> <ns> [gx3 grid dir]
                                                          CPU: the product's halo lists applied in
                                                          numpy == the reference's ice_HaloUpdate
    python tests/boundary_case.py gpu   <cfg> <ew> <ns>   MI355X: the reference's callers linked with
                                                          cice4_amd/fortran/rccl/ice_boundary.F90
                                                          == the pure reference, bit for bit
Prints 'BOUNDARY-OK <n checks>' on success.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BND = {"open": 0, "cyclic": 1, "closed": 2}
GRID_FIELDS = ("HTN", "HTE", "dxt", "dyt", "dxu", "dyu", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
               "tarea", "uarea", "tarear", "uarear", "tinyarea", "ULAT", "ULON", "TLAT", "TLON",
               "ANGLE", "hm", "uvm", "tmask", "umask")


def rand(rng, shape, dtype):
    if dtype == np.int32:
        return rng.integers(-1000, 1000, shape).astype(np.int32)
    return rng.uniform(-2.0, 2.0, shape).astype(dtype)


def main():
    mode, cfg, ew, ns = sys.argv[1:5]
    gridkw = {}
    if len(sys.argv) > 5:   # the reference's own gx3 grid + land mask (only where /root/reference exists)
        d = sys.argv[5]
        gridkw = dict(grid="displaced_pole", grid_file=os.path.join(d, "global_gx3.grid"),
                      kmt_file=os.path.join(d, "global_gx3.kmt"))
    from __graft_entry__ import REF_CONFIGS
    from oracle import refapi
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg)
    ref.init_domain(tempfile.mkdtemp(), dt=3600.0, ndte=4, ew=ew, ns=ns, **gridkw)
    nb, ny, nx = ref.max_blocks, ref.ny_block, ref.nx_block
    rng = np.random.default_rng(7)
    nchk = 0

    if mode == "lists":
        from cice4_amd import lib
        dom = lib.Context().domain_create(nxg, nyg, bsx, bsy, ew=BND[ew], ns=BND[ns])
        assert (dom["nx"], dom["ny"]) == (nx, ny) and dom["nblocks"] <= nb    # arrays carry max_blocks blocks
        for dtype in (np.float64, np.int32):
            a = rand(rng, (nb, ny, nx), dtype)
            want = a.copy(); ref.halo_nd(want)
            got = a.copy().reshape(-1); got[dom["hdst"]] = got[dom["hsrc"]]
            assert np.array_equal(got.reshape(a.shape), want), (dtype, ew, ns)
            assert not np.array_equal(want, a)
            nchk += 1
        print("BOUNDARY-OK", nchk)
        return

    dro = refapi.Ref(cfg, kind="dropin")
    dro.init_domain(tempfile.mkdtemp(), dt=3600.0, ndte=4, ew=ew, ns=ns, **gridkw)
    # 1. the grid the reference's init_grid2 builds THROUGH the boundary module (ice_HaloUpdate with
    #    fillValue, ice_HaloExtrapolate): identical arrays from both builds
    for name in GRID_FIELDS:
        assert np.array_equal(ref.get(name), dro.get(name)), ("grid", name, ew, ns)
        nchk += 1
    # 2. every specific of the generic ice_HaloUpdate
    for dtype in (np.float64, np.float32, np.int32):
        for shape in ((nb, ny, nx), (nb, 3, ny, nx), (nb, 2, 3, ny, nx)):
            a = rand(rng, shape, dtype)
            want = a.copy(); ref.halo_nd(want, 2, 2)
            got = a.copy(); dro.halo_nd(got, 2, 2)
            assert np.array_equal(got, want), ("update", dtype, shape, ew, ns)
            assert not np.array_equal(want, a)
            nchk += 1
    # 3. ice_HaloExtrapolate
    a = rand(rng, (nb, ny, nx), np.float64)
    want = a.copy(); ref.halo_extrapolate(want)
    got = a.copy(); dro.halo_extrapolate(got)
    assert np.array_equal(got, want), ("extrapolate", ew, ns)
    if ew != "cyclic" or ns != "cyclic":
        assert not np.array_equal(want, a)
    nchk += 1
    # 4. bound_state (ice_state.F90:162): the model's state-variable ghost update, 3-d and 4-d
    NC, NI, NS, NT = refapi.NCAT, refapi.NILYR, refapi.NSLYR, refapi.MAX_NTRCR
    st = [rand(rng, (nb, NC, ny, nx), np.float64), rand(rng, (nb, NC, NT, ny, nx), np.float64),
          rand(rng, (nb, NC, ny, nx), np.float64), rand(rng, (nb, NC, ny, nx), np.float64),
          rand(rng, (nb, NC * NI, ny, nx), np.float64), rand(rng, (nb, NC * NS, ny, nx), np.float64)]
    want = [x.copy() for x in st]; ref.bound_state(*want)
    got = [x.copy() for x in st]; dro.bound_state(*got)
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), ("bound_state", k, ew, ns)
        nchk += 1
    print("BOUNDARY-OK", nchk)


if __name__ == "__main__":
    main()
