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

NDTE_EVP = 40
BND = {"open": 0, "cyclic": 1, "closed": 2, "tripole": 3, "tripoleT": 4}
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
    if len(sys.argv) > 5:   # the reference's own gx3 grid + land mask, written from the committed fixture
        d = tempfile.mkdtemp()
        z = np.load(os.path.join(ROOT, "tests", "golden", "gx3_grid_kmt.npz"))
        with open(os.path.join(d, "global_gx3.grid"), "wb") as f:
            for k in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE"):
                f.write(z[k].astype(">f8").tobytes())
        with open(os.path.join(d, "global_gx3.kmt"), "wb") as f:
            f.write(z["kmt"].astype(">i4").tobytes())
        gridkw = dict(grid="displaced_pole", grid_file=os.path.join(d, "global_gx3.grid"),
                      kmt_file=os.path.join(d, "global_gx3.kmt"))
    from __graft_entry__ import REF_CONFIGS
    from oracle import refapi
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg)
    ref.init_domain(tempfile.mkdtemp(), dt=3600.0, ndte=NDTE_EVP, ew=ew, ns=ns, **gridkw)
    nb, ny, nx = ref.max_blocks, ref.ny_block, ref.nx_block
    rng = np.random.default_rng(7)
    nchk = 0

    if mode == "lists":
        from cice4_amd import lib
        dom = lib.Context().domain_create(nxg, nyg, bsx, bsy, ew=BND[ew], ns=BND[ns])
        assert (dom["nx"], dom["ny"]) == (nx, ny) and dom["nblocks"] <= nb    # arrays carry max_blocks blocks
        c = lib.Context()
        # the block -> task map as the reference's create_distribution made it (one task): blocks it kept, in its
        # local order; every other block was eliminated (all land, or outside the latitude bands, ice_domain.F90:403-420)
        nbx, nby = (nxg - 1) // bsx + 1, (nyg - 1) // bsy + 1
        owner = -np.ones(nbx * nby, np.int32); lid = -np.ones(nbx * nby, np.int32)
        for l in range(ref.nblocks):
            g = ref.block_info(l + 1)["block_id"] - 1
            owner[g] = 0; lid[g] = l
        dom = c.domain_create_map(nxg, nyg, bsx, bsy, owner, ew=BND[ew], ns=BND[ns], local_id=lid)
        assert dom["nblocks"] == ref.nblocks
        if cfg == "gx3e":
            assert ref.nblocks < nbx * nby and len(dom["hfill"]) > 0        # land-block elimination really happens
        else:
            plain = lib.Context().domain_create(nxg, nyg, bsx, bsy, ew=BND[ew], ns=BND[ns])   # cartesian map
            assert np.array_equal(plain["hsrc"], dom["hsrc"]) and np.array_equal(plain["hdst"], dom["hdst"])
        locs = (1, 2, 3, 4) if ns.startswith("tripole") else (1,)
        kinds = (1, 2, 3) if ns.startswith("tripole") else (1,)
        for dtype in (np.float64, np.float32, np.int32):
            for loc in locs:
                for kind in kinds:
                    a = rand(rng, (nb, ny, nx), dtype)
                    want = a.copy(); ref.halo_nd(want, loc, kind)
                    got = a.copy()
                    c.apply_halo_lists(got[:dom["nblocks"]], loc, kind, fill=0)
                    assert np.array_equal(got, want), (dtype, ew, ns, loc, kind, np.argwhere(got != want)[:8])
                    assert not np.array_equal(want, a)
                    nchk += 1
                    # the C-ABI entry the boundary module calls with a HOST array (cice_halo_update_blocked_*): on a
                    # one-rank domain the lists are applied on the host inside the caller's array (no device), in the
                    # reference's layout (block outermost, levels inside) -- 1 and 3 levels
                    for shape in ((nb, ny, nx), (nb, 3, ny, nx)):
                        a = rand(rng, shape, dtype)
                        want = a.copy(); ref.halo_nd(want, loc, kind)
                        got = a.copy()
                        c.halo_update_blocked(got[:dom["nblocks"]].reshape(dom["nblocks"], -1, ny, nx), loc, kind, fill=0)
                        assert np.array_equal(got, want), ("blocked", dtype, shape, ew, ns, loc, kind)
                        nchk += 1
        print("BOUNDARY-OK", nchk)
        return

    dro = refapi.Ref(cfg, kind="dropin")
    dro.init_domain(tempfile.mkdtemp(), dt=3600.0, ndte=NDTE_EVP, ew=ew, ns=ns, **gridkw)
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
    # 5. whole evp(dt) (source/ice_dyn_evp.F90:119-432) on the module arrays of both builds: the drop-in dynamics on
    #    this block distribution and boundary type -- tripole fold after every subcycle (u, v: NE corner, vector),
    #    ghost cells next to eliminated land blocks -- against the pure reference, every output field bit for bit
    from cice4_amd import lib, synth
    nbx, nby = (nxg - 1) // bsx + 1, (nyg - 1) // bsy + 1
    owner = -np.ones(nbx * nby, np.int32); lid = -np.ones(nbx * nby, np.int32)
    for l in range(ref.nblocks):
        g = ref.block_info(l + 1)["block_id"] - 1
        owner[g] = 0; lid[g] = l
    dom = lib.Context().domain_create_map(nxg, nyg, bsx, bsy, owner, ew=BND[ew], ns=BND[ns], local_id=lid)
    nbl = dom["nblocks"]
    grid = {k: ref.get(k)[:nbl] for k in ("tmask", "umask")}
    st = synth.evp_state(grid, dom, cover="patchy", seed=11)

    def full(a):
        out = np.zeros((nb * (a.shape[0] // nbl),) + a.shape[1:], a.dtype)
        out[:a.shape[0]] = a
        return out

    sig = tuple(synth.SIG_NAMES)
    outs = ("uvel", "vvel", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT", "strocnyT",
            "strocnx", "strocny", "strintx", "strinty", "strairx", "strairy", "fm", "strtltx", "strtlty", "iceumask") + sig
    res = []
    for r in (ref, dro):
        r.set_strength_parameters(1, 1, 1, 4.0)
        if r is dro:
            r.evp_gpu_setup()
        for k in ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty", "uvel",
                  "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty") + sig:
            r.set(k, full(st[k]))
        r.set("iceumask", full(st["iceumask"].astype(float)))
        r.set("aicen", full(st["aicen"].reshape(-1, ny, nx))); r.set("vicen", full(st["vicen"].reshape(-1, ny, nx)))
        r.evp(3600.0)
        r.evp(3600.0)          # second step: iceumask, velocities and stresses carried
        res.append({k: r.get(k)[:nbl] for k in outs})
    assert np.abs(res[0]["uvel"]).max() > 1e-3
    for k in outs:
        assert np.isfinite(res[0][k]).all(), ("evp: reference not finite", k)
        assert np.array_equal(res[0][k], res[1][k]), ("evp", k, ew, ns, np.abs(res[0][k] - res[1][k]).max())
        nchk += 1
    print("BOUNDARY-OK", nchk)


if __name__ == "__main__":
    main()
