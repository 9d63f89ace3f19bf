"""One horizontal-transport comparison in its own process (the compiled reference allows ONE init_domain per
process and library).  Started by tests/test_gpu_transport.py:

    python tests/transport_case.py <cfg> <ew> <ns> [gx3|-] [upwind]

`call transport_remap(dt)` of the compiled reference (source/ice_transport_driver.F90:179; oracle/_ref) on its own
module arrays, block distribution and boundary types, against cice_transport_remap on the MI355X with the same
inputs: every state array, ghost cells included, bit for bit.  With `upwind`: `call transport_upwind(dt)` (:672) against
cice_transport_upwind.  Prints 'TRANSPORT-OK <n checks>'.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BND = {"open": 0, "cyclic": 1, "closed": 2, "tripole": 3, "tripoleT": 4}
NC, NI, NS, NT = 5, 4, 1, 5
DT = 3600.0


def main():
    cfg, ew, ns = sys.argv[1:4]
    gridkw = {}
    upwind = len(sys.argv) > 5 and sys.argv[5] == "upwind"
    if len(sys.argv) > 4 and sys.argv[4] == "gx3":   # the reference's own gx3 grid + land mask, written from the committed fixture
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
    from cice4_amd import lib
    from oracle import refapi
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg)
    ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=4, ew=ew, ns=ns, **gridkw)
    ref.init_transport()
    nbm, ny, nx = ref.max_blocks, ref.ny_block, ref.nx_block
    nbl = ref.nblocks
    # device topology = the reference's block -> task map
    nbx, nby = (nxg - 1) // bsx + 1, (nyg - 1) // bsy + 1
    owner = -np.ones(nbx * nby, np.int32); lid = -np.ones(nbx * nby, np.int32)
    for l in range(nbl):
        g = ref.block_info(l + 1)["block_id"] - 1
        owner[g] = 0; lid[g] = l
    ctx = lib.Context(); ctx.sync()
    dom = ctx.domain_create_map(nxg, nyg, bsx, bsy, owner, ew=BND[ew], ns=BND[ns], local_id=lid)
    assert dom["nblocks"] == nbl
    grid = {k: np.ascontiguousarray(ref.get(k)[:nbl]) for k in ("HTN", "HTE", "dxt", "dyt", "dxu", "dyu", "tarear", "hm")}
    if upwind:
        ctx.transport_upwind_init(grid["HTE"], grid["HTN"], np.ascontiguousarray(ref.get("tarea")[:nbl]), ntrcr=2,
                                  trcr_depend=(0, 1), nt_Tsfc=1)
    else:
        ctx.transport_init(grid, ntrcr=2, trcr_depend=(0, 1))
    rng = np.random.default_rng(20261004)
    nchk = 0
    hm = grid["hm"]
    # largest displacement that stays inside the neighbouring cells (departure_points :1611-1620), per U point
    HTN, HTE = grid["HTN"], grid["HTE"]
    dloc = np.zeros_like(HTN)
    dloc[:, :-1, :-1] = np.minimum(np.minimum(HTN[:, :-1, :-1], HTN[:, :-1, 1:]), np.minimum(HTE[:, :-1, :-1], HTE[:, 1:, :-1]))
    dloc = np.maximum(dloc, 0.0)
    # global coordinates of every local cell (ghost cells through the block offsets) for smooth fields
    gi = np.zeros((nbl, ny, nx)); gj = np.zeros((nbl, ny, nx))
    for b in range(nbl):
        ii = (np.arange(nx) - (dom["ilo"][b] - 1) + dom["i0"][b]) % nxg
        jj = np.arange(ny) - (dom["jlo"][b] - 1) + dom["j0"][b]
        gi[b], gj[b] = np.meshgrid(ii, jj, indexing="xy")
    for trial, (speed, cover) in enumerate(((0.35, "patchy"), (0.9, "full"), (0.15, "edge"))):
        # ---- a state on the physical cells; ghost cells come from the reference's own bound_state
        sm = lambda k: 0.5 + 0.5 * np.sin(2 * np.pi * (k + 1) * gi / nxg + k) * np.cos(np.pi * (k + 2) * gj / nyg + 0.3 * k)
        conc = 0.2 + 0.75 * sm(0)
        if cover == "patchy":
            conc = np.where(sm(1) < 0.35, 0.0, conc)
        elif cover == "edge":
            conc = np.where(gj < nyg / 2, 0.0, conc) * (rng.uniform(0, 1, conc.shape) < 0.8)
        conc = conc * (hm > 0)
        w = np.array([0.1, 0.2, 0.3, 0.25, 0.15])
        aicen = np.zeros((nbl, NC, ny, nx)); vicen = np.zeros_like(aicen); vsnon = np.zeros_like(aicen)
        trcrn = np.zeros((nbl, NC, NT, ny, nx)); eicen = np.zeros((nbl, NC * NI, ny, nx)); esnon = np.zeros((nbl, NC * NS, ny, nx))
        for n in range(NC):
            aicen[:, n] = conc * w[n] * (0.7 + 0.6 * sm(n + 2)) * (rng.uniform(0, 1, conc.shape) < 0.9)
            h = 0.3 + n * 0.8 + 0.5 * sm(n + 7) + 0.05 * rng.uniform(0, 1, conc.shape)
            vicen[:, n] = aicen[:, n] * h
            hs = np.where(sm(n + 11) > 0.4, 0.25 * sm(n + 12), 0.0)          # snow-free patches
            vsnon[:, n] = aicen[:, n] * hs
            trcrn[:, n, 0] = np.where(aicen[:, n] > 0, -1.8 - 15.0 * sm(n + 13), 0.0)           # Tsfc
            trcrn[:, n, 1] = np.where(aicen[:, n] > 0, 1.0e5 * (1 + n) * sm(n + 14), 0.0)        # iage
            for l in range(NI):
                eicen[:, n * NI + l] = -vicen[:, n] / NI * 3.0e8 * (0.8 + 0.2 * sm(n + l + 15))
            esnon[:, n] = -vsnon[:, n] * 1.1e8 * (0.9 + 0.1 * sm(n + 20))
        tot = aicen.sum(axis=1)
        scale = np.where(tot > 0.98, 0.98 / np.maximum(tot, 1e-30), 1.0)[:, None]
        aicen *= scale; vicen *= scale; vsnon *= scale; eicen *= scale; esnon *= scale
        aice0 = 1.0 - aicen.sum(axis=1)
        # velocities at U points: displacements up to `speed` of the smallest cell edge, some cells at rest
        umax = speed * dloc / DT
        uvel = umax * (np.sin(2 * np.pi * gi / nxg * 2 + 0.5) * np.cos(np.pi * gj / nyg) + 0.4 * rng.uniform(-1, 1, gi.shape))
        vvel = umax * (np.cos(2 * np.pi * gi / nxg) * np.sin(np.pi * gj / nyg * 2) + 0.4 * rng.uniform(-1, 1, gi.shape))
        rest = rng.uniform(0, 1, gi.shape) < 0.1
        uvel = np.where(rest, 0.0, uvel) / 1.4; vvel = np.where(rest, 0.0, vvel) / 1.4

        def full(a):       # host arrays of the reference carry max_blocks blocks
            out = np.zeros((nbm,) + a.shape[1:], a.dtype); out[:nbl] = a
            return out
        st = [full(aicen), full(trcrn), full(vicen), full(vsnon), full(eicen), full(esnon)]
        ref.bound_state(*st)                       # consistent ghost cells, the way the model keeps its state
        a0 = full(aice0); ref.halo_nd(a0, 1, 1)
        uu = full(uvel); vv = full(vvel); ref.halo_nd(uu, 2, 2); ref.halo_nd(vv, 2, 2)
        ref.set("aicen", st[0].reshape(-1, ny, nx)); ref.set("trcrn", st[1].reshape(-1, ny, nx))
        ref.set("vicen", st[2].reshape(-1, ny, nx)); ref.set("vsnon", st[3].reshape(-1, ny, nx))
        ref.set("eicen", st[4].reshape(-1, ny, nx)); ref.set("esnon", st[5].reshape(-1, ny, nx))
        ref.set("aice0", a0); ref.set("uvel", uu); ref.set("vvel", vv)
        dev = dict(aicen=st[0][:nbl].copy(), trcrn=st[1][:nbl].copy(), vicen=st[2][:nbl].copy(), vsnon=st[3][:nbl].copy(),
                   eicen=st[4][:nbl].copy(), esnon=st[5][:nbl].copy(), aice0=a0[:nbl].copy(), uvel=uu[:nbl].copy(),
                   vvel=vv[:nbl].copy())
        before = {k: v.copy() for k, v in dev.items()}
        if upwind:
            ref.transport_upwind(DT)
            ctx.transport_upwind(DT, dev)
        else:
            ref.transport_remap(DT)
            assert ctx.transport_remap(DT, dev) == (0, 0, 0)
        want = dict(aicen=ref.get("aicen", nbm * NC).reshape(nbm, NC, ny, nx), trcrn=ref.get("trcrn", nbm * NC * NT).reshape(nbm, NC, NT, ny, nx),
                    vicen=ref.get("vicen", nbm * NC).reshape(nbm, NC, ny, nx), vsnon=ref.get("vsnon", nbm * NC).reshape(nbm, NC, ny, nx),
                    eicen=ref.get("eicen", nbm * NC * NI).reshape(nbm, NC * NI, ny, nx),
                    esnon=ref.get("esnon", nbm * NC * NS).reshape(nbm, NC * NS, ny, nx), aice0=ref.get("aice0"))
        moved = 0
        for k, wv in want.items():
            wv = wv[:nbl]
            if not np.array_equal(dev[k], wv):
                bad = np.argwhere(dev[k] != wv)
                raise AssertionError((cfg, ew, ns, trial, k, len(bad), bad[:6].tolist(),
                                      float(np.abs(dev[k] - wv).max()), float(np.abs(wv).max())))
            moved += int(not np.array_equal(wv, before[k]))
            nchk += 1
        assert moved >= 6, (trial, moved)          # the step really changed the state
    print("TRANSPORT-OK", nchk)


if __name__ == "__main__":
    main()
