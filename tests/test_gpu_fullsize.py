"""Parity at BASELINE.json's full sizes (gx1 320x384, 0.1-degree 3600x2400), through the C-ABI.

gx1: the CPU checker finishes a whole evp(dt) in ~1.5 s, so it is compared directly.
0.1 degree: the checker would need minutes for 240 subcycles, so
  * a direct comparison is made with ndte = 4 (same kernels, same tiles, same halos),
  * and the full ndte = 240 run is checked through size-independent properties:
    decomposition invariance (1 block == 4x3 blocks, bit for bit: every tile edge, block edge
    and on-rank halo path moves) and bounded, finite fields;
  * thermo columns are independent, so a random 1-in-64 sample of the 43 M columns is
    compared with the checker.
"""
import numpy as np
import pytest

from cice4_amd import lib, synth
from conftest import relerr, TOL_EXP

pytestmark = pytest.mark.gpu
DT = 3600.0
PRIMARY = ("uvel", "vvel") + synth.SIG_NAMES


def setup(ctx, nxg, nyg, bsx, bsy, **kw):
    dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, **kw), dom)
    return dom, grid, synth.evp_state(grid, dom, cover="full")


def physical(dom, f):
    """reassemble the global physical field from per-block arrays"""
    g = np.zeros((dom["nyg"], dom["nxg"]))
    for b in range(dom["nblocks"]):
        ni = dom["ihi"][b] - dom["ilo"][b] + 1; nj = dom["jhi"][b] - dom["jlo"][b] + 1
        g[dom["j0"][b]:dom["j0"][b] + nj, dom["i0"][b]:dom["i0"][b] + ni] = \
            f[b, dom["jlo"][b] - 1:dom["jhi"][b], dom["ilo"][b] - 1:dom["ihi"][b]]
    return g


def test_gx1_whole_evp_against_checker(ctx, orc):
    dom, grid, s = setup(ctx, 320, 384, 320, 384, perturb=0.1, land_frac=0.03)
    orc.set_evp_parameters(DT, 120); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=120, krdg_partic=0, krdg_redist=0)
    ctx.evp(DT, sg)
    for k in PRIMARY + ("divu", "shear", "strength", "strocnxT", "strocnyT"):
        assert np.array_equal(sg[k], so[k]), k           # exp-free strength: bit for bit
    # default strength: ice_strength calls exp().  The checker's own result moves by 1.8e-11 (u), 3.4e-11 (v),
    # 6.0e-10 (sigma) when `strength` changes by 1 ulp (perturb_strength_ulp probe, printed below), so a
    # device exp that is merely accurate cannot hold 1e-10 here (round 1: 6.8e-11 / 1.3e-10 / 2.2e-9).
    # The device now evaluates exp() with glibc's own algorithm -> bit for bit (TOL_EXP = 0), unconditionally
    # <= 1e-10 on a host with the other glibc build.
    orc.set_strength_parameters()
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    sp = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid, perturb_strength_ulp=1), sp)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=120)
    ctx.evp(DT, sg)
    for k in PRIMARY + ("strength", "divu", "shear", "strocnxT", "strocnyT"):
        assert relerr(sg[k], so[k]) <= TOL_EXP, (k, relerr(sg[k], so[k]))
    print("gx1 1-ulp-strength sensitivity of the checker:", {k: float(relerr(sp[k], so[k])) for k in ("uvel", "vvel", "stressp_1")},
          "gpu-vs-checker:", {k: float(relerr(sg[k], so[k])) for k in ("uvel", "vvel", "stressp_1")})


def test_tenth_degree_evp(ctx, orc):
    nxg, nyg = 3600, 2400
    dom1, grid1, s1 = setup(ctx, nxg, nyg, nxg, nyg)
    # (1) direct comparison with the checker, 4 subcycles
    orc.set_evp_parameters(DT, 4); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s1.items()}
    orc.evp(orc.make_domain(dom1, grid1), so)
    sg = {k: v.copy() for k, v in s1.items()}
    ctx.evp_init(grid1, ndte=4, krdg_partic=0, krdg_redist=0)
    ctx.evp(DT, sg)
    for k in PRIMARY + ("divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT", "strocnyT"):
        assert np.array_equal(sg[k], so[k]), k
    orc.set_strength_parameters()
    del so
    # (2) full ndte = 240: decomposition invariance 1 block vs 4 x 3 blocks
    a = {k: v.copy() for k, v in s1.items()}
    ctx.evp_init(grid1, ndte=240)
    ctx.evp(DT, a)
    one = {k: physical(dom1, a[k]) for k in PRIMARY + ("divu", "strocnxT")}
    for k in PRIMARY:
        assert np.isfinite(a[k]).all()
    assert 0.01 < np.abs(one["uvel"]).max() < 5.0
    del a, s1, grid1
    dom12, grid12, s12 = setup(ctx, nxg, nyg, 900, 800)
    assert dom12["nblocks"] == 12
    ctx.evp_init(grid12, ndte=240)
    ctx.evp_set_option("waves", 8); ctx.evp_set_option("rows_per_wave", 2)
    ctx.evp(DT, s12)
    for k in one:
        assert np.array_equal(physical(dom12, s12[k]), one[k]), k


def test_tenth_degree_rank_slab_against_checker(ctx, orc):
    """0.1 degree, ndte = 240, against the checker for ALL 240 subcycles: a full-width 3600 x 300 domain -- the slab
    one of 8 GPUs owns (BASELINE.json configs[4]) -- whole evp(dt) with the default (exp-using) strength, every
    output field; ~20 s of checker time.  Same kernels, tile shapes and launch pairing as the 3600 x 2400 run."""
    dom, grid, s = setup(ctx, 3600, 300, 3600, 300, perturb=0.1, land_frac=0.02)
    orc.set_evp_parameters(DT, 240); orc.set_strength_parameters()
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    ctx.evp_init(grid, ndte=240)
    assert ctx.evp_get_info("fused") == 1
    ctx.evp(DT, s)
    for k in PRIMARY + ("divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strength", "strocnxT", "strocnyT",
                        "strintx", "strinty", "iceumask"):
        assert relerr(s[k], so[k]) <= TOL_EXP, (k, relerr(s[k], so[k]))
    assert 0.01 < np.abs(s["uvel"]).max() < 5.0


def test_tenth_degree_24_hours(ctx):
    """BASELINE.json configs[4]: 0.1 degree, ndte = 240, 24 h = 24 steps of dt = 3600 s with the state
    resident on the device (velocity, stresses and masks carried from step to step).  The default path
    (K = 4 subcycles per sweep: 60 launches per step, the state in the sweep's pair layout in between) must
    reproduce one launch per subcycle (`fuse = 0` switches sweeps and pairs off: 5,760 launches) bit for bit, and the
    solution has to stay bounded."""
    nxg, nyg = 3600, 2400
    dom, grid, s0 = setup(ctx, nxg, nyg, nxg, nyg)
    res = []
    for fuse in (1, 0):
        s = {k: v.copy() for k, v in s0.items()}
        ctx.evp_init(grid, ndte=240)
        ctx.evp_set_option("fuse", fuse)
        assert ctx.evp_get_info("fused") == fuse
        ctx.evp_upload(s)
        for _ in range(24):
            ctx.evp_step(DT)
        ctx.evp_download(s)
        res.append({k: s[k] for k in PRIMARY + ("divu", "strocnxT", "iceumask")})
        del s
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    u = physical(dom, res[0]["uvel"])
    assert np.isfinite(u).all() and 0.01 < np.abs(u).max() < 5.0


def test_tenth_degree_thermo_sample(ctx, orc):
    """Batched thermo step at 3600x2400 x 5 categories; a random 1-in-64 sample of the columns
    of every category is recomputed by the checker."""
    ctx.thermo_init(); orc.init_thermo()
    ny, nx = 2402, 3602
    NC, NI, NS = 5, 4, 1
    z = lambda *s: np.zeros(s)
    b = dict(aicen=z(1, NC, ny, nx), trcrn=z(1, NC, 5, ny, nx), vicen=z(1, NC, ny, nx), vsnon=z(1, NC, ny, nx),
             eicen=z(1, NC * NI, ny, nx), esnon=z(1, NC * NS, ny, nx), lhcoef=z(1, NC, ny, nx),
             shcoef=z(1, NC, ny, nx), fswsfc=z(1, NC, ny, nx), fswint=z(1, NC, ny, nx), fswthrun=z(1, NC, ny, nx),
             Sswabs=z(1, NC, NS, ny, nx), Iswabs=z(1, NC, NI, ny, nx), mlt_onset=z(1, ny, nx), frz_onset=z(1, ny, nx))
    for k in lib.THERMO_FORCING:
        b[k] = z(1, ny, nx)
    for k in lib.THERMO_OUT:
        b[k] = z(1, NC, ny, nx)
    cols = {}
    for n in range(NC):
        a, icells, ii, jj = synth.thermo_columns(ny, nx, n, regime="mixed", seed=5, ice_frac=1.0)
        for k in ("aicen", "vicen", "vsnon", "lhcoef", "shcoef", "fswsfc", "fswint", "fswthrun"):
            b[k][0, n] = a[k]
        b["trcrn"][0, n] = a["trcrn"]; b["eicen"][0, n * NI:(n + 1) * NI] = a["eicen"]
        b["esnon"][0, n:n + 1] = a["esnon"]; b["Sswabs"][0, n] = a["Sswabs"]; b["Iswabs"][0, n] = a["Iswabs"]
        if n == 0:
            for k in lib.THERMO_FORCING + ("mlt_onset", "frz_onset"):
                b[k][0] = a[k]
            forcing0 = {k: a[k] for k in lib.THERMO_FORCING + ("mlt_onset", "frz_onset")}
        rng = np.random.default_rng(100 + n)
        pick = np.sort(rng.choice(icells, icells // 64, replace=False))
        li = np.zeros(nx * ny, np.int32); lj = np.zeros(nx * ny, np.int32)
        li[:len(pick)] = ii[pick]; lj[:len(pick)] = jj[pick]
        a.update(forcing0)
        cols[n] = (a, len(pick), li, lj)
    ctx.thermo_batch_alloc(nx, ny, 1)
    ctx.thermo_batch_upload(b)
    st = ctx.thermo_batch_step(DT, yday=150.0)
    assert st["l_stop"] == 0 and st["n_updates"] == 5 * 3600 * 2400
    ctx.thermo_batch_download(b)
    for n in range(NC):
        a, m, li, lj = cols[n]
        a["mlt_onset"] = a["mlt_onset"].copy(); a["frz_onset"] = a["frz_onset"].copy()
        assert orc.thermo_vertical(DT, m, li, lj, a, yday=150.0) == (0, 0, 0)
        jj, ii = lj[:m] - 1, li[:m] - 1
        for k, gpu in (("vicen", b["vicen"][0, n]), ("vsnon", b["vsnon"][0, n]), ("Tsfc", b["trcrn"][0, n, 0]),
                       ("eicen3", b["eicen"][0, n * NI + 2]), ("esnon", b["esnon"][0, n]), ("fsurfn", b["fsurfn"][0, n]),
                       ("congel", b["congel"][0, n]), ("flatn", b["flatn"][0, n])):
            cpu = {"Tsfc": a["trcrn"][0], "eicen3": a["eicen"][2], "esnon": a["esnon"][0]}.get(k, a.get(k))
            d = np.abs(gpu[jj, ii] - cpu[jj, ii]).max(); den = max(np.abs(cpu[jj, ii]).max(), 1e-4)
            assert d / den <= TOL_EXP, (n, k, d / den)


@pytest.mark.parametrize("ns", [3, 4], ids=["tripole", "tripoleT"])
def test_tenth_degree_width_with_a_tripole_fold(ctx, ns):
    """A 3600-wide grid with a tripole north boundary, ocean and ice up to the fold (COSIMA's 0.1-degree grid folds there):
    K = 4 subcycles per sweep + the band of top rows that carries the fold, against one launch per subcycle followed by the
    halo update with its fold (the path pinned to the reference on such grids, tests/tripole_evp_case.py): bit for bit."""
    nxg, nyg, ndte = 3600, 320, 24
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.03, land_rows=0), dom, north_ocean=True)
    s = synth.evp_state(grid, dom, cover="patchy")
    out = []
    for sweep in (0, 1):
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
        ctx.evp_set_option("resident", 0); ctx.evp_set_option("skew", sweep); ctx.evp_set_option("skew_fold", sweep)
        assert ctx.evp_get_info("skew_fold") == sweep
        ctx.evp(DT, sg)
        out.append(sg)
    assert np.abs(out[0]["uvel"][0, -3:]).max() > 1e-3
    for k in PRIMARY + ("strintx", "strocnx", "divu", "shear", "prs_sig"):
        assert np.array_equal(out[0][k], out[1][k]), (ns, k, np.argwhere(out[0][k] != out[1][k])[:5].tolist())


@pytest.mark.parametrize("ns", [3, 4], ids=["tripole", "tripoleT"])
def test_tenth_degree_width_tripole_grid_on_two_ranks(ctx, ns):
    """The same grid as two wide-halo slabs on two ranks (two contexts of this process, in-process link): the lower rank
    runs plain sweeps, the upper one sweeps + the band with the fold (3600 columns: the fold's four-kernel form), the
    overlap rows are refreshed every 8 subcycles.  Against the one-block domain through one launch per subcycle."""
    import threading
    from cice4_amd import lib
    nxg, nyg, ndte, R, H = 3600, 320, 24, 2, 8
    dom1 = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.03, land_rows=0)
    grid1 = synth.block_fields(gg, dom1, north_ocean=True)
    s1 = synth.evp_state(grid1, dom1, cover="patchy")
    ctx.evp_init(grid1, ndte=ndte, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 0); ctx.evp_set_option("skew", 0); ctx.evp_set_option("skew_fold", 0)
    ctx.evp(DT, s1)
    bar = threading.Barrier(R)
    out, errs = [None] * R, []

    def rank_fn(r):
        try:
            c = lib.Context(device=0); c.sync()
            dom = c.domain_create_slabs(nxg, nyg, R, ew=1, ns=ns, rank=r, nranks=R, overlap=H)
            c.comm_init_local(7300 + ns, r, R)
            grid = synth.block_fields(gg, dom, north_ocean=True)
            s = synth.evp_state(grid, dom, cover="patchy")
            c.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
            c.evp_set_option("resident", 0); c.evp_set_option("skew_min_cells", 0)
            assert c.evp_get_info("skew_fold" if r == R - 1 else "skew") == 1, r
            c.evp(DT, s)
            out[r] = (dom, s)
            bar.wait(timeout=300)
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))
            bar.abort()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    for k in ("uvel", "vvel", "stressp_1", "stressm_3", "stress12_4", "strintx", "divu", "shear", "prs_sig"):
        for r in range(R):
            dom, s = out[r]
            j0, jlo, olo, ohi = int(dom["j0"][0]), int(dom["jlo"][0]), int(dom["own_jlo"][0]), int(dom["own_jhi"][0])
            got = s[k][0, olo - 1:ohi, 1:-1]
            g0 = j0 + (olo - jlo)
            want = s1[k][0, 1 + g0:1 + g0 + (ohi - olo + 1), 1:-1]
            assert np.array_equal(got, want), (ns, k, r, np.argwhere(got != want)[:5].tolist())



# ---- BASELINE.json configs[3] and configs[4]: EIGHT ranks -----------------------------------------------------------------
# Eight ranks = eight contexts of this process on the one GPU, one host thread each (tests/ranks_case.py; the GPU boxes
# admit six processes on a card).  What never ran before round 5: interior ranks with two neighbours at both sizes,
# 48-row slabs under bench.auto_overlap's H, eight one-launch loops joined edge to edge, the sweep's tile lists with the
# extension trimmed on BOTH edges (Evp::tiles_for).

@pytest.mark.parametrize("mode", ["classic", "peer", "peer-tripole", "slabs", "slabs-sweep"])
def test_gx1_on_eight_ranks(orc, mode):
    """gx1 320 x 384, ndte 120, as 8 j-slabs of 48 rows (configs[3]) against the checker on the whole grid, bit for bit on
    every owned cell of u, v, the 12 stresses and the diagnostics:
      classic      one slab per rank, ghost rows exchanged after every subcycle (what the Fortran drop-in does under MPI,
                   /root/reference/mpi/ice_boundary.F90:1028-1417);
      peer         the whole subcycle loop in ONE launch per rank, tiles on a slab's first / last rows store their edge
                   velocities into the neighbour's exchange copies (6 interior ranks with two neighbours each);
      slabs        what `bench.py --gpus 8` runs at gx1: wide-halo slabs with bench.auto_overlap's H (24 rows of overlap on
                   48 owned ones), pairs of subcycles per launch, one refresh of u, v, 12 stresses every H subcycles;
      slabs-sweep  the 0.1-degree configuration's code path at gx1 size: H = 8, K = 4 sweeps over tile lists."""
    import importlib
    import ranks_case
    bench = importlib.import_module("bench")
    nxg, nyg, R, ndte = 320, 384, 8, 120
    if mode in ("peer", "peer-tripole"):
        # eight loops that wait for each other have to run at the same time: one hardware queue each (tests/ranks_peer_case.py)
        # (peer-tripole, round 5: the same under a tripole fold -- rank 7 runs the cross-rank loop with the fold inside)
        import os, subprocess, sys
        env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranks_peer_case.py"),
                            "1", str(R), str(nxg), str(nyg), str(ndte)] + (["3"] if mode == "peer-tripole" else []),
                           env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(r.stdout[-3000:]); print("\n".join(l[:600] for l in r.stderr.splitlines() if "amdgpu.ids" not in l)[-12000:])
        assert r.returncode == 0, "tests/ranks_peer_case.py failed (its output: captured stdout)"
        assert "bit-identical" in r.stdout
        return
    gg = synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.03, seed=31)
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, ndte, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    kw = {}
    if mode == "slabs":
        H = bench.auto_overlap(nxg, nyg // R)
        assert H == 24
        kw = dict(overlap=H)
    elif mode == "slabs-sweep":
        kw = dict(overlap=8, skew_k=4, min_cells=0, split=0)
    out = ranks_case.run_ranks(gg, R, mode.split("-")[0], ndte, DT, seed=31, cover="patchy", **kw)
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in PRIMARY + ("divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig"):
        want, got = ranks_case.owned(one, s1[k]), ranks_case.assemble(out, k, nxg, nyg)
        assert np.array_equal(got, want), (mode, k, np.argwhere(got != want)[:5].tolist())
    assert np.abs(s1["uvel"]).max() > 0.01


@pytest.mark.parametrize("split", [0, 1], ids=["one-launch", "refresh-beside-interior"])
def test_tenth_degree_on_eight_ranks(orc, split):
    """0.1 degree 3600 x 2400 as 8 wide-halo slabs of 300 rows (configs[4]'s decomposition: H = 8 overlap rows, K = 4 sweeps,
    refresh of u, v and the 12 stresses every 8 subcycles in one message per neighbour), ndte = 8 -- one sweep after a refresh
    and one in front of the next, i.e. both tile lists of Evp::tiles_for on top, bottom and interior ranks -- against the
    checker on the whole grid, bit for bit on every owned cell.  (ndte = 240 on one rank's slab against the checker:
    test_tenth_degree_rank_slab_against_checker; the 24-hour run: test_tenth_degree_24_hours.)"""
    import importlib
    import ranks_case
    bench = importlib.import_module("bench")
    nxg, nyg, R, ndte = 3600, 2400, 8, 8
    assert bench.auto_overlap(nxg, nyg // R) == 8
    gg = synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.02, seed=31)
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="full")
    orc.set_evp_parameters(DT, ndte, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    del grid1
    info = {}
    out = ranks_case.run_ranks(gg, R, "slabs", ndte, DT, seed=31, cover="full", overlap=8, skew_k=4, split=split, info=info, timeout=900)
    assert info["skew"] == 1 and info["skew_levels"] == 4
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in PRIMARY + ("divu", "shear", "prs_sig", "strintx"):
        want, got = ranks_case.owned(one, s1[k]), ranks_case.assemble(out, k, nxg, nyg)
        assert np.array_equal(got, want), (split, k, np.argwhere(got != want)[:5].tolist())
    assert 1e-4 < np.abs(s1["uvel"]).max() < 5.0


@pytest.mark.parametrize("npx,npy", [(2, 2), (6, 1), (4, 2)], ids=["2x2", "6-i-slabs", "4x2"])
def test_gx1_as_a_cartesian_layout_of_one_launch_loops(npx, npy):
    """The decompositions the reference and COSIMA configure under MPI (comp_ice:34-46: 2 x 2 tasks;
    bld/config.nci.access-om.360x300:7-8: 6 i-slabs; source/ice_blocks.F90:133-330), one block per rank, gx1 size, ndte 120:
    the whole subcycle loop as ONE launch per rank, the tiles on a block's four edges and corners exchanging with up to eight
    neighbouring ranks (round 5; before, such layouts ran one launch and one message per neighbour per subcycle).  Ranks =
    contexts of a child process (one hardware queue each), against the checker on the whole grid, bit for bit."""
    import os, subprocess, sys
    # (2 x 2: gx1.  More ranks on ONE device leave each launch an eighth / a sixth of the chip and every shader engine has
    #  to hold a workgroup of every launch at once: half the rows there -- 324 = a width that divides by 6)
    nxg, nyg, ndte = {2: (320, 384, 120), 6: (324, 192, 120), 4: (320, 192, 120)}[npx]
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranks_peer_case.py"),
                        str(npx), str(npy), str(nxg), str(nyg), str(ndte)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "bit-identical" in r.stdout
