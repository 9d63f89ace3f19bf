"""GPU parity tests of the EVP path (through the C-ABI) against the CPU checker.
Tolerances: stress/stepu/subcycle kernels contain only + - * / sqrt and are compiled
without FMA contraction -> required BIT-EXACT.  Whole evp(dt) passes through exp() in
ice_strength, which the device evaluates with glibc's own algorithm (cice4_amd/csrc/libm_exact.h):
also BIT-EXACT on a host whose glibc runs its FMA build of exp (conftest.TOL_EXP = 0); on any
other host the field-level bound 1e-10 of BASELINE.json applies."""
import os

import numpy as np
import pytest

from cice4_amd import lib, synth
from conftest import relerr, single_block_domain, TOL_EXP

pytestmark = pytest.mark.gpu

DT, NDTE = 3600.0, 120
TOL = TOL_EXP   # ice velocity and the 12 stress components (the fields BASELINE.json names)
TOL_DERIVED = TOL_EXP if TOL_EXP == 0.0 else 1e-8  # divergences / strain-rate diagnostics: differences of nearly cancelling
                    # stresses, which amplify the 1-ulp exp() difference in ice_strength
PRIMARY = ("uvel", "vvel") + synth.SIG_NAMES


def tol_of(k):
    return TOL if k in PRIMARY else TOL_DERIVED
EVP_OUT_FIELDS = ("uvel", "vvel", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig",
                  "strocnxT", "strocnyT", "strocnx", "strocny", "strintx", "strinty", "strairx",
                  "strairy", "fm", "strtltx", "strtlty") + synth.SIG_NAMES


def _setup(ctx, nxg, nyg, bsx, bsy, perturb=0.15, land_frac=0.05, cover="patchy", seed=1, moving=True):
    dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=perturb, land_frac=land_frac, seed=seed)
    grid = synth.block_fields(gg, dom)
    s = synth.evp_state(grid, dom, seed=seed, cover=cover, moving=moving)
    return dom, grid, s


def _lists_from_mask(mask_2d):
    jj, ii = np.nonzero(mask_2d)
    n = len(ii)
    li = np.zeros(mask_2d.size, np.int32); lj = np.zeros(mask_2d.size, np.int32)
    li[:n] = ii + 1; lj[:n] = jj + 1
    return n, li, lj


@pytest.mark.parametrize("ksub,damping", [(1, False), (NDTE, False), (NDTE, True), (7, True)])
def test_stress_and_stepu_bit_exact(ctx, orc, ksub, damping):
    nxg, nyg = 96, 70
    dom, grid, s = _setup(ctx, nxg, nyg, nxg, nyg)
    ny, nx = dom["ny"], dom["nx"]
    g = {k: np.ascontiguousarray(grid[k][0]) for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarear", "tinyarea", "uarear")}
    rng = np.random.default_rng(3)
    tmask = np.zeros((ny, nx), bool); tmask[1:, 1:] = rng.uniform(0, 1, (ny - 1, nx - 1)) < 0.7
    icellt, ti, tj = _lists_from_mask(tmask)
    strength = np.ascontiguousarray(rng.uniform(0, 3e4, (ny, nx)))
    uvel = np.ascontiguousarray(s["uvel"][0] + rng.uniform(-0.1, 0.1, (ny, nx)))
    vvel = np.ascontiguousarray(s["vvel"][0] + rng.uniform(-0.1, 0.1, (ny, nx)))
    orc.set_evp_parameters(DT, NDTE, damping)
    res = []
    for who in ("gpu", "cpu"):
        sig = [np.ascontiguousarray(s[n][0]).copy() for n in synth.SIG_NAMES]
        diag = {k: np.full((ny, nx), 7.0) for k in ("shear", "divu", "prs_sig", "rdg_conv", "rdg_shear")}
        str8 = np.full((8, ny, nx), 5.0)
        if who == "gpu":
            ctx.evp_stress(DT, NDTE, damping, ksub, icellt, ti, tj, uvel, vvel, g, strength, sig, diag, str8)
        else:
            orc.stress(ksub, icellt, ti, tj, uvel, vvel, g, strength, sig, diag, str8)
        res.append((sig, diag, str8))
    for k in range(12):
        assert np.array_equal(res[0][0][k], res[1][0][k]), synth.SIG_NAMES[k]
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    assert np.array_equal(res[0][2], res[1][2])
    # stepu on those str
    str8 = res[1][2]
    umask = np.zeros((ny, nx), bool); umask[1:-1, 1:-1] = rng.uniform(0, 1, (ny - 2, nx - 2)) < 0.8
    icellu, ui, uj = _lists_from_mask(umask)
    ins = {k: np.ascontiguousarray(rng.uniform(0.1, 1.0, (ny, nx))) for k in ("aiu", "waterx", "watery", "forcex", "forcey")}
    ins["uocn"] = np.ascontiguousarray(s["uocn"][0]); ins["vocn"] = np.ascontiguousarray(s["vocn"][0])
    ins["umassdtei"] = np.ascontiguousarray(rng.uniform(5, 80, (ny, nx)))
    ins["fm"] = np.ascontiguousarray(rng.uniform(-0.3, 0.3, (ny, nx)))
    out = []
    for who in ("gpu", "cpu"):
        io = [np.full((ny, nx), 3.0) for _ in range(4)] + [uvel.copy(), vvel.copy()]
        f = ctx.evp_stepu if who == "gpu" else orc.stepu
        f(icellu, ui, uj, ins["aiu"], str8, ins["uocn"], ins["vocn"], ins["waterx"], ins["watery"],
          ins["forcex"], ins["forcey"], ins["umassdtei"], ins["fm"], g["uarear"], *io)
        out.append(io)
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def _run_both(ctx, orc, dom, grid, s, damping=False, ndte=NDTE):
    orc.set_evp_parameters(DT, ndte, damping)
    orc.set_strength_parameters()
    d = orc.make_domain(dom, grid)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(d, so)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, evp_damping=damping)
    ctx.evp(DT, sg)
    return sg, so


@pytest.mark.parametrize("cover,bs", [("full", (96, 70)), ("patchy", (96, 70)), ("patchy", (48, 35)),
                                      ("patchy", (32, 24))])
def test_whole_evp_matches_oracle(ctx, orc, cover, bs):
    """evp(dt), 120 subcycles, non-uniform grid with islands; 1, 4 and 9 blocks (the 32x24
    decomposition has padded last blocks)."""
    dom, grid, s = _setup(ctx, 96, 70, bs[0], bs[1], cover=cover)
    sg, so = _run_both(ctx, orc, dom, grid, s)
    assert np.array_equal(sg["iceumask"], so["iceumask"])
    for k in EVP_OUT_FIELDS:
        assert relerr(sg[k], so[k]) <= tol_of(k), (k, relerr(sg[k], so[k]))
    assert np.abs(so["uvel"]).max() > 0.01  # the case is not trivially at rest


def test_evp_damping_and_small_ndte(ctx, orc):
    dom, grid, s = _setup(ctx, 64, 40, 64, 40, cover="patchy", seed=5)
    sg, so = _run_both(ctx, orc, dom, grid, s, damping=True, ndte=7)  # odd ndte: ping-pong parity
    for k in EVP_OUT_FIELDS:
        assert relerr(sg[k], so[k]) <= tol_of(k), k


@pytest.mark.parametrize("bs,shape", [((48, 35), (8, 1)), ((96, 70), (4, 4)), ((32, 24), (4, 2)), ((96, 7), (16, 1))])
def test_whole_evp_bit_exact_without_transcendentals(ctx, orc, bs, shape):
    """With krdg_partic = 0 and krdg_redist = 0 (Thorndike 75 / Hibler 80: ice_mechred.F90
    :881-895, :937-952) ice_strength needs no exp(), so the WHOLE evp(dt) -- prep, masks,
    T<->U averaging, strength, 120 fused subcycles with halos, finish -- must agree with the
    checker BIT FOR BIT (1, 4, 9 and 10 blocks; several tile shapes)."""
    dom, grid, s = _setup(ctx, 96, 70, bs[0], bs[1], cover="patchy", seed=9)
    orc.set_evp_parameters(DT, NDTE, False)
    orc.set_strength_parameters(1, 0, 0, 4.0)
    d = orc.make_domain(dom, grid)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(d, so)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("waves", shape[0]); ctx.evp_set_option("rows_per_wave", shape[1])
    ctx.evp_set_option("fuse", 0); ctx.evp_set_option("resident", 0)     # this test is about k_subcycle's tile shapes
    ctx.evp(DT, sg)
    orc.set_strength_parameters()
    for k in EVP_OUT_FIELDS + ("iceumask",):
        assert np.array_equal(sg[k], so[k]), k


def test_results_independent_of_tile_shape_and_graph(ctx, orc):
    """Tile shape (wavefronts per workgroup x rows per wavefront) and hipGraph replay are pure
    scheduling choices: results must be bit-identical across all of them."""
    dom, grid, s = _setup(ctx, 96, 70, 48, 35, cover="patchy", seed=9)
    first = None
    for waves, rows in ((8, 1), (4, 2), (4, 4), (4, 8), (8, 2), (8, 4), (4, 1), (16, 1), (16, 2)):
        for graph in ((1, 0) if (waves, rows) == (8, 1) else (1,)):
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=NDTE)
            ctx.evp_set_option("waves", waves)
            ctx.evp_set_option("rows_per_wave", rows)
            ctx.evp_set_option("use_graph", graph)
            ctx.evp(DT, sg)
            if first is None:
                first = sg
            else:
                for k in EVP_OUT_FIELDS:
                    assert np.array_equal(sg[k], first[k]), (waves, rows, graph, k)


def test_stepwise_api_equals_dropin(ctx, orc):
    dom, grid, s = _setup(ctx, 64, 40, 32, 40, cover="full", seed=2)
    a = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE)
    ctx.evp(DT, a)
    b = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE)
    ctx.evp_upload(b)
    ctx.evp_prepare(DT)
    nt, nu = ctx.evp_active_cells()
    assert nt > 0 and nu > 0
    ms = ctx.evp_subcycles(1, 50, timed=True)
    assert ms > 0.0
    ctx.evp_subcycles(51, NDTE - 50)
    ctx.evp_finish()
    ctx.evp_download(b)
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(a[k], b[k]), k


def test_three_steps_carry_state_bit_exact(ctx, orc):
    """Three consecutive evp(dt) calls from rest (iceumask, velocities, stresses and the
    ping-pong buffers carried from step to step), in the exp-free strength configuration so
    that the comparison is bit for bit on every host (the default strength goes through exp():
    test_three_steps_default_strength_bit_exact)."""
    dom, grid, s = _setup(ctx, 64, 40, 64, 40, cover="patchy", seed=4, moving=False)
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    d = orc.make_domain(dom, grid)
    so = {k: v.copy() for k, v in s.items()}
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    for step in range(3):
        orc.evp(d, so)
        ctx.evp(DT, sg)
        for k in EVP_OUT_FIELDS + ("iceumask",):
            assert np.array_equal(sg[k], so[k]), (step, k)
    orc.set_strength_parameters()


def test_three_steps_default_strength_bit_exact(ctx, orc):
    """The same spin-up with the DEFAULT strength (krdg_partic = 1: exp() in the participation function).
    Round 1 could check this for one step only: an exp() that differs from glibc's in the last bit is
    amplified to ~1e-10 after two steps.  With glibc's algorithm on the device: bit for bit, three steps."""
    dom, grid, s = _setup(ctx, 64, 40, 64, 40, cover="patchy", seed=4, moving=False)
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters()
    d = orc.make_domain(dom, grid)
    so = {k: v.copy() for k, v in s.items()}
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE)
    for step in range(3):
        orc.evp(d, so)
        ctx.evp(DT, sg)
        for k in EVP_OUT_FIELDS + ("iceumask",):
            assert relerr(sg[k], so[k]) <= TOL_EXP, (step, k, relerr(sg[k], so[k]))


def test_halo_update_through_device(ctx):
    dom = ctx.domain_create(60, 44, 20, 11, ew=1, ns=0)
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (3, dom["nblocks"], dom["ny"], dom["nx"]))
    want = a.copy().reshape(3, -1)
    want[:, dom["hdst"]] = want[:, dom["hsrc"]]
    got = a.copy()
    ctx.halo_update(got)
    assert np.array_equal(got.reshape(3, -1), want)
    ai = rng.integers(0, 9, (dom["nblocks"], dom["ny"], dom["nx"])).astype(np.int32)
    wi = ai.copy().reshape(-1); wi[dom["hdst"]] = wi[dom["hsrc"]]
    ctx.halo_update(ai)
    assert np.array_equal(ai.reshape(-1), wi)


def test_errors_are_reported_not_fatal(ctx):
    c2 = lib.Context()
    with pytest.raises(lib.CiceError):
        c2.evp_step(DT)          # no domain / init
    with pytest.raises(lib.CiceError):
        c2.domain_create(10, 10, 20, 20, ew=7)


@pytest.mark.parametrize("cfg,kind", [("gx3b4", "dropin"), ("gx3b4", "dropinmpi"), ("padx", "dropin")])
def test_fortran_dropin_module_inside_reference_callers(orc, cfg, kind):
    """The drop-in proof: the reference's own compiled modules (ice_state, ice_flux, ice_grid,
    ice_domain, ... and the capture wrapper that calls `evp(dt)`) linked with OUR
    cice4_amd/fortran/ice_dyn_evp.F90 instead of the reference's.  `call evp(dt)` then goes
    Fortran -> ISO_C_BINDING shim -> libcice4_amd.so -> GPU, on the reference's own module
    arrays and block layout, and must reproduce the checker (pinned to the pure reference
    bit for bit by tests/test_oracle_vs_ref.py).
    kind 'dropinmpi': the same with the reference's mpi/ modules (MPICH, this process is a 1-rank MPI
    job): our boundary module then also broadcasts the RCCL id over MPI_COMM_ICE and creates the RCCL
    communicator, as every task of an MPI build does.
    cfg 'padx': 3x3 blocks with padded last blocks in arrays dimensioned max_blocks = 12 > 9, as on a
    task of an MPI run that owns fewer blocks than the largest share."""
    import tempfile
    from __graft_entry__ import REF_CONFIGS
    from oracle import refapi
    if not refapi.available(cfg, kind):
        pytest.skip(f"oracle/_ref/libcice_{kind}_{cfg}.so not built")
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg, kind=kind)
    nb = ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE)
    dom = lib.Context().domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    assert nb == dom["nblocks"] <= mxb == ref.max_blocks
    ny, nx = dom["ny"], dom["nx"]

    def full(a):       # host arrays carry max_blocks blocks; the task's blocks are the first nb
        out = np.zeros((mxb * (a.shape[0] // nb),) + a.shape[1:], a.dtype)
        out[:a.shape[0]] = a
        return out

    def mine(a, per=1):
        return a[:nb * per]

    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05), dom)
    s = synth.evp_state(grid, dom, cover="patchy")
    for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear",
              "uarear", "tinyarea", "fcor", "HTN", "HTE"):
        ref.set(k, full(grid[k]))
    ref.set("tmask", full(grid["tmask"].astype(float))); ref.set("umask", full(grid["umask"].astype(float)))
    ref.set_strength_parameters(1, 0, 0, 4.0)      # exp-free strength: bit-for-bit comparison
    ref.evp_gpu_setup()
    for k in ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty",
              "uvel", "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx",
              "strinty") + synth.SIG_NAMES:
        ref.set(k, full(s[k]))
    ref.set("iceumask", full(s["iceumask"].astype(float)))
    ref.set("aicen", full(s["aicen"].reshape(-1, ny, nx))); ref.set("vicen", full(s["vicen"].reshape(-1, ny, nx)))
    ref.evp(DT)
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    orc.set_strength_parameters()
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(mine(ref.get(k)), so[k]), k
    assert np.array_equal(mine(ref.get("iceumask")), so["iceumask"])
    assert np.abs(so["uvel"]).max() > 0.01


def test_standalone_fortran_driver(ctx, tmp_path):
    """Host code in Fortran: cice4_amd/fortran/evp_driver (amdflang) calls the C-ABI through the
    ISO_C_BINDING shim; its evp(dt) result must equal the ctypes-driven one bit for bit."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(lib.HERE), "cice4_amd", "fortran", "evp_driver")
    if not os.path.exists(exe):
        pytest.skip("cice4_amd/fortran/evp_driver not built")
    dom, grid, s = _setup(ctx, 96, 70, 48, 35, cover="patchy", seed=12)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([96, 70, 48, 35, 1, 0, NDTE], np.int32).tofile(f)
        np.array([DT]).tofile(f)
        for k in lib.EVP_GRID[:14]:
            grid[k].tofile(f)
        grid["tmask"].astype(np.int32).tofile(f); grid["umask"].astype(np.int32).tofile(f)
        for k in ("aice", "vice", "vsno", "aice0", "aicen", "vicen", "strairxT", "strairyT", "uocn", "vocn",
                  "ss_tltx", "ss_tlty", "uvel", "vvel") + synth.SIG_NAMES:
            s[k].tofile(f)
        s["iceumask"].astype(np.int32).tofile(f)
        for k in ("fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty"):
            s[k].tofile(f)
    r = subprocess.run([exe, fin, fout, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE)
    ctx.evp(DT, sg); ctx.evp(DT, sg)
    n = sg["uvel"].size
    with open(fout, "rb") as f:
        for k in ("uvel", "vvel") + synth.SIG_NAMES:
            assert np.array_equal(np.fromfile(f, np.float64, n).reshape(sg[k].shape), sg[k]), k
        assert np.array_equal(np.fromfile(f, np.int32, n).reshape(sg["iceumask"].shape), sg["iceumask"])
        for k in ("fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty", "strairx", "strairy",
                  "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT", "strocnyT"):
            assert np.array_equal(np.fromfile(f, np.float64, n).reshape(sg[k].shape), sg[k]), k


def test_metric_derivation_is_verified_and_exact(ctx, orc):
    """The kernel may recompute the nine T-cell metrics from HTN/HTE only after the host has
    verified the identity bit for bit; with it on or off, and on a grid where the identity does
    NOT hold (so that it must switch itself off), results equal the checker bit for bit."""
    dom, grid, s = _setup(ctx, 96, 70, 48, 35, cover="patchy", seed=21)
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    outs = []
    for variant in ("derive", "off", "inconsistent"):
        g = dict(grid)
        if variant == "inconsistent":
            g["dxt"] = grid["dxt"] * (1.0 + 1e-9)      # metrics no longer functions of HTN/HTE
            g["tarear"] = grid["tarear"].copy()
        so = {k: v.copy() for k, v in s.items()}
        orc.evp(orc.make_domain(dom, g), so)
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp_init(g, ndte=NDTE, krdg_partic=0, krdg_redist=0)
        if variant == "off":
            ctx.evp_set_option("derive_metrics", 0)
        assert ctx.evp_get_info("derive_metrics") == (1 if variant == "derive" else 0)
        ctx.evp(DT, sg)
        for k in EVP_OUT_FIELDS:
            assert np.array_equal(sg[k], so[k]), (variant, k)
        outs.append(sg)
    orc.set_strength_parameters()
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("case", ["no_ice", "one_cell", "all_land", "ndte1"])
def test_degenerate_inputs(ctx, orc, case):
    """Edge cases: no ice anywhere (empty lists), a single ice cell, an all-land grid, ndte = 1."""
    dom, grid, s = _setup(ctx, 40, 30, 20, 15, cover="patchy", seed=6)
    ndte = 1 if case == "ndte1" else 6
    if case in ("no_ice", "one_cell"):
        for k in ("aice", "vice", "vsno", "aicen", "vicen", "strairxT", "strairyT"):
            s[k][...] = 0.0
        s["aice0"][...] = 1.0
        s["iceumask"][...] = 0; s["uvel"][...] = 0.0; s["vvel"][...] = 0.0
        if case == "one_cell":
            s["aicen"][1, 2, 7, 9] = 0.8; s["vicen"][1, 2, 7, 9] = 1.5
            s["aice"][1, 7, 9] = 0.8; s["vice"][1, 7, 9] = 1.5; s["aice0"][1, 7, 9] = 0.2
            s["strairxT"][1, 7, 9] = 0.1
    if case == "all_land":
        grid = dict(grid); grid["tmask"] = np.zeros_like(grid["tmask"]); grid["umask"] = np.zeros_like(grid["umask"])
    orc.set_evp_parameters(DT, ndte, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
    ctx.evp(DT, sg)
    orc.set_strength_parameters()
    for k in EVP_OUT_FIELDS + ("iceumask",):
        assert np.array_equal(sg[k], so[k]), (case, k)
    if case in ("no_ice", "all_land"):
        assert ctx.evp_active_cells() == (0, 0)


def test_message_path_on_one_gpu(orc, monkeypatch):
    """pack -> RCCL send/recv -> unpack (the off-rank halo path) exercised on a single GPU: with
    CICE4_AMD_SELF_COMM the copies between different blocks of the rank are routed through
    messages to the own rank (1-rank RCCL communicator).  Whole evp(dt) on 6 blocks, eager and
    hipGraph-captured (RCCL calls inside the graph), must equal the checker bit for bit."""
    monkeypatch.setenv("CICE4_AMD_SELF_COMM", "1")
    c = lib.Context()
    c.sync()
    dom = c.domain_create(96, 70, 32, 35, ew=1, ns=0)
    assert dom["nsend"] == 1 and dom["nsend_elems"] == dom["nrecv_elems"] > 0
    c.comm_init(c.comm_unique_id(), 0, 1)
    # the checker needs the plain on-rank list of the same decomposition
    monkeypatch.delenv("CICE4_AMD_SELF_COMM")
    dom_plain = lib.Context().domain_create(96, 70, 32, 35, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(96, 70, perturb=0.15, land_frac=0.05, seed=1), dom_plain)
    s = synth.evp_state(grid, dom_plain, seed=1, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom_plain, grid), so)
    orc.set_strength_parameters()
    for graph in (0, 1):
        sg = {k: v.copy() for k, v in s.items()}
        c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
        c.evp_set_option("use_graph", graph)
        c.evp_set_option("comm_graph", graph)     # multi-rank loops are eager unless asked otherwise
        c.evp(DT, sg)
        for k in EVP_OUT_FIELDS + ("iceumask",):
            assert np.array_equal(sg[k], so[k]), (graph, k)
    a = np.random.default_rng(0).uniform(0, 1, (2, dom["nblocks"], dom["ny"], dom["nx"]))
    want = a.copy().reshape(2, -1); want[:, dom_plain["hdst"]] = want[:, dom_plain["hsrc"]]
    c.halo_update(a)
    assert np.array_equal(a.reshape(2, -1), want)


@pytest.mark.parametrize("selfcomm", [False, True])
def test_multi_level_halo_resident_on_device(monkeypatch, selfcomm):
    """bound_state (source/ice_state.F90:162-217) updates aicen, trcrn, vicen, vsnon, eicen, esnon = 65 levels;
    ice_HaloUpdate3DR8/4DR8 (mpi/ice_boundary.F90:2216,3587) send all levels of a field in one message per
    neighbour.  cice_halo_update_dev_r8 does the same on a device-resident field: 65 levels, one update, no
    staging -- on-rank copies, and (CICE4_AMD_SELF_COMM) through pack -> one RCCL message -> unpack."""
    if selfcomm:
        monkeypatch.setenv("CICE4_AMD_SELF_COMM", "1")
    c = lib.Context(); c.sync()
    dom = c.domain_create(60, 44, 20, 11, ew=1, ns=0)
    if selfcomm:
        assert dom["nsend"] == 1
        c.comm_init(c.comm_unique_id(), 0, 1)
        monkeypatch.delenv("CICE4_AMD_SELF_COMM")
    plain = lib.Context().domain_create(60, 44, 20, 11, ew=1, ns=0)
    rng = np.random.default_rng(5)
    for nlev, dt in ((65, np.float64), (3, np.float64), (2, np.int32), (65, np.float64)):
        a = (rng.uniform(0, 1, (nlev, dom["nblocks"], dom["ny"], dom["nx"])) * 1000).astype(dt)
        want = a.copy().reshape(nlev, -1); want[:, plain["hdst"]] = want[:, plain["hsrc"]]
        b = a.copy(); a0 = a.copy()
        c.halo_update_resident(a)
        assert np.array_equal(a.reshape(nlev, -1), want), (nlev, dt)
        c.halo_update(b)                                  # host form: one update for all levels as well
        assert np.array_equal(b.reshape(nlev, -1), want), (nlev, dt)
        # the reference's array layout (block outermost): only the frame of each block crosses PCIe, the update runs
        # on the gathered frame (a second set of lists, addressed by frame position) -- messages included
        nb = dom["nblocks"]
        blk = np.ascontiguousarray(a0.reshape(nlev, nb, dom["ny"], dom["nx"]).transpose(1, 0, 2, 3))
        inside = blk.copy()
        c.halo_update_blocked(blk)
        got = np.ascontiguousarray(blk.transpose(1, 0, 2, 3)).reshape(nlev, -1)
        assert np.array_equal(got, want), ("blocked", nlev, dt)
        untouched = np.ones(got.shape[1], bool); untouched[plain["hdst"]] = False
        assert np.array_equal(got[:, untouched], np.ascontiguousarray(inside.transpose(1, 0, 2, 3)).reshape(nlev, -1)[:, untouched])


def _owned(dom, f):
    """global physical field from the OWNED rows of (possibly overlapping) slab blocks"""
    g = np.zeros((dom["nyg"], dom["nxg"]))
    for b in range(dom["nblocks"]):
        r0 = dom["j0"][b] + (dom["own_jlo"][b] - dom["jlo"][b])
        nr = dom["own_jhi"][b] - dom["own_jlo"][b] + 1
        g[r0:r0 + nr, :] = f[b, dom["own_jlo"][b] - 1:dom["own_jhi"][b], dom["ilo"][b] - 1:dom["ihi"][b]]
    return g


@pytest.mark.parametrize("overlap,selfcomm,skew_k", [(0, False, 0), (1, False, 0), (2, False, 0), (4, False, 0), (7, False, 0),
                                                     (4, True, 0), (6, True, 0), (7, True, 0),
                                                     (8, False, 4), (8, True, 4), (12, True, 4), (6, True, 3)])
def test_wide_halo_slabs_equal_single_domain(orc, monkeypatch, overlap, selfcomm, skew_k):
    """Wide-halo j-slabs: the overlap rows are recomputed and refreshed (u, v, 12 sigma in one
    message) only every `overlap` subcycles.  Owned rows must equal the single-domain checker run
    bit for bit; with CICE4_AMD_SELF_COMM the refresh goes through pack/RCCL/unpack.
    skew_k: K-subcycle SWEEPS between the refreshes, the overlap a multiple of K -- H = 8 with K = 4 is what
    bench.py --gpus N picks for slabs of the 0.1-degree grid (bench.auto_overlap)."""
    nxg, nyg, nb = 96, 72, 4
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=31)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    if selfcomm:
        monkeypatch.setenv("CICE4_AMD_SELF_COMM", "1")
    c = lib.Context(); c.sync()
    dom = c.domain_create_slabs(nxg, nyg, nb, ew=1, ns=0, overlap=overlap)
    if selfcomm:
        assert dom["nsend"] == 1
        c.comm_init(c.comm_unique_id(), 0, 1)
    grid = synth.block_fields(gg, dom)
    s = synth.evp_state(grid, dom, seed=31, cover="patchy")
    c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    if skew_k:
        c.evp_set_option("skew_min_cells", 0); c.evp_set_option("skew_levels", skew_k)
        assert c.evp_get_info("skew") == 1 and overlap % skew_k == 0
    c.evp(DT, s)
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig") + synth.SIG_NAMES:
        assert np.array_equal(_owned(dom, s[k]), _owned(one, s1[k])), (overlap, selfcomm, skew_k, k)
    nt, nu = c.evp_active_cells()
    c1.evp_init(grid1, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    s1b = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    c1.evp_upload(s1b); c1.evp_prepare(DT)
    nt1, nu1 = c1.evp_active_cells()
    assert nu == nu1                                # overlap rows are not counted twice
    assert nt1 <= nt <= nt1 + nb * (nxg + 2)        # T lists: each block also lists its N/E ghost ring


_LINK = [1000]


@pytest.mark.parametrize("mode,R,nyg", [("classic", 2, 72), ("peer", 2, 72), ("peer", 3, 72), ("peer", 2, 16), ("slabs4", 2, 72),
                                       ("slabs4", 3, 72), ("slabs6-sweep", 2, 96), ("peer-cyclic", 2, 40),
                                       ("slabs8-sweep4", 2, 128), ("slabs8-sweep4", 3, 192), ("slabs4-sweep4", 2, 96),
                                       ("slabs8-sweep4-nosplit", 2, 128)])
def test_ranks_in_one_process(orc, mode, R, nyg):
    """The multi-rank path with R ranks = R contexts of this process on the one GPU, one host thread each, messages through
    the in-process link (cice_comm_init_local: pack kernel -> host mailbox -> unpack kernel), against the single-domain
    checker run, bit for bit on every owned cell:
      classic  one slab per rank, ghost rows exchanged after every subcycle (what the Fortran drop-in does under MPI);
      peer     the WHOLE subcycle loop as one launch per rank: the tiles on a slab's first / last rows exchange their
               edge velocities with the neighbouring rank's tiles by stores into the neighbour's exchange copies and
               progress words (cice_evp_peer_export / _connect: here plain device pointers), no message inside the loop;
               2 and 3 ranks (a middle rank has two neighbours), slabs one tile tall, cyclic north-south (both
               neighbours are the same rank);
      slabs    wide-halo slabs as bench.py --gpus N cuts them (refresh of u, v, 12 sigma every H subcycles), pairs of
               subcycles per launch and, "sweep", K subcycles per sweep between the refreshes (K = 3, "sweep4": K = 4 with
               H = 8 -- what bench.auto_overlap picks -- and H = 4); the sweep in front of a refresh is split into edge
               and interior launches, the refresh overlapping the interior ("nosplit": the one-launch form)."""
    import threading
    nxg = 96
    ns = 1 if mode == "peer-cyclic" else 0
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    # (cyclic north-south: ocean and ice across the seam, or the wrap carries nothing)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=31, land_rows=0 if ns == 1 else 2)
    grid1 = synth.block_fields(gg, dom1, ns_cyclic=(ns == 1))
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    _LINK[0] += 1
    link = _LINK[0]
    bar = threading.Barrier(R)
    exports, out, errs = [None] * R, [None] * R, []

    def rank_fn(r):
        try:
            c = lib.Context(device=0); c.sync()
            bar.wait(timeout=60)         # (every rank's main stream before anybody's copy streams: tests/ranks_case.py)
            if mode.startswith("slabs"):
                H = int(mode[5])
                dom = c.domain_create_slabs(nxg, nyg, R, ew=1, ns=0, rank=r, nranks=R, overlap=H)
            else:
                dom = c.domain_create(nxg, nyg, nxg, nyg // R, ew=1, ns=ns, rank=r, npx=1, npy=R)
            assert dom["nblocks"] == 1 and dom["nsend"] >= 1
            c.comm_init_local(link, r, R)
            grid = synth.block_fields(gg, dom, ns_cyclic=(ns == 1))
            s = synth.evp_state(grid, dom, seed=31, cover="patchy")
            c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
            if mode.startswith("peer"):
                c.evp_set_option("resident_peer_share", R)
                exports[r] = c.evp_peer_export()
                bar.wait(timeout=60)
                if r > 0 or ns == 1:
                    c.evp_peer_connect(0, exports[(r - 1) % R])
                if r < R - 1 or ns == 1:
                    c.evp_peer_connect(1, exports[(r + 1) % R])
                assert c.evp_get_info("resident_peer") == 1
                # what the neighbour writes or polls is fine-grained device memory (coherent across devices during a launch)
                assert c.evp_get_info("resident_peer_fine") == (0 if os.environ.get("CICE4_AMD_PEER_COARSE") == "1" else 1)
                bar.wait(timeout=60)
            else:
                c.evp_set_option("resident", 0)
                if "sweep" in mode:
                    c.evp_set_option("skew_min_cells", 0); c.evp_set_option("skew_levels", 4 if "sweep4" in mode else 3)
                    assert c.evp_get_info("skew") == 1
                    # the sweep in front of every refresh runs as two launches: the edge segments, followed by the
                    # refresh, on the main stream; the interior beside them on a second one (round 4)
                    # (off by default -- on one GPU it costs more than it hides; bench.py --gpus N decides by timing)
                    c.evp_set_option("skew_split", 0 if mode.endswith("nosplit") else 1)
                    assert c.evp_get_info("skew_split") == (0 if mode.endswith("nosplit") else 1)
                    assert c.evp_get_info("skew_trim_ext") == 1    # extension rows trimmed to what the next sweeps need
                else:
                    c.evp_set_option("skew", 0)
            if mode.startswith("peer"):   # (no rank's loop starts while another rank's uploads are queued: tests/ranks_case.py)
                c.evp_upload(s); bar.wait(timeout=120)
                c.evp_step(DT); bar.wait(timeout=120)
                c.evp_download(s)
                assert c.evp_get_info("resident_peer") == 1, "the cross-rank loop timed out and fell back"
            else:
                c.evp(DT, s)
            out[r] = (dom, s)
            bar.wait(timeout=120)        # nobody frees buffers a neighbour may still be writing to
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))
            bar.abort()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig") + synth.SIG_NAMES:
        want = _owned(one, s1[k])
        got = np.zeros_like(want)
        for r in range(R):
            dom, s = out[r]
            part = _owned(dom, s[k])
            rows = slice(int(dom["j0"][0] + dom["own_jlo"][0] - dom["jlo"][0]),
                         int(dom["j0"][0] + dom["own_jhi"][0] - dom["jlo"][0]) + 1)
            got[rows] = part[rows]
        assert np.array_equal(got, want), (mode, R, k, np.argwhere(got != want)[:5])
    # ghost rows a neighbour owns are current after evp(dt) (the last exchange of the loop, ice_dyn_evp.F90:397-402)
    if not mode.startswith("slabs"):
        for r in range(R):
            dom, s = out[r]
            j0, jlo, jhi = int(dom["j0"][0]), int(dom["jlo"][0]), int(dom["jhi"][0])
            for gj, jg in ((jlo - 2, j0 - 1), (jhi, j0 + (jhi - jlo) + 1)):     # array row, global row
                if ns == 1:
                    jg %= nyg
                if 0 <= jg < nyg:
                    assert np.array_equal(s["uvel"][0, gj, 1:-1], s1["uvel"][0, jg + 1, 1:-1]), (mode, r, gj)


@pytest.mark.parametrize("mode,npx,npy,nxg,nyg,blocks", [("peer", 2, 2, 96, 72, (1, 1)), ("classic", 2, 2, 96, 72, (1, 1)),
                                                          ("peer", 2, 1, 130, 40, (1, 1)), ("peer", 4, 1, 128, 30, (1, 1)),
                                                          ("peer", 1, 4, 70, 64, (1, 1)), ("peer", 2, 2, 20, 16, (1, 1)),
                                                          ("peer", 2, 1, 96, 72, (1, 2)), ("peer", 1, 2, 96, 72, (2, 2)),
                                                          ("peer", 2, 2, 144, 80, (3, 2)), ("classic", 2, 1, 96, 72, (1, 2))])
def test_cartesian_layouts_of_ranks_in_one_process(orc, mode, npx, npy, nxg, nyg, blocks):
    """One block per rank in a CARTESIAN layout (source/ice_blocks.F90:133-330: 2 x 2 tasks as comp_ice:34-46 gives the MPI
    build, i-slabs as bld/config.nci.access-om.360x300:7-8), ranks = contexts of this process: the cross-rank one-launch loop
    with EAST-WEST and DIAGONAL neighbours (round 5: cice_evp_peer_ranks / cice_evp_peer_connect_rank; two task columns on a
    cyclic grid: the eastern and the western neighbour are the same rank) against the per-subcycle message path ("classic")
    and the checker on the whole grid, bit for bit; blocks narrower and shorter than a tile; SEVERAL blocks per rank
    (`blocks`: tiles numbered block by block, ghost cells between a rank's own blocks forwarded on the device)."""
    import ranks_case
    R = npx * npy
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=31)
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    out = ranks_case.run_ranks(gg, R, mode, NDTE, DT, seed=31, cover="patchy", npx=npx, blocks=blocks)
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig") + synth.SIG_NAMES:
        want, got = _owned(one, s1[k]), ranks_case.assemble_blocks(out, k, nxg, nyg)
        assert np.array_equal(got, want), (mode, npx, npy, k, np.argwhere(got != want)[:5].tolist())


@pytest.mark.parametrize("mode", ["peer", "classic"])
def test_ranks_with_an_eliminated_land_block(orc, mode):
    """A block -> rank map as the reference's distributions produce them (source/ice_distribution.F90: blocks without an ocean
    cell are dropped): 4 x 3 blocks of 24 x 24 on three ranks, one all-land block eliminated, the ranks' blocks scattered.
    The ghost cells that face the eliminated block have no producer and keep the fill value of the first halo update; the
    cross-rank one-launch loop (several blocks per rank, neighbours by rank) and the per-subcycle message path against the
    checker on the whole grid, bit for bit on every ocean cell."""
    import ranks_case
    nxg, nyg, bs, R = 96, 72, 24, 3
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=31)
    gg["hm"][24:48, 48:72] = 0.0                         # block (ib = 2, jb = 1): all land
    owner = np.array([0, 1, 2, 0,  1, 2, -1, 1,  2, 0, 1, 2], np.int32)       # global block g = jb * 4 + ib
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    out = ranks_case.run_ranks(gg, R, mode, NDTE, DT, seed=31, cover="patchy", block_map=(bs, bs, owner))
    assert sorted(int(d["nblocks"]) for d, _ in out) == [3, 4, 4]
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    keep = np.ones((nyg, nxg), bool); keep[24:48, 48:72] = False
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strintx", "prs_sig") + synth.SIG_NAMES:
        want, got = _owned(one, s1[k]), ranks_case.assemble_blocks(out, k, nxg, nyg)
        assert np.array_equal(got[keep], want[keep]), (mode, k, np.argwhere((got != want) & keep)[:5].tolist())
    assert np.abs(s1["uvel"]).max() > 0.01


@pytest.mark.parametrize("ns", [3, 4])
@pytest.mark.parametrize("mode,R,nyg", [("slabs0", 2, 72), ("slabs4", 2, 72), ("slabs4", 3, 96), ("slabs6-sweep", 2, 96),
                                       ("slabs6-sweep", 3, 144), ("slabs6-sweep4", 2, 96), ("peer", 2, 72), ("peer", 3, 96),
                                       ("peer", 3, 36), ("peer-W11", 2, 72), ("peer-W8", 3, 96), ("peer-W6", 2, 72)])
def test_tripole_grid_cut_into_slabs(ctx, ns, mode, R, nyg):
    """Wide-halo slabs under a tripole north boundary (ns 3: fold through U points, 4: through T points), R ranks = R
    contexts of this process: the rank with the top slab folds u, v after every subcycle (its overlap rows come with the
    refresh like everybody's); it runs one launch per subcycle or, "sweep", K subcycles per sweep with the band of top
    rows beside it (K = 3, or 4 which does not divide the refresh interval), while the ranks below keep their pairs /
    plain sweeps.  slabs0: no overlap, ghost rows and the fold
    after every subcycle.  peer (round 5): the whole loop in ONE launch per rank -- the rank with the top slab runs the
    cross-rank loop WITH the fold inside (its top-row tiles exchange their raw velocities among themselves, as on one rank),
    the others the plain cross-rank loop.  Against the one-block domain through one launch per subcycle (pinned to the compiled reference
    on such a grid), bit for bit on every owned cell; ocean and patchy ice up to the fold."""
    import threading
    nxg = 96
    dom1 = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=31 + ns, land_rows=0)
    grid1 = synth.block_fields(gg, dom1, ew_cyclic=True, north_ocean=True)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    s1, _ = _evp_with(ctx, grid1, s1, NDTE, False, resident=0, skew=0, skew_fold=0)
    assert np.abs(s1["uvel"][0, -3:]).max() > 1e-4
    peer_w = int(mode[6:]) if mode.startswith("peer-W") else 0
    if peer_w:
        mode = "peer"
    H = 0 if mode == "peer" else int(mode[5])
    _LINK[0] += 1
    link = _LINK[0]
    bar = threading.Barrier(R)
    out, errs, exports = [None] * R, [], [None] * R

    def rank_fn(r):
        try:
            c = lib.Context(device=0); c.sync()
            bar.wait(timeout=120)        # (every rank's main stream before anybody's copy streams: tests/ranks_case.py)
            if mode == "peer":
                dom = c.domain_create(nxg, nyg, nxg, nyg // R, ew=1, ns=ns, rank=r, npx=1, npy=R)
            else:
                dom = c.domain_create_slabs(nxg, nyg, R, ew=1, ns=ns, rank=r, nranks=R, overlap=H)
            assert dom["nblocks"] == 1 and dom["nsend"] >= 1
            c.comm_init_local(link, r, R)
            grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True)
            s = synth.evp_state(grid, dom, seed=31, cover="patchy")
            c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
            if mode == "peer":
                c.evp_set_option("resident_peer_share", R)
                if peer_w:
                    c.evp_set_option("resident_waves", peer_w)
                exports[r] = c.evp_peer_export()
                bar.wait(timeout=120)
                nbrs = c.evp_peer_ranks()
                assert nbrs == [x for x in (r - 1, r + 1) if 0 <= x < R], nbrs      # (nobody is the top slab's northern neighbour)
                for nr in nbrs:
                    c.evp_peer_connect_rank(nr, exports[nr])
                assert c.evp_get_info("resident_peer") == 1
                bar.wait(timeout=120)
            else:
                c.evp_set_option("resident", 0)
            if mode == "peer":
                pass
            elif "sweep" in mode:      # "sweep4": K = 4 does not divide the 6 subcycles between refreshes (4 + 1 + 1 / 4 + 2)
                c.evp_set_option("skew_min_cells", 0); c.evp_set_option("skew_levels", int(mode[-1]) if mode[-1].isdigit() else 3)
                assert c.evp_get_info("skew_fold" if r == R - 1 else "skew") == 1, r
            else:
                c.evp_set_option("skew", 0); c.evp_set_option("skew_fold", 0)
            if mode == "peer":      # (no rank's loop starts while another rank's uploads are queued: tests/ranks_case.py)
                c.evp_upload(s); bar.wait(timeout=120)
                c.evp_step(DT); bar.wait(timeout=120)
                c.evp_download(s)
                assert c.evp_get_info("resident_peer") == 1 and c.evp_get_info("last_launches") == 1, "the cross-rank loop fell back"
            else:
                c.evp(DT, s)
            out[r] = (dom, s)
            bar.wait(timeout=120)
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))
            bar.abort()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig") + synth.SIG_NAMES:
        want = _owned(one, s1[k])
        got = np.zeros_like(want)
        for r in range(R):
            dom, s = out[r]
            part = _owned(dom, s[k])
            rows = slice(int(dom["j0"][0] + dom["own_jlo"][0] - dom["jlo"][0]),
                         int(dom["j0"][0] + dom["own_jhi"][0] - dom["jlo"][0]) + 1)
            got[rows] = part[rows]
        assert np.array_equal(got, want), (ns, mode, R, k, np.argwhere(got != want)[:5])
    # the ghost row beyond the fold, as the last halo update of the loop leaves it (a wide-halo domain hands back its
    # owned rows only)
    if H == 0:
        dom, s = out[R - 1]
        for k in ("uvel", "vvel"):
            a, b = s[k][0, -1], s1[k][0, -1]
            assert np.array_equal(a, b), (k, np.argwhere(a != b)[:8].ravel().tolist())


@pytest.mark.parametrize("ns", [3, 4])
@pytest.mark.parametrize("nb,overlap", [(2, 0), (3, 4), (2, 6)])
def test_tripole_grid_cut_into_slabs_on_one_rank(ctx, ns, nb, overlap):
    """the same with all slabs on one rank (on-rank refresh lists instead of messages; the top slab is one of several
    local blocks, so the fold addresses a block that is not the first)"""
    nxg, nyg = 96, 72
    dom1 = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=17 + ns, land_rows=0)
    grid1 = synth.block_fields(gg, dom1, ew_cyclic=True, north_ocean=True)
    s1 = synth.evp_state(grid1, dom1, seed=17, cover="patchy")
    s1, _ = _evp_with(ctx, grid1, s1, NDTE, False, resident=0, skew=0, skew_fold=0)
    c = lib.Context(); c.sync()
    dom = c.domain_create_slabs(nxg, nyg, nb, ew=1, ns=ns, overlap=overlap)
    assert dom["nblocks"] == nb
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True)
    s = synth.evp_state(grid, dom, seed=17, cover="patchy")
    c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    c.evp(DT, s)
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig") + synth.SIG_NAMES:
        got, want = _owned(dom, s[k]), _owned(one, s1[k])
        assert np.array_equal(got, want), (ns, nb, overlap, k, np.argwhere(got != want)[:5].tolist())


def _evp_with(ctx, grid, s, ndte, damping, **opts):
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
    for k, v in opts.items():
        ctx.evp_set_option(k, v)
    info = (ctx.evp_get_info("fused"), ctx.evp_get_info("fused_waves"))
    ctx.evp(DT, sg)
    return sg, info


@pytest.mark.parametrize("nxg,nyg,ew", [(96, 70, 1), (20, 33, 1), (57, 18, 1), (58, 18, 1), (59, 18, 1),
                                         (117, 9, 1), (118, 41, 1), (119, 5, 1), (200, 50, 1), (96, 70, 0),
                                         (130, 27, 2), (7, 6, 1)])
def test_two_subcycles_per_launch(ctx, orc, nxg, nyg, ew):
    """k_subcycle2 (two subcycles per launch, redundant rim, E-W ring handled inside the kernel) against
    k_subcycle (one per launch) and the checker: bit for bit.  Widths around the 59-column tile
    stride, blocks narrower than a tile (the ring wraps inside one wavefront), open / closed E-W
    edges; even and odd ndte (an odd count ends with a single-subcycle launch); damping; all
    workgroup heights; loaded vs recomputed metrics; graph replay and eager."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg)
    grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1))
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    keys = EVP_OUT_FIELDS + ("iceumask",)
    for ndte, damping in ((NDTE, False), (7, True), (2, False)):
        ref, info = _evp_with(ctx, grid, s, ndte, damping, fuse=0, resident=0)
        assert info[0] == 0
        if (ndte, damping) == (NDTE, False):
            orc.set_evp_parameters(DT, ndte, damping); orc.set_strength_parameters(1, 0, 0, 4.0)
            so = {k: v.copy() for k, v in s.items()}
            orc.evp(orc.make_domain(dom, grid), so)
            orc.set_strength_parameters()
            for k in keys:
                assert np.array_equal(ref[k], so[k]), ("unfused vs checker", k)
        for opts in (dict(fused_waves=8), dict(fused_waves=12), dict(fused_waves=13), dict(fused_waves=14),
                     dict(fused_waves=16), dict(),
                     dict(derive_metrics=0), dict(use_graph=0)):
            got, info = _evp_with(ctx, grid, s, ndte, damping, fuse=1, resident=0, **opts)
            assert info[0] == 1 and info[1] in (8, 12, 13, 14, 16)
            for k in keys:
                assert np.array_equal(got[k], ref[k]), (ndte, damping, opts, k)


@pytest.mark.parametrize("nxg,nyg,ew", [(96, 70, 1), (20, 33, 1), (53, 18, 1), (54, 18, 1), (55, 18, 1), (107, 9, 1),
                                         (109, 41, 1), (119, 5, 1), (200, 50, 1), (96, 70, 0), (130, 27, 2), (7, 6, 1),
                                         (300, 120, 1), (56, 12, 1), (57, 15, 1), (110, 12, 1), (111, 14, 1), (112, 11, 1),
                                         (167, 13, 1), (111, 14, 2), (115, 12, 1), (103, 12, 1), (95, 14, 1),
                                         (179, 10, 1), (180, 10, 1), (181, 11, 1), (182, 10, 1), (237, 10, 1), (238, 9, 1),
                                         (299, 9, 1), (360, 8, 1), (361, 8, 1), (57, 12, 1), (120, 9, 1), (400, 30, 0)])
def test_k_subcycles_per_sweep(ctx, orc, nxg, nyg, ew):
    """k_subcycle_skew (K subcycles in one sweep: a pipeline of K time levels, one wavefront each, two rows apart,
    rows handed from level to level through LDS) against one launch per subcycle and the checker: bit for bit.
    K = 4 in both workgroup shapes (four wavefronts, one per level: the default; or twelve, three per level side by side:
    180 columns per strip, widths 179 .. 182, 360, 361, and 57, 119, 237, 299 where the seam would fall on the columns two
    wavefronts share).
    Every K; widths around the strip strides (64 - 2K columns, the strip at the ring's seam one less; widths 55, 111, 167
    (K = 4), 119 (K = 2), 115 (K = 3), 53, 107 (K = 5), 103 (K = 6), 95 (K = 8) are the ones where ihi would fall on a strip's
    last owned lane and the layout shifts by one), blocks narrower than a strip (the ring wraps inside
    one wavefront), open / closed E-W edges; row segments of 1 .. many rows (interior segment ends: K rim rows) and
    the automatic choice, segments of unequal length (longer ones for the workgroups dispatched first) and the rotation
    of issue priorities; subcycle counts that are no multiple of K (the rest runs as pairs / single launches);
    damping; graph replay and eager."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg)
    grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1))
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    keys = EVP_OUT_FIELDS + ("iceumask",)
    big = nxg * nyg > 20000
    for ndte, damping in (((NDTE, False),) if big else ((NDTE, False), (7, True), (13, False))):
        ref, _ = _evp_with(ctx, grid, s, ndte, damping, fuse=0, resident=0, skew=0)
        if (ndte, damping) == (NDTE, False):
            orc.set_evp_parameters(DT, ndte, damping); orc.set_strength_parameters(1, 0, 0, 4.0)
            so = {k: v.copy() for k, v in s.items()}
            orc.evp(orc.make_domain(dom, grid), so)
            orc.set_strength_parameters()
            for k in keys:
                assert np.array_equal(ref[k], so[k]), ("one launch per subcycle vs checker", k)
        # (K, rows per workgroup (0: automatic), graph, unequal segments in %, priority rotation)
        # (K = 5, 6, 8 and three wavefronts per level were measured slower and live in -DCICE4_AMD_EXPERIMENTS builds only)
        ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
        experiments = ctx.evp_get_info("experiments") == 1
        for K, seg, graph, pct, prio in ((4, 0, 1, 0, 0), (2, 0, 1, 30, 1), (3, 5, 1, 0, 1), (4, 1, 1, 0, 0), (4, 7, 0, 0, 1),
                                         (5, 3, 1, 0, 0), (6, 11, 1, 0, 0), (8, 4, 1, 0, 1), (8, 0, 1, 10, 0),
                                         (4, 0, 1, 25, 1), (3, 0, 1, 60, 1)):
            if K > 4 and not experiments:
                continue
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            for key, v in (("resident", 0), ("skew", 1), ("skew_min_cells", 0), ("skew_levels", K),
                           ("skew_seg_rows", seg), ("use_graph", graph), ("skew_gen_pct", pct), ("skew_prio", prio)):
                ctx.evp_set_option(key, v)
            assert ctx.evp_get_info("skew") == 1 and ctx.evp_get_info("skew_levels") == K
            # K = 4 on a cyclic one-block grid: the state lives in the sweep's pair layout between the first and the last
            # sweep of evp(dt) (16-byte loads and stores; the last sweep stores planes; a tail of pairs / singles converts back)
            assert ctx.evp_get_info("skew_pairs") == (1 if K == 4 and ew == 1 else 0)
            assert ctx.evp_get_info("skew_subs") == 1
            ctx.evp(DT, sg)
            for k in keys:
                assert np.array_equal(sg[k], ref[k]), (ndte, damping, K, seg, graph, pct, prio, k)
            if K == 4 and experiments:       # ... and as ONE 12-wavefront workgroup per CU: three wavefronts per level, their strips side by side
                s3w = {k: v.copy() for k, v in s.items()}      # (option "skew_subs": correct, measured slower, off by default)
                ctx.evp_set_option("skew_subs", 3)
                assert ctx.evp_get_info("skew_subs") == 3
                ctx.evp(DT, s3w)
                ctx.evp_set_option("skew_subs", 1)
                for k in keys:
                    assert np.array_equal(s3w[k], ref[k]), ("three wavefronts per level", ndte, damping, seg, graph, pct, prio, k)
            if K == 4 and ew == 1 and seg == 0 and pct == 0:     # the plane layout all the way: same bits
                sp = {k: v.copy() for k, v in s.items()}
                ctx.evp_set_option("skew_pairs", 0)
                assert ctx.evp_get_info("skew_pairs") == 0
                ctx.evp(DT, sp)
                ctx.evp_set_option("skew_pairs", 1)
                for k in keys:
                    assert np.array_equal(sp[k], ref[k]), ("planes", ndte, damping, k)
                # a loop cut into ranges: sweeps, a single subcycle in between (planes), sweeps again
                if ndte == NDTE:
                    b = {k: v.copy() for k, v in s.items()}
                    ctx.evp_upload(b); ctx.evp_prepare(DT)
                    ctx.evp_subcycles(1, 8); ctx.evp_subcycles(9, 1); ctx.evp_subcycles(10, 3); ctx.evp_subcycles(13, NDTE - 12)
                    ctx.evp_finish(); ctx.evp_download(b)
                    for k in EVP_OUT_FIELDS:
                        assert np.array_equal(b[k], ref[k]), ("ranges in the pair layout", k)


@pytest.mark.parametrize("nxg,nyg,ew,ns", [(96, 70, 1, 0), (20, 33, 1, 0), (62, 18, 1, 0), (63, 18, 1, 0), (64, 18, 1, 0),
                                            (125, 9, 1, 0), (126, 41, 1, 0), (127, 5, 1, 0), (200, 50, 1, 0),
                                            (96, 70, 0, 0), (130, 27, 2, 2), (7, 6, 1, 0), (96, 40, 1, 1), (320, 384, 1, 0),
                                            (250, 200, 1, 0)])
def test_whole_loop_in_one_launch(ctx, orc, nxg, nyg, ew, ns):
    """k_evp_resident (all subcycles in ONE launch: stresses, metrics and forcing stay in registers / LDS, tile-edge
    velocities travel through agent-scope stores, progress words and agent-scope loads) against one launch per
    subcycle and the checker: bit for bit.  Widths around the 63-column tile stride, blocks narrower than a tile, every
    boundary type incl. cyclic N-S (ghost rows mirrored by far tiles), every workgroup height, damping, 1- and
    2-subcycle loops (2 = one hand-off), odd counts, and a loop cut into ranges (the stepwise API); gx1 size (768
    four-wavefront tiles: three on every CU) and 250 x 200 (268 tiles: one or two per CU)."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=ns)
    # (a cyclic north-south boundary only matters if nothing closes the domain there: ocean and ice across the seam)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg, land_rows=0 if ns == 1 else 2)
    grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1), ns_cyclic=(ns == 1))
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    if ns == 1:
        assert grid["umask"][0, -2].sum() > 10 and grid["umask"][0, 1].sum() > 10
    keys = EVP_OUT_FIELDS + ("iceumask",)
    big = nxg * nyg >= 50000                 # more 4-wavefront tiles than CUs: the dense shape
    for ndte, damping in (((NDTE, False),) if big else ((NDTE, False), (7, True), (2, False), (3, False))):
        ref, _ = _evp_with(ctx, grid, s, ndte, damping, fuse=0, resident=0)
        if (ndte, damping) == (NDTE, False):
            orc.set_evp_parameters(DT, ndte, damping); orc.set_strength_parameters(1, 0, 0, 4.0)
            so = {k: v.copy() for k, v in s.items()}
            orc.evp(orc.make_domain(dom, grid), so)
            orc.set_strength_parameters()
            for k in keys:
                assert np.array_equal(ref[k], so[k]), ("one launch per subcycle vs checker", k)
        # (W, dense, granules): at gx1 size the default is "dense" -- three 4-wavefront workgroups on every CU; the edge
        # velocities travel as data-tagged granules (round 5, the default) or behind progress words (granules = 0)
        for W, dense, gran in (((0, 1, 1), (0, 1, 0), (4, 1, 1), (0, 0, 1), (8 if nxg == 250 else 11, 0, 1), (12, 0, 1), (12, 0, 0)) if big else
                               ((0, 1, 1), (0, 1, 0), (4, 1, 1), (6, 1, 1), (8, 1, 0), (8, 1, 1), (11, 1, 1), (12, 1, 1))):
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_waves", W)
            ctx.evp_set_option("resident_dense", dense); ctx.evp_set_option("resident_granules", gran)
            assert ctx.evp_get_info("resident") == 1, (W, dense, "grid should fit")
            # the granule loop runs with ONE workgroup per CU (its automatic shape where one fits); three workgroups per CU keep
            # the progress words
            is_dense = 1 if big and dense and (W == 4 or not gran) else 0
            assert ctx.evp_get_info("resident_granules") == (1 if gran and not is_dense else 0)
            assert ctx.evp_get_info("resident_dense") == is_dense
            if big and W == 0:
                assert ctx.evp_get_info("resident_waves") == (4 if is_dense else (6 if nxg == 250 else 11))
            ctx.evp(DT, sg)
            assert ctx.evp_get_info("resident") == 1 and ctx.evp_get_info("resident_dense") == is_dense, \
                "the resident loop timed out and fell back"
            for k in keys:
                assert np.array_equal(sg[k], ref[k]), (ndte, damping, W, dense, gran, k, np.argwhere(sg[k] != ref[k])[:6].tolist())
    # a loop cut into ranges: 1..5 (one launch), 6 (single subcycle: the ordinary kernel), 7..NDTE
    b = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 2)
    ctx.evp_upload(b); ctx.evp_prepare(DT)
    ctx.evp_subcycles(1, 5); ctx.evp_subcycles(6, 1); ctx.evp_subcycles(7, NDTE - 6)
    ctx.evp_finish(); ctx.evp_download(b)
    ref, _ = _evp_with(ctx, grid, s, NDTE, False, fuse=0, resident=0)
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(b[k], ref[k]), ("ranges", k)


@pytest.mark.parametrize("nxg,nyg,bsx,bsy,ew,ns", [(96, 70, 48, 35, 1, 0), (100, 116, 50, 58, 1, 0), (26, 22, 12, 10, 1, 0),
                                                    (100, 116, 10, 10, 1, 0), (130, 54, 65, 18, 1, 1), (96, 70, 32, 70, 0, 0),
                                                    (96, 72, 96, 18, 1, 2), (320, 384, 160, 192, 1, 0), (200, 120, 67, 41, 2, 0)])
def test_whole_loop_in_one_launch_on_several_blocks(ctx, orc, nxg, nyg, bsx, bsy, ew, ns):
    """The one-launch loop on a ONE-RANK DOMAIN OF SEVERAL BLOCKS (source/ice_blocks.F90:133-330; comp_ice:34-46 gives the
    serial build 5 x 5-cell blocks, COSIMA's configurations full-height slabs): tiles are numbered block by block, a ghost
    cell is produced by the tile that owns its source cell in the NEIGHBOURING block, forwarded like the east-west
    images.  Against one launch per subcycle + the on-rank halo copies (the path such domains ran before), and that
    against the checker: bit for bit, ghost cells included.  2 x 2 blocks, the gx3 2 x 2 decomposition, padded 3 x 3
    blocks (the last block row / column smaller: workgroups without a cell leave at once), 10 x 12 tiny blocks,
    cyclic north-south, full-height and full-width slabs, gx1 size in 2 x 2 blocks, blocks wider than a tile."""
    dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=ew, ns=ns)
    assert dom["nblocks"] > 1
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg, land_rows=0 if ns == 1 else 2)
    grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1), ns_cyclic=(ns == 1))
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    keys = EVP_OUT_FIELDS + ("iceumask",)
    big = nxg * nyg >= 50000
    for ndte, damping in (((NDTE, False),) if big else ((NDTE, False), (7, True), (2, False))):
        ref, _ = _evp_with(ctx, grid, s, ndte, damping, fuse=0, resident=0)
        assert np.abs(ref["uvel"]).max() > 0.01
        if (ndte, damping) == (NDTE, False):
            orc.set_evp_parameters(DT, ndte, damping); orc.set_strength_parameters(1, 0, 0, 4.0)
            so = {k: v.copy() for k, v in s.items()}
            orc.evp(orc.make_domain(dom, grid), so)
            orc.set_strength_parameters()
            for k in keys:
                assert np.array_equal(ref[k], so[k]), ("one launch per subcycle vs checker", k)
        tried = 0
        for W, dense in ((0, 1), (4, 1), (6, 0), (8, 0), (11, 0)):
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_waves", W)
            ctx.evp_set_option("resident_dense", dense)
            if not ctx.evp_get_info("resident"):
                assert W != 0, "some shape has to fit this domain"
                continue                      # this height does not give every tile a CU
            ctx.evp(DT, sg)
            assert ctx.evp_get_info("resident") == 1, ("the resident loop timed out and fell back", W, dense)
            for k in keys:
                assert np.array_equal(sg[k], ref[k]), (ndte, damping, W, dense, k, np.argwhere(sg[k] != ref[k])[:6].tolist())
            tried += 1
        assert tried >= 2
    # switched off, such a domain runs one launch per subcycle as before
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident_blocks", 0)
    assert ctx.evp_get_info("resident") == 0


@pytest.mark.parametrize("nxg,nyg", [(96, 70), (320, 384), (130, 27), (64, 40), (20, 33)])
@pytest.mark.parametrize("ns", [3, 4], ids=["tripole", "tripoleT"])
def test_whole_loop_in_one_launch_across_the_tripole_fold(ctx, nxg, nyg, ns):
    """A tripole north boundary inside the one-launch loop: the tiles of the top row hand each other the raw velocities of
    the subcycle, form the symmetric averages / mirror images of the degenerate row themselves, and forward the values of
    the ghost row across the pole -- against one launch per subcycle followed by the halo update with its fold (which
    tests/test_boundary.py pins to the reference's ice_HaloUpdate and evp): bit for bit.  Both folds, every workgroup
    height that fits, damping, short and long loops, patchy ice (top-row cells without ice take their partner's average)."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg, land_rows=0)   # ocean up to the fold
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True)
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    keys = EVP_OUT_FIELDS + ("iceumask",)
    # the fold has to matter in this case: the same loop with an open north boundary gives other velocities in the top rows
    domo = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    opn, _ = _evp_with(ctx, grid, s, NDTE, False, resident=0)
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    for ndte, damping in ((NDTE, False), (7, True), (2, False)):
        ref, _ = _evp_with(ctx, grid, s, ndte, damping, resident=0, resident_fold=0)
        assert np.abs(ref["uvel"]).max() > 0.01 and np.abs(ref["uvel"][0, -3:]).max() > 1e-4
        if ndte == NDTE and not damping:
            assert not np.array_equal(ref["uvel"][0, -4:-1], opn["uvel"][0, -4:-1]) and np.abs(ref["uvel"][0, -2]).max() > 1e-4
        tried = 0
        # (W, granules): round 5 -- the free-running granule loop under the fold too (the raw top row travels as granules of
        # its own, every top-row lane polls its partner across the pole); 0: the progress words of rounds 3-4
        for W, gran in ((0, 1), (0, 0), (4, 1), (4, 0), (6, 1), (8, 0), (8, 1), (11, 1), (11, 0)):
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            ctx.evp_set_option("resident_fold", 1); ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_waves", W)
            ctx.evp_set_option("resident_granules", gran)
            if not ctx.evp_get_info("resident"):
                continue                      # this height does not give every tile a CU
            if gran:
                assert ctx.evp_get_info("resident_granules") == (0 if ctx.evp_get_info("resident_dense") else 1)
            ctx.evp(DT, sg)
            assert ctx.evp_get_info("resident") == 1, ("fell back", W, gran)
            for k in keys:
                assert np.array_equal(sg[k], ref[k]), (ns, ndte, damping, W, gran, k, np.argwhere(sg[k] != ref[k])[:6].tolist())
            tried += 1
        assert tried >= 2
    ctx.evp_set_option("resident_waves", 0)
    # the same grid cut into blocks (fold after every subcycle through the halo update, ghost rows between blocks):
    # the owned cells must come out as on one block
    if nxg % 2 == 0 and nyg % 2 == 0 and nyg // 2 >= 3:
        ref, _ = _evp_with(ctx, grid, s, NDTE, False, resident=0, resident_fold=0)
        dom2 = ctx.domain_create(nxg, nyg, nxg // 2, nyg // 2, ew=1, ns=ns)
        grid2 = synth.block_fields(gg, dom2, ew_cyclic=True, north_ocean=True)
        s2 = synth.evp_state(grid2, dom2, seed=nxg, cover="patchy")
        got, _ = _evp_with(ctx, grid2, s2, NDTE, False)
        for k in ("uvel", "vvel", "stressp_1", "stress12_4", "strintx"):
            for b in range(dom2["nblocks"]):
                i0, j0 = int(dom2["i0"][b]), int(dom2["j0"][b])
                ni, nj = int(dom2["ihi"][b] - dom2["ilo"][b] + 1), int(dom2["jhi"][b] - dom2["jlo"][b] + 1)
                assert np.array_equal(got[k][b, 1:1 + nj, 1:1 + ni], ref[k][0, 1 + j0:1 + j0 + nj, 1 + i0:1 + i0 + ni]), (ns, k, b)


@pytest.mark.parametrize("nxg,nyg", [(300, 120), (96, 70), (130, 48), (20, 64)])
@pytest.mark.parametrize("ns", [3, 4], ids=["tripole", "tripoleT"])
def test_sweeps_on_a_tripole_grid(ctx, nxg, nyg, ns):
    """K subcycles per sweep where a fold couples the two halves of the top row after every subcycle: the sweep runs as on
    an open boundary and a band of the top 2K + 1 rows runs the K subcycles one at a time beside it (k_subcycle + the halo
    update with the fold, on buffers of its own); the band's rows above jhi - K replace the sweep's.  Against one launch
    per subcycle (pinned to the reference on such a grid): bit for bit, every K, subcycle counts that are no multiple of
    K, damping, graph and eager, ocean and patchy ice up to the fold."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg, land_rows=0)
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True)
    s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
    keys = EVP_OUT_FIELDS + ("iceumask",)
    for ndte, damping in ((NDTE, False), (13, True), (7, False)):
        ref, _ = _evp_with(ctx, grid, s, ndte, damping, resident=0, skew=0, skew_fold=0)
        assert np.abs(ref["uvel"][0, -3:]).max() > 1e-4
        for K, graph in ((4, 1), (3, 1), (2, 0), (6, 1)):
            if nyg < 4 * K + 6 or (K > 4 and not ctx.evp_get_info("experiments")):   # (K = 5, 6, 8: -DCICE4_AMD_EXPERIMENTS builds)
                continue
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            for key, v in (("resident", 0), ("skew", 1), ("skew_fold", 1), ("skew_min_cells", 0), ("skew_levels", K),
                           ("use_graph", graph)):
                ctx.evp_set_option(key, v)
            assert ctx.evp_get_info("skew_fold") == 1, (K, "the sweep path should apply")
            ctx.evp(DT, sg)
            for k in keys:
                assert np.array_equal(sg[k], ref[k]), (ns, ndte, damping, K, graph, k, np.argwhere(sg[k] != ref[k])[:6].tolist())


@pytest.mark.parametrize("cover,expect", [("caps", 1), ("full", 0)])
def test_one_launch_loop_chooses_its_tile_map_by_the_ice_cover(ctx, cover, expect, ns=0):
    """Which tile a workgroup of the one-launch loop takes is free (everything between tiles goes by tile number): under an
    ice cover in latitude bands the loop deals the tiles so that a CU holds one tile with ice and two without (k_res_choose_map,
    once per evp(dt), on the device); a fully covered grid keeps the tiles of an XCD together.  Both maps and the choice:
    the bits of one launch per subcycle, gx1 size (768 tiles: the dense shape)."""
    nxg, nyg = 320, 384
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.04, seed=5, land_rows=0 if ns else 1)
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=bool(ns))
    s = synth.evp_state(grid, dom, seed=9, cover=cover)
    keys = EVP_OUT_FIELDS + ("iceumask",)
    ref, _ = _evp_with(ctx, grid, s, 24, False, resident=0, skew=0, skew_fold=0)
    for m in (-1, 0, 1):
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp_init(grid, ndte=24, krdg_partic=0, krdg_redist=0)
        ctx.evp_set_option("resident", 1); ctx.evp_set_option("resident_granules", 0)    # (three workgroups per CU: progress words)
        ctx.evp_set_option("resident_map", m)
        assert ctx.evp_get_info("resident_dense") == 1
        ctx.evp(DT, sg)
        assert ctx.evp_get_info("last_launches") == 1, "the one-launch loop should have run"
        assert ctx.evp_get_info("resident_map") == (expect if m < 0 else m), (cover, m)
        for k in keys:
            assert np.array_equal(sg[k], ref[k]), (cover, m, k, np.argwhere(sg[k] != ref[k])[:6].tolist())


@pytest.mark.parametrize("cover,dense_after", [("caps", 1), ("full", 0), ("patchy", 0)])
def test_one_launch_loop_chooses_its_shape_by_the_ice_cover(ctx, cover, dense_after):
    """gx1 size: the free-running granule loop (one 11-wavefront workgroup per CU) is the shape of a covered grid; where the
    last step's ice cover left most tiles empty (polar caps) the next evp(dt) takes three barrier-coupled workgroups per CU
    instead, whose tile map puts one tile with ice on every CU.  Both shapes: the bits of one launch per subcycle."""
    nxg, nyg = 320, 384
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.04, seed=5, land_rows=1), dom, ew_cyclic=True)
    s = synth.evp_state(grid, dom, seed=9, cover=cover)
    keys = EVP_OUT_FIELDS + ("iceumask",)
    ref, _ = _evp_with(ctx, grid, s, 24, False, resident=0, skew=0, skew_fold=0)
    ctx.evp_init(grid, ndte=24, krdg_partic=0, krdg_redist=0)
    assert (ctx.evp_get_info("resident_granules"), ctx.evp_get_info("resident_dense"), ctx.evp_get_info("resident_waves")) == (1, 0, 11)
    for call in range(3):
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp(DT, sg)
        assert ctx.evp_get_info("last_launches") == 1, "the one-launch loop should have run"
        assert ctx.evp_get_info("resident_dense") == dense_after, (cover, call)       # (what the NEXT call will take)
        assert ctx.evp_get_info("resident_granules") == 1 - dense_after
        for k in keys:
            assert np.array_equal(sg[k], ref[k]), (cover, call, k)


@pytest.mark.parametrize("nxg,nyg,ns,cover", [(130, 400, 0, "caps"), (70, 260, 0, "patchy"), (96, 300, 3, "caps")])
def test_sweep_segments_follow_the_measured_cost(ctx, nxg, nyg, ns, cover):
    """The sweep's row segments are re-cut after every measured sweep of a tuning phase (Evp::balance_after_sweep: start / end
    ticks per workgroup -> cost per row -> boundaries): the table stays a partition of every strip's rows, segments over
    open water grow at the expense of those over ice, and the results are the bits of one launch per subcycle in the loop
    that tunes (eager), in the graph replay after it and in a later tuning phase -- on a plain and on a tripole grid."""
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=nxg + nyg, land_rows=0 if ns else 1)
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=bool(ns))
    s = synth.evp_state(grid, dom, seed=nxg, cover=cover)
    keys = EVP_OUT_FIELDS + ("iceumask",)
    ndte, K = 24, 4
    ref, _ = _evp_with(ctx, grid, s, ndte, False, resident=0, skew=0, skew_fold=0)
    ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
    for key, v in (("resident", 0), ("skew", 1), ("skew_fold", 1), ("skew_min_cells", 0), ("skew_levels", K), ("skew_balance", 1),
                   ("skew_balance_every", 2)):
        ctx.evp_set_option(key, v)
    assert ctx.evp_get_info("skew_balance") == 1
    tables = []
    for call in range(5):      # 1: tunes (eager); 2: graph; 3: tunes again (every 2 loops); 4, 5: graph / tunes
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp(DT, sg)
        for k in keys:
            assert np.array_equal(sg[k], ref[k]), (call, k, np.argwhere(sg[k] != ref[k])[:6].tolist())
        t = ctx.evp_debug("skew_rows").reshape(-1, 3)      # per tile: strip, first / last row
        strips = ctx.evp_get_info("skew_strips")
        rows = nyg
        for sx in range(strips):   # the tiles of a strip: consecutive, complete
            ts = t[t[:, 0] == sx]
            ts = ts[np.argsort(ts[:, 1], kind="stable")]
            assert len(ts) >= 1 and ts[0, 1] == 0 and ts[:, 2].max() == rows - 1, (call, sx, ts.tolist())
            assert np.all(ts[1:, 1] == ts[:-1, 2] + 1), (call, sx, ts.tolist())
        tables.append(t.copy())
    assert ctx.evp_get_info("skew_balanced") >= ndte // K, "the first loop's sweeps should have been measured"
    n0 = tables[0][:, 2] - tables[0][:, 1] + 1
    assert n0.max() - n0.min() >= 2, ("segments should have moved away from equal lengths", n0[:12].tolist())
    assert len(tables[-1]) >= len(tables[0]) >= strips


def test_sweep_segments_cut_for_one_ice_cover_serve_another(ctx):
    """The rows that hold ice are found anew in every evp(dt); the segment table is re-cut only now and then.  A table cut
    for polar caps, then ice in blobs, then a full cover, and back -- one context, no re-initialisation, a tuning phase every
    third loop: every call gives the bits of one launch per subcycle on that state."""
    nxg, nyg, ndte, K = 130, 400, 24, 4
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=nxg + nyg)
    grid = synth.block_fields(gg, dom, ew_cyclic=True)
    keys = EVP_OUT_FIELDS + ("iceumask",)
    states, refs = {}, {}
    for cover in ("caps", "patchy", "full"):
        states[cover] = synth.evp_state(grid, dom, seed=nxg, cover=cover)
        refs[cover], _ = _evp_with(ctx, grid, states[cover], ndte, False, resident=0, skew=0)
    ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
    for key, v in (("resident", 0), ("skew", 1), ("skew_min_cells", 0), ("skew_levels", K), ("skew_balance", 1), ("skew_balance_every", 3)):
        ctx.evp_set_option(key, v)
    for call, cover in enumerate(("caps", "caps", "patchy", "full", "caps", "patchy", "patchy", "full", "caps")):
        sg = {k: v.copy() for k, v in states[cover].items()}
        ctx.evp(DT, sg)
        for k in keys:
            assert np.array_equal(sg[k], refs[cover][k]), (call, cover, k, np.argwhere(sg[k] != refs[cover][k])[:6].tolist())
    assert ctx.evp_get_info("skew_balanced") > ndte // K


@pytest.mark.parametrize("ns", ["tripole", "tripoleT"])
def test_tripole_fold_inside_the_loop_against_the_compiled_reference(ns):
    """the same against `call evp(dt)` of the reference itself on a one-block 100 x 116 domain (own process)"""
    import os, subprocess, sys
    from oracle import refapi
    if not refapi.available("gx3"):
        pytest.skip("oracle/_ref/libcice_ref_gx3.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tripole_evp_case.py"), ns], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "TRIPOLE-EVP-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("gran", [1, 0], ids=["granules", "progress-words"])
def test_resident_loop_is_repeatable(ctx, gran):
    """300 one-launch loops (36,000 subcycles, 768 tiles, ~90 exchanged velocities per tile and subcycle) from the same
    state: every call must return the bits of the first one, which are those of the launch-per-pair loop.  A hand-off
    that once in a while let a stale velocity through would show here: one wrong ulp grows to 1e-2 within a step."""
    nxg, nyg = 320, 384
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=8)
    grid = synth.block_fields(gg, dom, ew_cyclic=True)
    s = synth.evp_state(grid, dom, seed=8, cover="patchy")
    ref, _ = _evp_with(ctx, grid, s, NDTE, False, fuse=1, resident=0)
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_granules", gran)
    assert ctx.evp_get_info("resident_dense") == 1 - gran and ctx.evp_get_info("resident_granules") == gran
    ctx.evp_upload({k: v.copy() for k, v in s.items()})
    out = {k: np.empty_like(v) for k, v in s.items()}
    for rep in range(300):
        if rep % 100 == 0:                      # the whole call now and then, the loop alone otherwise
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp(DT, sg)
            for k in EVP_OUT_FIELDS:
                assert np.array_equal(sg[k], ref[k]), (rep, k)
            ctx.evp_upload({k: v.copy() for k, v in s.items()})
        ctx.evp_prepare(DT)
        ctx.evp_subcycles(1, NDTE)
        ctx.evp_finish()
        ctx.evp_download(out)
        for k in ("uvel", "vvel", "stressp_1", "stress12_3"):
            assert np.array_equal(out[k], ref[k]), (rep, k)
        ctx.evp_upload({k: v.copy() for k, v in s.items()})
    assert ctx.evp_get_info("resident") == 1


@pytest.mark.parametrize("gran", [1, 0], ids=["granules", "progress-words"])
def test_resident_loop_gives_up_cleanly(ctx, orc, gran):
    """A tile of the one-launch loop that does not hear from a neighbour in time raises the abort word, every
    workgroup leaves, and the caller's state is as it was: the range is then run by the launch-per-pair loop.  With
    resident_spin_us = 0 every wait fails: the dense shape (gx1 size, progress words) falls back to one workgroup per CU
    for the next call, that one falls back for good -- and all three calls give the bits of the ordinary loop.  The granule
    loop starts with one workgroup per CU and falls back for good at once."""
    nxg, nyg = 320, 384
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=5)
    grid = synth.block_fields(gg, dom, ew_cyclic=True)
    s = synth.evp_state(grid, dom, seed=3, cover="patchy")
    ref, _ = _evp_with(ctx, grid, s, 12, False, fuse=1, resident=0)
    ctx.evp_init(grid, ndte=12, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_spin_us", 0); ctx.evp_set_option("resident_granules", gran)
    want = [(1, 0, 11), (0, 0, None), (0, 0, None)] if gran else [(1, 1, 4), (1, 0, 11), (0, 0, None)]   # (resident, dense, waves) before each call
    for call in range(3):
        assert (ctx.evp_get_info("resident"), ctx.evp_get_info("resident_dense")) == want[call][:2], call
        if want[call][2]:
            assert ctx.evp_get_info("resident_waves") == want[call][2]
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp(DT, sg)
        for k in EVP_OUT_FIELDS + ("iceumask",):
            assert np.array_equal(sg[k], ref[k]), (call, k)
    ctx.evp_set_option("resident_spin_us", 200000); ctx.evp_set_option("resident", 2)    # forgiven: the first shape again, and it works
    assert (ctx.evp_get_info("resident"), ctx.evp_get_info("resident_dense")) == (1, 1 - gran)
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp(DT, sg)
    assert ctx.evp_get_info("resident_dense") == 1 - gran
    for k in EVP_OUT_FIELDS + ("iceumask",):
        assert np.array_equal(sg[k], ref[k]), ("after", k)


def test_resident_loop_is_tried_again_later(ctx):
    """a time-out may be somebody else's doing (a co-tenant holding CUs): after resident_retry_steps calls of evp(dt) the
    shape that timed out is tried again"""
    nxg, nyg = 320, 384
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=5), dom, ew_cyclic=True)
    s = synth.evp_state(grid, dom, seed=3, cover="patchy")
    ref, _ = _evp_with(ctx, grid, s, 12, False, fuse=1, resident=0)
    ctx.evp_init(grid, ndte=12, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_retry_steps", 2); ctx.evp_set_option("resident_spin_us", 0)
    ctx.evp_set_option("resident_granules", 0)     # (the dense shape and its fall-back to one workgroup per CU: progress words)
    seen = []
    for call in range(4):
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp(DT, sg)              # call 0: the dense shape times out
        ctx.evp_set_option("resident_spin_us", 200000)
        seen.append((ctx.evp_get_info("resident"), ctx.evp_get_info("resident_dense")))
        for k in EVP_OUT_FIELDS + ("iceumask",):
            assert np.array_equal(sg[k], ref[k]), (call, k)
    # after call 0 and 1: one workgroup per CU; call 2 is the second call after the time-out: dense again, and it stays
    assert seen == [(1, 0), (1, 0), (1, 1), (1, 1)], seen
    ctx.evp_set_option("resident_retry_steps", 64)


def test_fused_pairs_not_used_where_ghost_rows_change(ctx):
    """Several blocks per rank without overlap rows, or a cyclic N-S edge: ghost rows are refreshed
    after every subcycle, so the one-subcycle kernel must run."""
    for args, kw in (((96, 70, 48, 35), dict(ew=1, ns=0)), ((96, 70, 96, 35), dict(ew=1, ns=0)),
                     ((96, 70, 96, 70), dict(ew=1, ns=1))):
        dom = ctx.domain_create(*args, **kw)
        gg = synth.global_grid(args[0], args[1], perturb=0.1, seed=3)
        grid = synth.block_fields(gg, dom)
        ctx.evp_init(grid, ndte=4)
        assert ctx.evp_get_info("fused") == 0
    dom = ctx.domain_create_slabs(96, 72, 4, ew=1, ns=0, overlap=3)   # odd overlap
    ctx.evp_init(synth.block_fields(synth.global_grid(96, 72, seed=3), dom), ndte=4)
    assert ctx.evp_get_info("fused") == 0
    dom = ctx.domain_create_slabs(96, 72, 4, ew=1, ns=0, overlap=4)
    ctx.evp_init(synth.block_fields(synth.global_grid(96, 72, seed=3), dom), ndte=4)
    assert ctx.evp_get_info("fused") == 1


def test_pairing_fuzz():
    """scripts/fuzz_pairing.py: 30 random grids / boundary types (E-W and N-S) / subcycle counts / workgroup
    heights and shapes, with the never-written ghost cells made different from the cells they would mirror: two
    subcycles per launch == the whole loop in one launch == one subcycle per launch, bit for bit."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_pairing.py"), "30", "11"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FUZZ-OK 30" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_page_locked_host_arrays_give_the_same_result(orc):
    """cice_evp_pin_fields / cice_host_register only change how the host arrays travel."""
    c = lib.Context(); c.sync()
    dom = c.domain_create(64, 40, 64, 40, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(64, 40, perturb=0.1, land_frac=0.05, seed=4), dom)
    s = synth.evp_state(grid, dom, seed=4, cover="patchy")
    c.evp_init(grid, ndte=8)
    a = {k: v.copy() for k, v in s.items()}
    c.evp(DT, a)
    b = {k: v.copy() for k, v in s.items()}
    c.evp_pin_fields(b)
    extra = np.zeros(1000)
    c.host_register(extra); c.host_register(extra)      # registering twice is harmless
    c.evp(DT, b)
    c.evp(DT, b)                                        # second call on the same (pinned) arrays: state carried
    c.evp(DT, a)
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(a[k], b[k]), k
    c.host_unregister_all()                             # before the arrays go away
    del c


FLUX7 = ("fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty")


@pytest.mark.parametrize("keep,lazy,pinned,blocks", [(1, 0, False, 1), (2, 0, True, 1), (2, 1, True, 1), (1, 1, False, 1), (2, 1, True, 4)])
def test_evp_over_pcie_leaves_on_the_device_what_the_caller_does_not_touch(keep, lazy, pinned, blocks):
    """cice_evp with the caller's two statements (include/cice4_amd.h: "keep_state", "lazy_stresses") against plain cice_evp,
    five steps of a driver that does what the reference's does between two evp calls: new forcing and state every step,
    the dynamic history fields zeroed (init_history_dyn, /root/reference/source/ice_flux.F90:585-602).  Bit for bit the same
    fields on the host after every step.  That the planes really stay where they are: the kept host arrays are filled with
    NaN behind the library's back (a caller that breaks its statement) and nothing changes."""
    c = lib.Context(); c.sync()
    dom = c.domain_create(100, 116, 100 if blocks == 1 else 50, 116 if blocks == 1 else 58, ew=1, ns=0)   # (blocks = 4: 2 x 2 blocks on the one rank)
    assert dom["nblocks"] == blocks
    grid = synth.block_fields(synth.global_grid(100, 116, perturb=0.1, land_frac=0.05, seed=5), dom)
    s0 = synth.evp_state(grid, dom, seed=5, cover="patchy", moving=False)
    rng = np.random.default_rng(11)
    forcing = [{k: s0[k] * (1.0 + 0.2 * rng.standard_normal()) for k in ("strairxT", "strairyT", "uocn", "vocn")} for _ in range(5)]
    cover = [np.clip(s0["aicen"] * (1.0 + 0.05 * step), 0.0, 0.19) for step in range(5)]

    def drive(state, step):
        """the host side of one step, before evp"""
        for k, v in forcing[step].items():
            state[k][...] = v
        state["aicen"][...] = cover[step]
        state["aice"][...] = state["aicen"].sum(axis=1) if state["aicen"].ndim == 4 else state["aicen"].sum(axis=0)
        state["aice0"][...] = 1.0 - state["aice"]
        for k in FLUX7 + ("prs_sig",):
            state[k][...] = 0.0

    c.evp_init(grid, ndte=8)
    plain = {k: v.copy() for k, v in s0.items()}
    want = []
    for step in range(5):
        drive(plain, step)
        c.evp(DT, plain)
        want.append({k: plain[k].copy() for k in EVP_OUT_FIELDS + ("iceumask",)})
    assert np.abs(want[-1]["uvel"]).max() > 1e-3 and not np.array_equal(want[1]["stressp_1"], want[4]["stressp_1"])

    c.evp_init(grid, ndte=8)
    c.evp_set_option("keep_state", keep); c.evp_set_option("lazy_stresses", lazy)
    got = {k: v.copy() for k, v in s0.items()}
    if pinned:
        c.evp_pin_fields(got)
    for step in range(5):
        drive(got, step)
        c.evp(DT, got)
        for k in EVP_OUT_FIELDS + ("iceumask",):
            if lazy and k in synth.SIG_NAMES:
                continue
            assert np.array_equal(got[k], want[step][k]), (step, k)
        if lazy:
            if step in (2, 4):          # a step that writes history / a restart file
                c.evp_download_stresses(got)
                for k in synth.SIG_NAMES:
                    assert np.array_equal(got[k], want[step][k]), (step, k)
            else:
                for k in synth.SIG_NAMES:
                    assert not np.array_equal(got[k], want[step][k]), (step, k, "was downloaded all the same")
        # the kept planes are not read again: a caller that overwrites them on the host changes nothing
        for k in ("uvel", "vvel", "iceumask") + synth.SIG_NAMES:
            got[k][...] = 0 if k == "iceumask" else np.nan
    # an explicit upload ends the statement for one call: the host arrays are the input again
    fresh = {k: v.copy() for k, v in s0.items()}
    drive(fresh, 0)
    c.evp_upload(fresh); c.evp_step(DT); c.evp_download(fresh)
    for k in EVP_OUT_FIELDS:
        assert np.array_equal(fresh[k], want[0][k]), k
    c.host_unregister_all()
    del c
