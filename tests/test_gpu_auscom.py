"""GPU parity tests of the coupled flavour of the library (libcice4_amd_auscom.so, the replacement for a reference
built -DAusCOM -Dcoupled with drivers/access-om/ice_constants.F90) against the CPU checker built and switched the same
way -- which tests/test_oracle_auscom.py pins bit for bit to the reference compiled that way.  Every kernel family of
the EVP loop (one launch per subcycle, pairs, K-level sweep, whole loop in one launch), several blocks, both
hemispheres on one grid, namelist turning angles / drag / ocean-slope tilt; the thermodynamic entries with MOM's
cp_ocn, reference salinity and a namelist chio; the Fortran drop-in modules compiled -DAusCOM inside the reference's
callers."""
import tempfile

import numpy as np
import pytest

from cice4_amd import lib, synth
from conftest import relerr, TOL_EXP, TOL_POW
from test_gpu_evp import DT, NDTE, EVP_OUT_FIELDS
from test_gpu_thermo import _cmp, CHECK
from test_oracle_auscom import NAMELISTS, two_hemispheres

pytestmark = pytest.mark.gpu
KEYS = EVP_OUT_FIELDS + ("iceumask",)


@pytest.fixture(scope="module")
def ctx_aus():
    c = lib.Context(flavour="auscom")
    c.sync()
    yield c
    c.set_auscom(); c.set_chio()


def test_the_two_builds_say_which_they_are(ctx, ctx_aus):
    assert ctx.lib.cice_build_flavour() == b"standalone" and ctx_aus.lib.cice_build_flavour() == b"auscom"
    with pytest.raises(lib.CiceError, match="stand-alone build"):
        ctx.set_auscom(sinw=0.1)
    with pytest.raises(lib.CiceError, match="stand-alone build"):
        ctx.set_chio(0.004)


def _case(c, nxg, nyg, bsx, bsy, seed=3):
    dom = c.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    grid = two_hemispheres(synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=seed), dom))
    s = synth.evp_state(grid, dom, seed=seed, cover="patchy")
    rng = np.random.default_rng(seed)     # a surface slope for use_ocnslope (the stand-alone build never reads it)
    s["ss_tltx"] = rng.uniform(-2e-5, 2e-5, s["ss_tltx"].shape); s["ss_tlty"] = rng.uniform(-2e-5, 2e-5, s["ss_tlty"].shape)
    return dom, grid, s


FAMILIES = [dict(fuse=0, resident=0, skew=0), dict(fuse=1, resident=0, skew=0),
            dict(resident=0, skew=1, skew_min_cells=0, skew_levels=4), dict(resident=0, skew=1, skew_min_cells=0, skew_levels=3,
                                                                          skew_seg_rows=5),
            dict(resident=2), dict(resident=2, resident_waves=8)]


@pytest.mark.parametrize("nml", NAMELISTS[1:])
@pytest.mark.parametrize("bs", [(96, 70), (48, 35)])
def test_whole_evp_every_kernel_family(ctx_aus, orc_aus, nml, bs):
    c, orc = ctx_aus, orc_aus
    dom, grid, s = _case(c, 96, 70, *bs)
    c.set_auscom(**nml); orc.set_auscom(True, **nml)
    for ndte, damping in ((NDTE, False), (7, True)):
        orc.set_evp_parameters(DT, ndte, damping); orc.set_strength_parameters(1, 0, 0, 4.0)   # exp-free: bit for bit
        so = {k: v.copy() for k, v in s.items()}
        orc.evp(orc.make_domain(dom, grid), so)
        orc.set_strength_parameters()
        assert (so["fm"] < 0).any() and (so["fm"] > 0).any() and np.abs(so["uvel"]).max() > 0.01
        for opts in FAMILIES:
            if dom["nblocks"] > 1 and ("skew" in opts and opts["skew"] or opts.get("resident")):
                continue      # those two run on one block per rank
            sg = {k: v.copy() for k, v in s.items()}
            c.evp_init(grid, ndte=ndte, evp_damping=damping, krdg_partic=0, krdg_redist=0)
            for k, v in opts.items():
                c.evp_set_option(k, v)
            c.evp(DT, sg)
            if opts.get("skew"):
                assert c.evp_get_info("skew") == 1
            for k in KEYS:
                assert np.array_equal(sg[k], so[k]), (nml, bs, ndte, opts, k)
    # default strength (exp in ice_strength): the tolerance of the stand-alone tests
    orc.set_evp_parameters(DT, NDTE, False)
    so = {k: v.copy() for k, v in s.items()}
    orc.evp(orc.make_domain(dom, grid), so)
    sg = {k: v.copy() for k, v in s.items()}
    c.evp_init(grid, ndte=NDTE)
    c.evp(DT, sg)
    for k in EVP_OUT_FIELDS:
        assert relerr(sg[k], so[k]) <= (TOL_EXP if TOL_EXP == 0.0 else 1e-8), k


def test_a_changed_namelist_reaches_a_replayed_graph(ctx_aus, orc_aus):
    """the loop is captured once and replayed: the turning angle is read at execution time, not baked in"""
    c, orc = ctx_aus, orc_aus
    dom, grid, s = _case(c, 64, 40, 64, 40, seed=11)
    c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    c.evp_set_option("resident", 0)
    for nml in (NAMELISTS[1], NAMELISTS[2], NAMELISTS[0], NAMELISTS[1]):
        c.set_auscom(**nml); orc.set_auscom(True, **nml)
        orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
        so = {k: v.copy() for k, v in s.items()}
        orc.evp(orc.make_domain(dom, grid), so)
        orc.set_strength_parameters()
        sg = {k: v.copy() for k, v in s.items()}
        c.evp(DT, sg)
        for k in KEYS:
            assert np.array_equal(sg[k], so[k]), (nml, k)


def test_no_turning_angle_gives_the_stand_alone_dynamics(ctx, ctx_aus):
    """sinw = 0, cosw = 1, the stand-alone drag and no ocean slope: the hemisphere drops out and evp(dt) of the two
    builds agrees bit for bit (the constants that differ are thermodynamic)"""
    dom, grid, s = _case(ctx_aus, 96, 70, 96, 70, seed=5)
    ctx.domain_create(96, 70, 96, 70, ew=1, ns=0)
    ctx_aus.set_auscom()
    out = []
    for c in (ctx, ctx_aus):
        sg = {k: v.copy() for k, v in s.items()}
        c.evp_init(grid, ndte=NDTE)
        c.evp(DT, sg)
        out.append(sg)
    for k in KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("chio", [0.006, 0.004])
def test_frzmlt_bottom_lateral(ctx_aus, orc_aus, chio):
    c, orc = ctx_aus, orc_aus
    c.thermo_init(); orc.init_thermo()
    c.set_chio(chio); orc.set_chio(chio)
    ny, nx = 30, 44
    rng = np.random.default_rng(8)
    aice = np.where(rng.uniform(0, 1, (ny, nx)) < 0.8, rng.uniform(0.01, 1, (ny, nx)), 0.0)
    args = (2, nx - 1, 2, ny - 1, DT, aice, rng.uniform(-60, 20, (ny, nx)), -rng.uniform(1e6, 3e8, (20, ny, nx)),
            -rng.uniform(0, 5e7, (5, ny, nx)), np.full((ny, nx), -1.8) + rng.uniform(0, 1.5, (ny, nx)),
            np.full((ny, nx), -1.8), rng.uniform(-0.2, 0.2, (ny, nx)), rng.uniform(-0.2, 0.2, (ny, nx)))
    g = c.frzmlt_bottom_lateral(*args); w = orc.frzmlt_bottom_lateral(*args)
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])     # Tbot, fbot: no pow
    assert relerr(g[2], w[2]) <= TOL_POW
    assert (w[1] < 0).any()
    c.set_chio(); orc.set_chio()


@pytest.mark.parametrize("conduct", ["MU71", "bubbly"])
def test_thermo_vertical(ctx_aus, orc_aus, conduct):
    """cp_ocn in the enthalpy <-> temperature relation, ice_ref_salinity in the salt flux"""
    c, orc = ctx_aus, orc_aus
    so, to = orc.init_thermo(conduct=conduct); sg, tg = c.thermo_init(conduct=conduct)
    assert relerr(sg, so) < 1e-15
    for regime in ("winter", "summer", "mixed"):
        for n in (0, 2, 4):
            a, icells, ii, jj = synth.thermo_columns(37, 70, n, regime=regime, seed=11)
            ag = {k: v.copy() for k, v in a.items()}; ac = {k: v.copy() for k, v in a.items()}
            assert c.thermo_vertical(DT, icells, ii, jj, ag, yday=200.0) == \
                orc.thermo_vertical(DT, icells, ii, jj, ac, yday=200.0) == (0, 0, 0)
            _cmp(ag, ac, (regime, conduct, n))
    c.thermo_init(); orc.init_thermo()


def test_thermo_differs_from_the_stand_alone_build(ctx, ctx_aus):
    ctx.thermo_init(); ctx_aus.thermo_init()
    a, icells, ii, jj = synth.thermo_columns(37, 70, 1, regime="summer", seed=11)
    a1 = {k: v.copy() for k, v in a.items()}; a2 = {k: v.copy() for k, v in a.items()}
    ctx.thermo_vertical(DT, icells, ii, jj, a1, yday=200.0); ctx_aus.thermo_vertical(DT, icells, ii, jj, a2, yday=200.0)
    assert not np.array_equal(a1["fsaltn"], a2["fsaltn"])
    m = (a1["fsaltn"] != 0) & (a2["fsaltn"] != 0)
    assert m.sum() > 100 and abs(np.median(a2["fsaltn"][m] / a1["fsaltn"][m]) - 1.25) < 0.05   # reference salinity 4 -> 5 ppt


def test_fortran_dropin_modules_compiled_auscom_inside_reference_callers(orc_aus):
    """The reference's own compiled modules (constants of drivers/access-om, the coupler's data modules) and its capture
    wrapper, linked with OUR ice_dyn_evp.F90 and ice_therm_vertical.F90 compiled -DAusCOM and libcice4_amd_auscom.so:
    the namelist variables are set through the modules' public cosw / sinw / dragio / chio, `call evp(dt)` and
    `call frzmlt_bottom_lateral` go to the GPU, and sicemass is filled as the reference's evp fills it."""
    from __graft_entry__ import REF_CONFIGS
    from oracle import refapi
    orc = orc_aus
    cfg, kind = "gx3b4", "dropinaus"
    if not refapi.available(cfg, kind):
        pytest.skip(f"oracle/_ref/libcice_{kind}_{cfg}.so not built")
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg, kind=kind)
    nb = ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE)
    dom = lib.Context(flavour="auscom").domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    assert nb == dom["nblocks"] == mxb
    ny, nx = dom["ny"], dom["nx"]
    grid = two_hemispheres(synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05), dom))
    s = synth.evp_state(grid, dom, cover="patchy")
    rng = np.random.default_rng(2)
    s["ss_tltx"] = rng.uniform(-2e-5, 2e-5, s["ss_tltx"].shape); s["ss_tlty"] = rng.uniform(-2e-5, 2e-5, s["ss_tlty"].shape)
    for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear",
              "uarear", "tinyarea", "fcor", "HTN", "HTE"):
        ref.set(k, grid[k])
    ref.set("tmask", grid["tmask"].astype(float)); ref.set("umask", grid["umask"].astype(float))
    ref.set_strength_parameters(1, 0, 0, 4.0)
    ref.evp_gpu_setup()
    for nml in NAMELISTS[1:]:
        ref.set_auscom(chio=0.004, **nml); orc.set_auscom(True, **nml); orc.set_chio(0.004)
        for k in ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty",
                  "uvel", "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx",
                  "strinty") + synth.SIG_NAMES:
            ref.set(k, s[k])
        ref.set("iceumask", s["iceumask"].astype(float))
        ref.set("aicen", s["aicen"].reshape(-1, ny, nx)); ref.set("vicen", s["vicen"].reshape(-1, ny, nx))
        ref.evp(DT)
        orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
        so = {k: v.copy() for k, v in s.items()}
        orc.evp(orc.make_domain(dom, grid), so)
        orc.set_strength_parameters()
        for k in EVP_OUT_FIELDS:
            assert np.array_equal(ref.get(k), so[k]), (nml, k)
        assert np.array_equal(ref.get("iceumask"), so["iceumask"])
        want = np.where(grid["tmask"] != 0, 917.0 * s["vice"] + 330.0 * s["vsno"], 0.0)
        assert np.array_equal(ref.get("sicemass"), want)
    # the thermodynamic module of the same library
    ref.init_thermo(); orc.init_thermo()
    ny, nx = 30, 44
    rng = np.random.default_rng(8)
    aice = np.where(rng.uniform(0, 1, (ny, nx)) < 0.8, rng.uniform(0.01, 1, (ny, nx)), 0.0)
    args = (2, nx - 1, 2, ny - 1, DT, aice, rng.uniform(-60, 20, (ny, nx)), -rng.uniform(1e6, 3e8, (20, ny, nx)),
            -rng.uniform(0, 5e7, (5, ny, nx)), np.full((ny, nx), -1.8) + rng.uniform(0, 1.5, (ny, nx)),
            np.full((ny, nx), -1.8), rng.uniform(-0.2, 0.2, (ny, nx)), rng.uniform(-0.2, 0.2, (ny, nx)))
    g = ref.frzmlt_bottom_lateral(*args); w = orc.frzmlt_bottom_lateral(*args)
    assert np.array_equal(g[1], w[1]) and relerr(g[2], w[2]) <= TOL_POW
    a, icells, ii, jj = synth.thermo_columns(30, 44, 2, regime="mixed", seed=77)
    ag = {k: v.copy() for k, v in a.items()}; ac = {k: v.copy() for k, v in a.items()}
    assert ref.thermo_vertical(DT, icells, ii, jj, ag, yday=100.0) == \
        orc.thermo_vertical(DT, icells, ii, jj, ac, yday=100.0) == (0, 0, 0)
    _cmp(ag, ac, "dropinaus")
    orc.set_chio()
