"""The AusCOM / coupled flavour of the hot path (bld/Macros.nci:56-57: -DAusCOM -Dcoupled, constants of
drivers/access-om/ice_constants.F90): the CPU checker built with those constants and switched to those branches
against the reference COMPILED that way (AUS=1 oracle/build_ref.sh -> oracle/_ref/libcice_refaus_<cfg>.so),
bit for bit.  What differs from the stand-alone build: cosw / sinw / dragio / chio are namelist variables, the
ocean turning angle rotates with the hemisphere (sign of fm) in evp_prep2 / stepu / evp_finish, the tilt term comes
from the ocean's surface slope when use_ocnslope is set, cp_ocn and ice_ref_salinity are MOM's."""
import tempfile

import numpy as np
import pytest

from cice4_amd import lib, synth
from test_oracle_vs_ref import DT, NDTE, EVP_OUT, inject

ANGLE = np.deg2rad(25.0)   # a turning angle as used with a non-resolving ocean boundary layer
NAMELISTS = [dict(cosw=1.0, sinw=0.0, dragio=0.00536, use_ocnslope=False),
             dict(cosw=float(np.cos(ANGLE)), sinw=float(np.sin(ANGLE)), dragio=0.00536, use_ocnslope=False),
             dict(cosw=float(np.cos(ANGLE)), sinw=float(np.sin(ANGLE)), dragio=0.0035, use_ocnslope=True)]


def both_set(ref, orc, nml, chio=0.006):
    ref.set_auscom(chio=chio, **nml)
    orc.set_auscom(True, **nml); orc.set_chio(chio)


def two_hemispheres(grid):
    """the synthetic grid sits in one hemisphere; mirror the Coriolis parameter over the middle row so that fm takes
    both signs (and is exactly zero nowhere but on land)"""
    f = grid["fcor"]
    nyh = f.shape[1] // 2
    f[:, :nyh] = -np.abs(f[:, :nyh])
    f[:, nyh:] = np.abs(f[:, nyh:])
    return grid


@pytest.mark.parametrize("nml", NAMELISTS)
def test_prep2_stepu_finish(refaus_gx3b4, orc_aus, nml):
    ref, orc = refaus_gx3b4, orc_aus
    both_set(ref, orc, nml)
    ny, nx = 30, 40
    dom = dict(nx=nx, ny=ny, nblocks=1, ilo=[2], ihi=[nx - 1], jlo=[2], jhi=[ny - 1], i0=[0], j0=[0],
               nxg=nx - 2, nyg=ny - 2)
    grid = synth.block_fields(synth.global_grid(nx - 2, ny - 2, perturb=0.1, land_frac=0.05), dom)
    s = synth.evp_state(grid, dom, cover="patchy")
    a1 = (2, nx - 1, 2, ny - 1, s["aice"][0], s["vice"][0], s["vsno"][0], grid["tmask"][0],
          s["strairxT"][0], s["strairyT"][0])
    icetmask = orc.evp_prep1(*a1)[3]
    ref.set_evp_parameters(DT, NDTE); orc.set_evp_parameters(DT, NDTE)

    def mk():
        r = np.random.default_rng(5)
        U = lambda lo, hi: np.ascontiguousarray(r.uniform(lo, hi, (ny, nx)))
        return dict(aiu=U(0, 1) * (U(0, 1) > 0.2), umass=U(0, 900), umassdtei=U(0, 1), fcor=U(-1e-4, 1e-4),
                    umask=np.ascontiguousarray(grid["umask"][0]), uocn=U(-.1, .1), vocn=U(-.1, .1),
                    strairx=U(-.1, .1), strairy=U(-.1, .1), ss_tltx=U(-1e-5, 1e-5), ss_tlty=U(-1e-5, 1e-5),
                    icetmask=icetmask.copy(), iceumask=(U(0, 1) > 0.5).astype(np.int32), fm=U(-1, 1),
                    strtltx=U(-1, 1), strtlty=U(-1, 1), strocnx=U(-1, 1), strocny=U(-1, 1), strintx=U(-1, 1),
                    strinty=U(-1, 1), waterx=U(-1, 1), watery=U(-1, 1), forcex=U(-1, 1), forcey=U(-1, 1),
                    sig=[U(-1e3, 1e3) for _ in range(12)], uvel=U(-.2, .2), vvel=U(-.2, .2))
    ar, ao = mk(), mk()
    rr = ref.evp_prep2(2, nx - 1, 2, ny - 1, ar); ro = orc.evp_prep2(2, nx - 1, 2, ny - 1, ao)
    assert rr[0] == ro[0] and rr[1] == ro[1] and rr[1] > 100
    for k in ar:
        if k == "sig":
            for x, y in zip(ar[k], ao[k]):
                assert np.array_equal(x, y)
        else:
            assert np.array_equal(ar[k], ao[k]), k
    icellu, ui, uj = rr[1], rr[2][2], rr[2][3]
    fm = ar["fm"]
    assert (fm[uj[:icellu] - 1, ui[:icellu] - 1] < 0).any() and (fm[uj[:icellu] - 1, ui[:icellu] - 1] > 0).any()
    if nml["sinw"]:   # the hemisphere matters: the same cell with fm mirrored gives another water stress
        assert not np.array_equal(ar["waterx"], ar["uocn"] * nml["cosw"] - ar["vocn"] * nml["sinw"])

    # stepu on the prepared fields
    r = np.random.default_rng(9)
    str8 = np.ascontiguousarray(r.uniform(-1e3, 1e3, (8, ny, nx)))
    uarear = np.ascontiguousarray(1.0 / grid["uarea"][0])
    aiu = np.maximum(ar["aiu"], 0.01)
    outs = []
    for api, a in ((ref, ar), (orc, ao)):
        o = {k: a[k].copy() for k in ("strocnx", "strocny", "strintx", "strinty", "uvel", "vvel")}
        api.stepu(icellu, ui, uj, aiu, str8, a["uocn"], a["vocn"], a["waterx"], a["watery"], a["forcex"],
                  a["forcey"], a["umassdtei"], a["fm"], uarear, o["strocnx"], o["strocny"], o["strintx"],
                  o["strinty"], o["uvel"], o["vvel"])
        outs.append(o)
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert not np.array_equal(outs[0]["uvel"], ar["uvel"])

    fr = [ar[k].copy() for k in ("strocnx", "strocny")] + [np.ones((ny, nx)), np.ones((ny, nx))]
    fo = [a.copy() for a in fr]
    ref.evp_finish_fm(icellu, ui, uj, ar["uvel"], ar["vvel"], ar["uocn"], ar["vocn"], aiu, fm, *fr)
    orc.evp_finish_fm(icellu, ui, uj, ar["uvel"], ar["vvel"], ar["uocn"], ar["vocn"], aiu, fm, *fo)
    for x, y in zip(fr, fo):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("nml", NAMELISTS[1:])
def test_whole_evp(refaus_gx3b4, orc_aus, nml):
    """evp(dt), 120 subcycles, 2x2 blocks with the reference's own halo, both hemispheres on the grid"""
    ref, orc = refaus_gx3b4, orc_aus
    nb = ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE)
    both_set(ref, orc, nml)
    dom = lib.Context().domain_create(100, 116, 50, 58, ew=1, ns=0)
    assert nb == dom["nblocks"]
    grid = two_hemispheres(synth.block_fields(synth.global_grid(100, 116, perturb=0.15, land_frac=0.05), dom))
    for cover, damping in (("full", False), ("patchy", True)):
        s = synth.evp_state(grid, dom, cover=cover)
        ref.set_evp_parameters(DT, NDTE, damping); ref.set_strength_parameters()
        orc.set_evp_parameters(DT, NDTE, damping); orc.set_strength_parameters()
        inject(ref, grid, s, dom)
        ref.evp(DT)
        so = {k: v.copy() for k, v in s.items()}
        orc.evp(orc.make_domain(dom, grid), so)
        for k in EVP_OUT:
            assert np.array_equal(ref.get(k), so[k]), (cover, damping, k)
        assert (so["fm"] < 0).any() and (so["fm"] > 0).any() and np.abs(so["uvel"]).max() > 0.01
        # the coupler's ice + snow mass, written by the reference's evp (ice_dyn_evp.F90:246-248)
        assert np.array_equal(ref.get("sicemass"), np.where(grid["tmask"] != 0, 917.0 * s["vice"] + 330.0 * s["vsno"], 0.0))


def test_the_stand_alone_namelist_on_the_auscom_build_is_not_the_stand_alone_build(refaus_gx3b4, ref_gx3b4):
    """with sinw = 0 the dynamics coincide with the stand-alone build (the hemisphere only enters through sinw);
    the thermodynamic constants do not: frzmlt_bottom_lateral differs through cp_ocn"""
    ny, nx = 30, 44
    rng = np.random.default_rng(8)
    refaus_gx3b4.init_thermo(); ref_gx3b4.init_thermo(); refaus_gx3b4.set_auscom()
    aice = np.where(rng.uniform(0, 1, (ny, nx)) < 0.8, rng.uniform(0.01, 1, (ny, nx)), 0.0)
    args = (2, nx - 1, 2, ny - 1, DT, aice, rng.uniform(-60, 20, (ny, nx)), -rng.uniform(1e6, 3e8, (20, ny, nx)),
            -rng.uniform(0, 5e7, (5, ny, nx)), np.full((ny, nx), -1.8) + rng.uniform(0, 1.5, (ny, nx)),
            np.full((ny, nx), -1.8), rng.uniform(-0.2, 0.2, (ny, nx)), rng.uniform(-0.2, 0.2, (ny, nx)))
    a, b = refaus_gx3b4.frzmlt_bottom_lateral(*args), ref_gx3b4.frzmlt_bottom_lateral(*args)
    assert np.array_equal(a[0], b[0]) and not np.array_equal(a[1], b[1])


@pytest.mark.parametrize("chio", [0.006, 0.004])
def test_frzmlt_bottom_lateral(refaus_gx3b4, orc_aus, chio):
    ref, orc = refaus_gx3b4, orc_aus
    ref.init_thermo(); orc.init_thermo()
    both_set(ref, orc, NAMELISTS[0], chio=chio)
    ny, nx = 30, 44
    rng = np.random.default_rng(8)
    aice = np.where(rng.uniform(0, 1, (ny, nx)) < 0.8, rng.uniform(0.01, 1, (ny, nx)), 0.0)
    args = (2, nx - 1, 2, ny - 1, DT, aice, rng.uniform(-60, 20, (ny, nx)), -rng.uniform(1e6, 3e8, (20, ny, nx)),
            -rng.uniform(0, 5e7, (5, ny, nx)), np.full((ny, nx), -1.8) + rng.uniform(0, 1.5, (ny, nx)),
            np.full((ny, nx), -1.8), rng.uniform(-0.2, 0.2, (ny, nx)), rng.uniform(-0.2, 0.2, (ny, nx)))
    got = orc.frzmlt_bottom_lateral(*args)
    for x, y in zip(ref.frzmlt_bottom_lateral(*args), got):
        assert np.array_equal(x, y)
    assert (got[1] < 0).any()
    both_set(ref, orc, NAMELISTS[0])


@pytest.mark.parametrize("conduct", ["MU71", "bubbly"])
def test_thermo_vertical(refaus_gx3b4, orc_aus, conduct):
    """cp_ocn enters the enthalpy <-> temperature relation and ice_ref_salinity the salt flux"""
    ref, orc = refaus_gx3b4, orc_aus
    sr, tr = ref.init_thermo(conduct=conduct); so, to = orc.init_thermo(conduct=conduct)
    assert np.array_equal(sr, so) and np.array_equal(tr, to)
    for regime in ("winter", "summer", "mixed"):
        for n in range(3):
            a, icells, ii, jj = synth.thermo_columns(40, 50, n, regime=regime)
            a1 = {k: v.copy() for k, v in a.items()}; a2 = {k: v.copy() for k, v in a.items()}
            l1 = ref.thermo_vertical(DT, icells, ii, jj, a1, yday=123.0)
            l2 = orc.thermo_vertical(DT, icells, ii, jj, a2, yday=123.0)
            assert l1 == l2 == (0, 0, 0)
            for k in a1:
                assert np.array_equal(a1[k], a2[k]), (regime, n, k)
    ref.init_thermo(); orc.init_thermo()
