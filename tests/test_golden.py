"""CPU checker against the committed golden vectors (tests/golden/*.npz, minted from the
compiled reference by tests/golden/make_golden.py).  Needs neither /root/reference nor
oracle/_ref: this is what pins the checker on the GPU box.  Bit-for-bit."""
import os

import numpy as np
import pytest

from cice4_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT, NDTE = 3600.0, 120


def load(name):
    return np.load(os.path.join(G, name))


def evp_case(fname="evp_small.npz", dims=(14, 12, 4, 24, 20)):
    z = load(fname)
    grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
    s = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("in_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    dom = dict(zip(("nx", "ny", "nblocks", "nxg", "nyg"), dims))
    for k in ("ilo", "ihi", "jlo", "jhi", "i0", "j0", "hsrc", "hdst"):
        dom[k] = z["dom_" + k]
    return dom, grid, s, out


def test_evp_small_golden(orc):
    dom, grid, s, out = evp_case()
    orc.set_evp_parameters(DT, NDTE); orc.set_strength_parameters()
    orc.evp(orc.make_domain(dom, grid), s)
    for k, v in out.items():
        assert np.array_equal(s[k], v), k
    assert np.abs(out["uvel"]).max() > 0.01


GX3 = ("evp_gx3.npz", (102, 118, 1, 100, 116))


def test_evp_gx3_real_grid_golden(orc):
    """The reference's own gx3 displaced-pole grid and land mask (grid arrays as its init_grid1/2 left
    them), ice on both polar caps: the checker reproduces the reference's evp(dt) bit for bit."""
    dom, grid, s, out = evp_case(*GX3)
    orc.set_evp_parameters(DT, NDTE); orc.set_strength_parameters()
    orc.evp(orc.make_domain(dom, grid), s)
    for k, v in out.items():
        assert np.array_equal(s[k], v), k
    assert np.abs(out["uvel"]).max() > 0.01 and 0 < (out["iceumask"] != 0).sum() < 4000


def test_stress_stepu_golden(orc):
    z = load("stress_stepu.npz")
    g = {k[2:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("g_")}
    ny, nx = z["uvel"].shape
    for damping in (0, 1):
        orc.set_evp_parameters(DT, NDTE, bool(damping))
        sig = [np.ascontiguousarray(a).copy() for a in z["sig_in"]]
        diag = {k: np.zeros((ny, nx)) for k in ("shear", "divu", "prs_sig", "rdg_conv", "rdg_shear")}
        str8 = np.ones((8, ny, nx))
        orc.stress(NDTE, int(z["icellt"]), z["indxti"], z["indxtj"], z["uvel"], z["vvel"], g, z["strength"],
                   sig, diag, str8)
        assert np.array_equal(np.array(sig), z[f"sig_out_d{damping}"])
        assert np.array_equal(str8, z[f"str_d{damping}"])
        for k in diag:
            assert np.array_equal(diag[k], z[f"{k}_d{damping}"]), k
    io = [np.zeros((ny, nx)) for _ in range(4)] + [z["uvel"].copy(), z["vvel"].copy()]
    su = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("su_") and k != "su_out"}
    orc.stepu(int(z["icellu"]), z["indxui"], z["indxuj"], su["aiu"], np.ascontiguousarray(z["str_d0"]), su["uocn"],
              su["vocn"], su["waterx"], su["watery"], su["forcex"], su["forcey"], su["umassdtei"], su["fm"],
              su["uarear"], *io)
    assert np.array_equal(np.array(io), z["su_out"])


def thermo_cases(fname="thermo_cols.npz"):
    z = load(fname)
    tags = sorted(k[5:] for k in z.files if k.startswith("list_"))
    for tag in tags:
        conduct = tag.split("_")[0]
        a = {k[len(f"in_{tag}_"):]: np.ascontiguousarray(z[k]).copy() for k in z.files if k.startswith(f"in_{tag}_")}
        out = {k[len(f"out_{tag}_"):]: z[k] for k in z.files if k.startswith(f"out_{tag}_")}
        lst = z[f"list_{tag}"]; icells = int(lst[0])
        ny, nx = a["aicen"].shape
        ii = np.zeros(nx * ny, np.int32); jj = np.zeros(nx * ny, np.int32)
        ii[:icells] = lst[1:1 + icells]; jj[:icells] = lst[1 + icells:1 + 2 * icells]
        yield tag, conduct, a, out, icells, ii, jj, tuple(int(x) for x in z[f"stop_{tag}"])


def test_thermo_golden(orc):
    n = 0
    for tag, conduct, a, out, icells, ii, jj, stop in thermo_cases():
        orc.init_thermo(conduct=conduct)
        assert orc.thermo_vertical(DT, icells, ii, jj, a, yday=180.0) == stop
        for k in out:
            assert np.array_equal(a[k], out[k]), (tag, k)
        n += icells
    assert n > 1000
    orc.init_thermo()


def test_thermo_known_tsfc_golden(orc):
    """calc_Tsfc = F vectors minted by the compiled reference (get_matrix_elements_know_Tsfc path)."""
    n = 0
    for tag, conduct, a, out, icells, ii, jj, stop in thermo_cases("thermo_known_tsfc.npz"):
        orc.init_thermo(calc_Tsfc=False, conduct=conduct)
        assert orc.thermo_vertical(DT, icells, ii, jj, a, yday=180.0) == stop == (0, 0, 0)
        for k in out:
            assert np.array_equal(a[k], out[k]), (tag, k)
        n += icells
    assert n > 800
    orc.init_thermo()


def test_frzmlt_golden(orc):
    z = load("frzmlt.npz")
    orc.init_thermo()
    ny, nx = z["aice"].shape
    c = lambda k: np.ascontiguousarray(z[k])
    r = orc.frzmlt_bottom_lateral(2, nx - 1, 2, ny - 1, DT, c("aice"), c("frzmlt"), c("eicen"), c("esnon"), c("sst"),
                                  c("Tf"), c("strocnxT"), c("strocnyT"))
    for a, k in zip(r, ("out_Tbot", "out_fbot", "out_rside")):
        assert np.array_equal(a, z[k]), k


def test_golden_files_carry_provenance():
    for f in ("evp_small.npz", "evp_gx3.npz", "stress_stepu.npz", "thermo_cols.npz", "thermo_known_tsfc.npz", "frzmlt.npz"):
        assert "amdflang" in str(load(f)["meta"][0]) or "flang" in str(load(f)["meta"][0])
