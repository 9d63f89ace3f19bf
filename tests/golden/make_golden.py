#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/ FROM THE COMPILED REFERENCE ITSELF
(oracle/_ref/libcice_ref_small.so, i.e. the Fortran under /root/reference built by
oracle/build_ref.sh with amdflang; flags recorded in each file's `meta`).

Only numbers are stored (inputs and the reference's outputs); no reference source.
Run from the repo root in the build container:  python tests/golden/make_golden.py

  evp_small.npz     whole evp(dt): 24x20 global grid as 2x2 blocks of 12x10 (cyclic E-W,
                    open N-S), non-uniform synthetic grid with an island, patchy ice cover,
                    dt=3600, ndte=120; inputs = module arrays before the call (set through
                    the reference's own arrays), outputs = module arrays after `call evp(dt)`.
  stress_stepu.npz  one call of `stress` (ksub = ndte, so the strain-rate diagnostics are
                    written) and of `stepu` on a 20x16 block with random index lists.
  evp_gx3.npz       whole `evp(dt)` on the reference's own gx3 displaced-pole grid and land mask (100x116)
  thermo_known_tsfc.npz  `thermo_vertical` with calc_Tsfc = F (surface fluxes given), 10x12 block
  thermo_cols.npz   `thermo_vertical` on a 10x12 block: conduct='MU71' 5 categories x 3 regimes,
                    conduct='bubbly' 2 categories.
  frzmlt.npz        `frzmlt_bottom_lateral` on a 14x18 block.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cice4_amd import lib, synth  # noqa: E402
from oracle import refapi  # noqa: E402

DT, NDTE = 3600.0, 120


def meta():
    fc = subprocess.run(["/opt/rocm/bin/amdflang", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    return np.array([f"reference: COSIMA/cice4 @ /root/reference; compiler: {fc}; flags: -O2 -fdefault-real-8 "
                     f"-ffp-contract=off (see oracle/build_ref.sh); generator: tests/golden/make_golden.py"])


def evp_small(ref):
    ctx = lib.Context()
    dom = ctx.domain_create(24, 20, 12, 10, ew=1, ns=0)   # host logic only
    nb = ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE)
    assert nb == 4 and ref.nx_block == dom["nx"] and ref.ny_block == dom["ny"]
    ref.set_strength_parameters()
    gg = synth.global_grid(24, 20, perturb=0.12, land_frac=0.03, seed=7)
    grid = synth.block_fields(gg, dom)
    s = synth.evp_state(grid, dom, seed=7, cover="patchy")
    for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear", "uarear",
              "tinyarea", "fcor"):
        ref.set(k, grid[k])
    ref.set("tmask", grid["tmask"].astype(float)); ref.set("umask", grid["umask"].astype(float))
    ins = ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty",
           "uvel", "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty") + synth.SIG_NAMES
    for k in ins:
        ref.set(k, s[k])
    ref.set("iceumask", s["iceumask"].astype(float))
    ny, nx = dom["ny"], dom["nx"]
    ref.set("aicen", s["aicen"].reshape(-1, ny, nx)); ref.set("vicen", s["vicen"].reshape(-1, ny, nx))
    ref.evp(DT)
    outs = ("uvel", "vvel", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT",
            "strocnyT", "strocnx", "strocny", "strintx", "strinty", "strairx", "strairy", "fm", "strtltx",
            "strtlty") + synth.SIG_NAMES
    out = {"out_" + k: ref.get(k) for k in outs}
    out["out_iceumask"] = ref.get("iceumask").astype(np.int32)
    data = {"grid_" + k: v for k, v in grid.items()}
    data.update({"in_" + k: v for k, v in s.items()})
    data.update(out)
    for k in ("ilo", "ihi", "jlo", "jhi", "i0", "j0", "hsrc", "hdst"):
        data["dom_" + k] = dom[k]
    data["meta"] = meta()
    np.savez_compressed(os.path.join(HERE, "evp_small.npz"), **data)


GX3_GRID = ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear", "uarear",
            "tinyarea", "fcor", "HTN", "HTE", "TLAT", "ULAT")


def evp_gx3():
    """evp(dt) on the reference's OWN gx3 grid: displaced-pole metrics and land mask read by the
    reference's init_grid1/2 from input_templates/gx3/global_gx3.{grid,kmt} (100x116, one block), ice on
    the two polar caps (|lat| > 55 deg, patchy), ndte = 120.  The grid arrays in the fixture are the
    reference's module arrays after its own initialisation -- numbers, not files."""
    ref = refapi.Ref("gx3")
    d = os.path.join(os.environ.get("CICE_REFERENCE_ROOT", "/root/reference"), "input_templates", "gx3")
    nb = ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE, grid="displaced_pole",
                         grid_file=os.path.join(d, "global_gx3.grid"), kmt_file=os.path.join(d, "global_gx3.kmt"))
    assert nb == 1
    ref.set_strength_parameters()
    ctx = lib.Context()
    dom = ctx.domain_create(100, 116, 100, 116, ew=1, ns=0)
    grid = {k: ref.get(k) for k in GX3_GRID}
    grid["tmask"] = ref.get("tmask").astype(np.int32); grid["umask"] = ref.get("umask").astype(np.int32)
    cap = np.abs(grid["TLAT"]) > np.deg2rad(55.0)
    s = synth.evp_state(grid, dom, seed=3, cover="patchy", ice_mask=cap)
    ins = ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty",
           "uvel", "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty") + synth.SIG_NAMES
    for k in ins:
        ref.set(k, s[k])
    ref.set("iceumask", s["iceumask"].astype(float))
    ny, nx = dom["ny"], dom["nx"]
    ref.set("aicen", s["aicen"].reshape(-1, ny, nx)); ref.set("vicen", s["vicen"].reshape(-1, ny, nx))
    ref.evp(DT)
    outs = ("uvel", "vvel", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT",
            "strocnyT", "strocnx", "strocny", "strintx", "strinty", "strairx", "strairy", "fm", "strtltx",
            "strtlty") + synth.SIG_NAMES
    data = {"grid_" + k: v for k, v in grid.items()}
    data.update({"in_" + k: v for k, v in s.items()})
    data.update({"out_" + k: ref.get(k) for k in outs})
    data["out_iceumask"] = ref.get("iceumask").astype(np.int32)
    for k in ("ilo", "ihi", "jlo", "jhi", "i0", "j0", "hsrc", "hdst"):
        data["dom_" + k] = dom[k]
    data["meta"] = meta()
    np.savez_compressed(os.path.join(HERE, "evp_gx3.npz"), **data)
    nice = int((s["aice"] > 0).sum())
    print("evp_gx3: ocean T-cells", int(grid["tmask"].sum()), "ice cells", nice, "max |u|", float(np.abs(data["out_uvel"]).max()))


def stress_stepu(ref):
    rng = np.random.default_rng(42)
    ny, nx = 16, 20
    U = lambda lo, hi: np.ascontiguousarray(rng.uniform(lo, hi, (ny, nx)))
    g = {k: U(2.5e4, 3.5e4) for k in ("dxt", "dyt", "cxp", "cyp")}
    g["cxm"] = -U(2.5e4, 3.5e4); g["cym"] = -U(2.5e4, 3.5e4)
    g["dxhy"] = U(-800, 800); g["dyhx"] = U(-800, 800)
    g["tarear"] = 1.0 / (g["dxt"] * g["dyt"]); g["tinyarea"] = 1e-11 * g["dxt"] * g["dyt"]
    uvel, vvel = U(-0.3, 0.3), U(-0.3, 0.3)
    strength = U(0, 4e4)
    tm = np.zeros((ny, nx), bool); tm[1:, 1:] = rng.uniform(0, 1, (ny - 1, nx - 1)) < 0.75
    jj, ii = np.nonzero(tm); icellt = len(ii)
    ti = np.zeros(nx * ny, np.int32); tj = np.zeros(nx * ny, np.int32); ti[:icellt] = ii + 1; tj[:icellt] = jj + 1
    sig_in = [U(-3e3, 3e3) for _ in range(12)]
    data = {"g_" + k: v for k, v in g.items()}
    data.update(uvel=uvel, vvel=vvel, strength=strength, icellt=icellt, indxti=ti, indxtj=tj,
                sig_in=np.array(sig_in))
    for damping in (0, 1):
        ref.set_evp_parameters(DT, NDTE, bool(damping))
        sig = [a.copy() for a in sig_in]
        diag = {k: np.zeros((ny, nx)) for k in ("shear", "divu", "prs_sig", "rdg_conv", "rdg_shear")}
        str8 = np.zeros((8, ny, nx))
        ref.stress(NDTE, icellt, ti, tj, uvel, vvel, g, strength, sig, diag, str8)
        data[f"sig_out_d{damping}"] = np.array(sig)
        data[f"str_d{damping}"] = str8
        for k, v in diag.items():
            data[f"{k}_d{damping}"] = v
    um = np.zeros((ny, nx), bool); um[1:-1, 1:-1] = rng.uniform(0, 1, (ny - 2, nx - 2)) < 0.8
    jj, ii = np.nonzero(um); icellu = len(ii)
    ui = np.zeros(nx * ny, np.int32); uj = np.zeros(nx * ny, np.int32); ui[:icellu] = ii + 1; uj[:icellu] = jj + 1
    ins = dict(aiu=U(0.1, 1), uocn=U(-0.1, 0.1), vocn=U(-0.1, 0.1), waterx=U(-0.1, 0.1), watery=U(-0.1, 0.1),
               forcex=U(-0.2, 0.2), forcey=U(-0.2, 0.2), umassdtei=U(5, 80), fm=U(-0.3, 0.3),
               uarear=1.0 / (U(2.5e4, 3.5e4) ** 2))
    io = [np.zeros((ny, nx)) for _ in range(4)] + [uvel.copy(), vvel.copy()]
    ref.stepu(icellu, ui, uj, ins["aiu"], data["str_d0"], ins["uocn"], ins["vocn"], ins["waterx"],
              ins["watery"], ins["forcex"], ins["forcey"], ins["umassdtei"], ins["fm"], ins["uarear"], *io)
    data.update({"su_" + k: v for k, v in ins.items()})
    data.update(icellu=icellu, indxui=ui, indxuj=uj, su_out=np.array(io))
    data["meta"] = meta()
    np.savez_compressed(os.path.join(HERE, "stress_stepu.npz"), **data)


def thermo_cols(ref):
    data = {"meta": meta()}
    for conduct in ("MU71", "bubbly"):
        salin, tmlt = ref.init_thermo(conduct=conduct)
        data[f"salin"] = salin; data["Tmlt"] = tmlt
        for regime in (("winter", "summer", "mixed") if conduct == "MU71" else ("mixed",)):
            for n in (range(5) if conduct == "MU71" else (0, 3)):
                a, icells, ii, jj = synth.thermo_columns(10, 12, n, regime=regime, seed=99)
                tag = f"{conduct}_{regime}_{n}"
                for k, v in a.items():
                    data[f"in_{tag}_{k}"] = v.copy()
                data[f"list_{tag}"] = np.array([icells] + list(ii[:icells]) + list(jj[:icells]), np.int32)
                st = ref.thermo_vertical(DT, icells, ii, jj, a, yday=180.0)
                data[f"stop_{tag}"] = np.array(st, np.int32)
                for k, v in a.items():
                    data[f"out_{tag}_{k}"] = v
    ref.init_thermo()
    np.savez_compressed(os.path.join(HERE, "thermo_cols.npz"), **data)


def thermo_known_tsfc(ref):
    """calc_Tsfc = F: inputs are the fluxes / surface temperature of the reference's own calc_Tsfc = T
    solve of the same columns, perturbed as synth.known_tsfc_inputs describes."""
    data = {"meta": meta()}
    for conduct in ("MU71", "bubbly"):
        for regime in ("winter", "summer", "mixed"):
            for n in ((0, 2, 4) if conduct == "MU71" else (1,)):
                a, icells, ii, jj = synth.thermo_columns(10, 12, n, regime=regime, seed=199)
                ref.init_thermo(conduct=conduct)
                t = {k: v.copy() for k, v in a.items()}
                assert ref.thermo_vertical(DT, icells, ii, jj, t, yday=180.0)[0] == 0
                b = synth.known_tsfc_inputs(a, t, seed=n)
                tag = f"{conduct}_{regime}_{n}"
                for k, v in b.items():
                    data[f"in_{tag}_{k}"] = v.copy()
                data[f"list_{tag}"] = np.array([icells] + list(ii[:icells]) + list(jj[:icells]), np.int32)
                ref.init_thermo(calc_Tsfc=False, conduct=conduct)
                st = ref.thermo_vertical(DT, icells, ii, jj, b, yday=180.0)
                data[f"stop_{tag}"] = np.array(st, np.int32)
                for k, v in b.items():
                    data[f"out_{tag}_{k}"] = v
    ref.init_thermo()
    np.savez_compressed(os.path.join(HERE, "thermo_known_tsfc.npz"), **data)


def frzmlt(ref):
    ref.init_thermo()
    rng = np.random.default_rng(8)
    ny, nx = 14, 18
    aice = np.where(rng.uniform(0, 1, (ny, nx)) < 0.8, rng.uniform(0.01, 1, (ny, nx)), 0.0)
    d = dict(aice=aice, frzmlt=rng.uniform(-60, 20, (ny, nx)), eicen=-rng.uniform(1e6, 3e8, (20, ny, nx)),
             esnon=-rng.uniform(0, 5e7, (5, ny, nx)), Tf=np.full((ny, nx), -1.8))
    d["sst"] = d["Tf"] + rng.uniform(0, 1.5, (ny, nx))
    d["strocnxT"] = rng.uniform(-0.2, 0.2, (ny, nx)); d["strocnyT"] = rng.uniform(-0.2, 0.2, (ny, nx))
    Tbot, fbot, rside = ref.frzmlt_bottom_lateral(2, nx - 1, 2, ny - 1, DT, d["aice"], d["frzmlt"], d["eicen"],
                                                  d["esnon"], d["sst"], d["Tf"], d["strocnxT"], d["strocnyT"])
    d.update(out_Tbot=Tbot, out_fbot=fbot, out_rside=rside, meta=meta())
    np.savez_compressed(os.path.join(HERE, "frzmlt.npz"), **d)


if __name__ == "__main__":
    ref = refapi.Ref("small")
    todo = sys.argv[1:] or ["stress_stepu", "thermo_cols", "thermo_known_tsfc", "frzmlt", "evp_small", "evp_gx3"]
    for name in ("stress_stepu", "thermo_cols", "thermo_known_tsfc", "frzmlt"):
        if name in todo:
            globals()[name](ref)
    if "evp_gx3" in todo:
        evp_gx3()           # its own library (gx3 configuration)
    if "evp_small" in todo:
        evp_small(ref)      # last: init_domain is once per process
    print("golden vectors", todo, "written to", HERE)
