"""Deterministic synthetic grids, forcing and ice states for the EVP and
column-thermodynamics hot path (host-side numpy; no reference data needed).

Follows the reference's own synthetic set-up where it has one:
  * rectangular grid: source/ice_grid.F90:976-1130 (`rectgrid`: 30 km cells,
    land on the two southern- and northernmost rows, ULAT from 71.35 deg growing
    by dy/radius per row), metric formulas ice_grid.F90:332-363 and
    primary_grid_lengths_* :1139-1289, masks `makemask` :1298;
  * default forcing: source/ice_flux.F90:316-377;
  * default ice state: source/ice_init.F90:1040-1193 (category thicknesses from
    hin_max, ice_itd.F90:162-186, kcatbound = 0; linear temperature profile).
`perturb` > 0 makes the grid non-uniform so that every metric term of the
stress kernel (dxhy, dyhx, cxp != -cxm ...) is exercised.

Arrays: a Fortran (nx_block,ny_block,nblocks) field is a C-order numpy array
(nblocks, ny_block, nx_block); ghost width 1; 1-based reference indices map to
python index - 1.
"""
import numpy as np

NCAT, NILYR, NSLYR = 5, 4, 1
# drivers/cice4/ice_constants.F90:49-121
rhos, rhoi, rhow = 330.0, 917.0, 1026.0
cp_ice, cp_ocn, depressT = 2106.0, 4218.0, 0.054
Lsub, Lvap = 2.835e6, 2.501e6
Lfresh = Lsub - Lvap
omega, radius = 7.292e-5, 6.37e6
puny = 1.0e-11
saltmax = 3.2


def hin_max(ncat=NCAT):
    """ice_itd.F90:162-186 (kcatbound = 0, kitd = 1)."""
    rncat = float(ncat)
    cc1 = 3.0 / rncat
    cc2 = 15.0 * cc1
    cc3 = 3.0
    h = [0.0]
    for n in range(1, ncat + 1):
        x1 = float(n - 1) / rncat
        h.append(h[-1] + cc1 + cc2 * (1.0 + np.tanh(cc3 * (x1 - 1.0))))
    return np.array(h)


def salinity_profile():
    """ice_therm_vertical.F90:567-576."""
    nsal, msal = 0.407, 0.573
    salin = np.zeros(NILYR + 1)
    for k in range(1, NILYR + 1):
        zn = (k - 0.5) / NILYR
        salin[k - 1] = (saltmax / 2.0) * (1.0 - np.cos(np.pi * zn ** (nsal / (msal + zn))))
    salin[NILYR] = saltmax
    return salin, -salin * depressT


# ----------------------------------------------------------------------------
# global grid
# ----------------------------------------------------------------------------
def global_grid(nxg, nyg, perturb=0.0, seed=20261003, dx=30.0e3, dy=30.0e3, land_rows=2,
                land_frac=0.0):
    """Global (nyg, nxg) arrays HTN, HTE, ULAT, hm."""
    rng = np.random.default_rng(seed)
    HTN = np.full((nyg, nxg), dx)
    HTE = np.full((nyg, nxg), dy)
    if perturb > 0.0:
        # smooth-ish multiplicative perturbation, different in i and j
        jj, ii = np.meshgrid(np.arange(nyg), np.arange(nxg), indexing="ij")
        HTN *= 1.0 + perturb * (np.sin(2 * np.pi * ii / nxg * 3 + 0.3) * np.cos(np.pi * jj / nyg)
                                + 0.2 * rng.uniform(-1, 1, (nyg, nxg)))
        HTE *= 1.0 + perturb * (np.cos(2 * np.pi * ii / nxg * 2) * np.sin(np.pi * jj / nyg * 2 + 0.1)
                                + 0.2 * rng.uniform(-1, 1, (nyg, nxg)))
    length = dy / radius  # radians per row
    ULAT = np.radians(71.35) + length * np.arange(nyg)[:, None] * np.ones((1, nxg))
    hm = np.zeros((nyg, nxg))
    hm[land_rows:nyg - land_rows, :] = 1.0
    if land_frac > 0.0:
        # a few rectangular islands
        n_isl = max(1, int(land_frac * 20))
        for _ in range(n_isl):
            w = max(2, int(nxg * np.sqrt(land_frac / n_isl)))
            h = max(2, int(nyg * np.sqrt(land_frac / n_isl)))
            i0 = rng.integers(0, nxg - w)
            j0 = rng.integers(land_rows, max(land_rows + 1, nyg - land_rows - h))
            hm[j0:j0 + h, i0:i0 + w] = 0.0
    return dict(HTN=HTN, HTE=HTE, ULAT=ULAT, hm=hm, nxg=nxg, nyg=nyg)


def _ext(g, ew_cyclic=True, ns_cyclic=False):
    """Extend a global (nyg,nxg) array by 2 cells on each side: wrap (cyclic) or
    replicate in i, and in j."""
    a = np.concatenate([g[:, -2:], g, g[:, :2]], axis=1) if ew_cyclic else \
        np.concatenate([g[:, :1], g[:, :1], g, g[:, -1:], g[:, -1:]], axis=1)
    if ns_cyclic:
        return np.concatenate([a[-2:], a, a[:2]], axis=0)
    return np.concatenate([a[:1], a[:1], a, a[-1:], a[-1:]], axis=0)


def block_fields(gg, dom, ew_cyclic=True, north_ocean=False, ns_cyclic=False):
    """Per-block grid arrays (nblocks, ny_block, nx_block) for the blocks described by
    `dom` (dict with nx, ny, nblocks, ilo, ihi, jlo, jhi, i0, j0: 0-based global index of
    local cell ilo / jlo).  Metrics follow init_grid2 (ice_grid.F90:332-363).
    north_ocean: the cells beyond the northern edge are ocean where the top row is (a tripole grid continues across
    the fold), so that the U points ON the edge are ocean too; otherwise land surrounds the domain north and south.
    ns_cyclic: the grid wraps north-south as well (nothing closes the domain there)."""
    nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
    E = {k: _ext(gg[k], ew_cyclic, ns_cyclic) for k in ("HTN", "HTE", "ULAT", "hm")}
    # dxu, dxt, dyu, dyt on the extended global grid (ice_grid.F90:1139-1289)
    HTN, HTE = E["HTN"], E["HTE"]
    dxu = 0.5 * (HTN + np.roll(HTN, -1, axis=1))
    dxt = 0.5 * (HTN + np.roll(HTN, 1, axis=0))
    dyu = 0.5 * (HTE + np.roll(HTE, -1, axis=0))
    dyt = 0.5 * (HTE + np.roll(HTE, 1, axis=1))
    hm = E["hm"]
    if not ew_cyclic:
        hm[:, :2] = 0.0
        hm[:, -2:] = 0.0
    if not ns_cyclic:
        hm[:2, :] = 0.0
        if not north_ocean:
            hm[-2:, :] = 0.0
    uvm = np.minimum(np.minimum(hm, np.roll(hm, -1, axis=1)),
                     np.minimum(np.roll(hm, -1, axis=0), np.roll(np.roll(hm, -1, axis=0), -1, axis=1)))
    G = dict(HTN=HTN, HTE=HTE, dxu=dxu, dxt=dxt, dyu=dyu, dyt=dyt, ULAT=E["ULAT"], hm=hm, uvm=uvm)
    G["HTE_w"] = np.roll(HTE, 1, axis=1)   # HTE(i-1,j)
    G["HTN_s"] = np.roll(HTN, 1, axis=0)   # HTN(i,j-1)
    out = {k: np.zeros((nb, ny, nx)) for k in
           ("dxt", "dyt", "dxu", "dyu", "HTN", "HTE", "tarea", "uarea", "tarear", "uarear",
            "tinyarea", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "ULAT", "fcor", "hm", "uvm")}
    for b in range(nb):
        # local python index li <-> extended global index: (li - (ilo-1)) + i0 + 2
        gi = np.arange(nx) - (dom["ilo"][b] - 1) + dom["i0"][b] + 2
        gj = np.arange(ny) - (dom["jlo"][b] - 1) + dom["j0"][b] + 2
        gi = np.clip(gi, 0, HTN.shape[1] - 1)
        gj = np.clip(gj, 0, HTN.shape[0] - 1)
        sel = np.ix_(gj, gi)
        L = {k: G[k][sel] for k in G}
        out["dxt"][b], out["dyt"][b], out["dxu"][b], out["dyu"][b] = L["dxt"], L["dyt"], L["dxu"], L["dyu"]
        out["HTN"][b], out["HTE"][b], out["ULAT"][b] = L["HTN"], L["HTE"], L["ULAT"]
        out["hm"][b], out["uvm"][b] = L["hm"], L["uvm"]
        out["tarea"][b] = L["dxt"] * L["dyt"]
        out["uarea"][b] = L["dxu"] * L["dyu"]
        out["tarear"][b] = 1.0 / out["tarea"][b]
        out["uarear"][b] = 1.0 / out["uarea"][b]
        out["tinyarea"][b] = puny * out["tarea"][b]
        out["dxhy"][b] = 0.5 * (L["HTE"] - L["HTE_w"])
        out["dyhx"][b] = 0.5 * (L["HTN"] - L["HTN_s"])
        out["cyp"][b] = 1.5 * L["HTE"] - 0.5 * L["HTE_w"]
        out["cxp"][b] = 1.5 * L["HTN"] - 0.5 * L["HTN_s"]
        out["cym"][b] = -(1.5 * L["HTE_w"] - 0.5 * L["HTE"])
        out["cxm"][b] = -(1.5 * L["HTN_s"] - 0.5 * L["HTN"])
        out["fcor"][b] = 2.0 * omega * np.sin(L["ULAT"])   # ice_dyn_evp.F90:503
    out["tmask"] = (out["hm"] > 0.5).astype(np.int32)
    out["umask"] = (out["uvm"] > 0.5).astype(np.int32)
    return out


# ----------------------------------------------------------------------------
# ice state for EVP
# ----------------------------------------------------------------------------
def evp_state(grid, dom, seed=20261003, cover="full", moving=True, ice_mask=None):
    """Module-array-shaped inputs of evp(dt) (ice_dyn_evp.F90:119) for the blocks in dom.
    cover: 'full' (ice on every ocean cell), 'patchy' (ice-free regions, thin-ice edges), 'caps' (two polar caps with a
    wavy edge, ~25 % of the rows: what a global grid looks like to the dynamics -- most rows are open water).
    ice_mask: optional boolean (nblocks, ny, nx): ice only there (e.g. the polar caps of a real grid)."""
    rng = np.random.default_rng(seed)
    nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
    shp = (nb, ny, nx)
    tm = grid["tmask"].astype(bool)
    hmax = hin_max()
    # smooth concentration field built from global coordinates so that halos agree
    gi = np.zeros(shp); gj = np.zeros(shp)
    for b in range(nb):
        ii = np.arange(nx) - (dom["ilo"][b] - 1) + dom["i0"][b]
        jj = np.arange(ny) - (dom["jlo"][b] - 1) + dom["j0"][b]
        gi[b], gj[b] = np.meshgrid(ii % dom["nxg"], jj, indexing="xy")
    nxg, nyg = dom["nxg"], dom["nyg"]

    def field(k, lo, hi):
        f = 0.5 + 0.25 * np.sin(2 * np.pi * (k + 1) * gi / nxg + k) * np.cos(np.pi * (k + 2) * gj / nyg) \
            + 0.25 * np.sin(2 * np.pi * gi / nxg * (k + 3) + 1.7 * gj / nyg * (k + 1))
        return lo + (hi - lo) * np.clip(f, 0.0, 1.0)

    conc = field(0, 0.6, 0.99)
    if cover == "patchy":
        hole = field(1, 0.0, 1.0)
        conc = np.where(hole < 0.35, 0.0, conc * np.clip((hole - 0.35) / 0.15, 0.0, 1.0))
    if cover == "caps":
        lat = (gj + 0.5) / nyg + 0.02 * np.sin(2 * np.pi * 5 * gi / nxg) + 0.01 * np.cos(2 * np.pi * 13 * gi / nxg)
        edge = np.minimum(np.clip((0.10 - lat) / 0.02, 0.0, 1.0) + np.clip((lat - 0.84) / 0.02, 0.0, 1.0), 1.0)
        conc = conc * edge
    conc = np.where(tm, conc, 0.0)
    if ice_mask is not None:
        conc = np.where(ice_mask, conc, 0.0)
    w = np.array([0.1, 0.2, 0.3, 0.25, 0.15])
    aicen = np.zeros((nb, NCAT, ny, nx)); vicen = np.zeros_like(aicen); vsnon = np.zeros_like(aicen)
    for n in range(NCAT):
        frac = w[n] * (0.6 + 0.8 * field(n + 2, 0.0, 1.0))
        aicen[:, n] = conc * frac
        h = hmax[n] + (min(hmax[n + 1], hmax[n] + 2.0) - hmax[n]) * field(n + 7, 0.2, 0.8)
        vicen[:, n] = aicen[:, n] * h
        vsnon[:, n] = aicen[:, n] * 0.2 * field(n + 11, 0.0, 1.0)
    tot = aicen.sum(axis=1)
    scale = np.where(tot > 0.99, 0.99 / np.maximum(tot, 1e-30), 1.0)
    aicen *= scale[:, None]; vicen *= scale[:, None]; vsnon *= scale[:, None]
    aice = aicen.sum(axis=1); vice = vicen.sum(axis=1); vsno = vsnon.sum(axis=1)
    s = dict(aicen=aicen, vicen=vicen, aice=aice, vice=vice, vsno=vsno, aice0=1.0 - aice)
    # wind stress on the T grid, already multiplied by aice (ice_dyn_evp.F90:666-669)
    s["strairxT"] = aice * (0.08 + 0.06 * np.sin(2 * np.pi * gj / nyg * 2) + 0.03 * np.cos(2 * np.pi * gi / nxg * 3))
    s["strairyT"] = aice * (0.04 * np.cos(2 * np.pi * gi / nxg * 2) - 0.05 * np.sin(2 * np.pi * gj / nyg))
    um = grid["umask"].astype(bool)
    s["uocn"] = np.where(um, 0.05 * np.sin(2 * np.pi * gj / nyg) + 0.02 * np.cos(2 * np.pi * gi / nxg * 2), 0.0)
    s["vocn"] = np.where(um, 0.04 * np.cos(2 * np.pi * gi / nxg) * np.sin(np.pi * gj / nyg), 0.0)
    s["ss_tltx"] = np.zeros(shp); s["ss_tlty"] = np.zeros(shp)
    z = lambda: np.zeros(shp)
    for n in ("fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty", "strairx",
              "strairy", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig",
              "strocnxT", "strocnyT"):
        s[n] = z()
    s["uvel"] = z(); s["vvel"] = z()
    s["iceumask"] = np.zeros(shp, np.int32)
    if moving:
        # a previous step's velocity / stress state on (roughly) the new ice cover
        act = um & (np.roll(aice, -1, axis=2) + aice > 0.3)
        s["iceumask"] = act.astype(np.int32)
        s["uvel"] = np.where(act, 0.08 * np.sin(2 * np.pi * gi / nxg * 2 + 0.5) * np.cos(np.pi * gj / nyg), 0.0)
        s["vvel"] = np.where(act, 0.06 * np.cos(2 * np.pi * gi / nxg * 3) * np.sin(2 * np.pi * gj / nyg + 0.2), 0.0)
        for k, n in enumerate(SIG_NAMES):
            amp = 2.0e3 if k < 4 else 8.0e2
            base = -1.0 if k < 4 else 0.0
            s[n] = np.where(aice > 0.05, amp * (base + 0.5 * np.sin(2 * np.pi * gi / nxg * (k % 4 + 1) + k)
                                                 * np.cos(np.pi * gj / nyg * (k % 3 + 1))), 0.0)
    else:
        for n in SIG_NAMES:
            s[n] = z()
    for k in s:
        s[k] = np.ascontiguousarray(s[k])
    return s


SIG_NAMES = ("stressp_1", "stressp_2", "stressp_3", "stressp_4", "stressm_1", "stressm_2",
             "stressm_3", "stressm_4", "stress12_1", "stress12_2", "stress12_3", "stress12_4")


# ----------------------------------------------------------------------------
# column states for thermo_vertical
# ----------------------------------------------------------------------------
def qin_from_T(T, Tmlt):
    """ice_therm_vertical.F90:1989-1991"""
    return -rhoi * (cp_ice * (Tmlt - T) + Lfresh * (1.0 - Tmlt / T) - cp_ocn * Tmlt)


def _smooth_uniform(rng, shp, length):
    """A spatially correlated field with (close to) uniform(0,1) marginals: Gaussian-filtered white
    noise (correlation length `length` cells, periodic), mapped through its own normal CDF."""
    from math import sqrt
    from scipy.ndimage import gaussian_filter
    from scipy.special import erf
    z = gaussian_filter(rng.standard_normal(shp), length, mode="wrap")
    z = (z - z.mean()) / max(z.std(), 1e-300)
    return 0.5 * (1.0 + erf(z / sqrt(2.0)))


def thermo_columns(ny, nx, ncat_index=0, seed=20261003, regime="mixed", ice_frac=0.9, coherent=0):
    """Arguments of thermo_vertical (ice_therm_vertical.F90:108-132) for ONE category on an
    (nx,ny) block: state, forcing and flux arrays + the compressed (indxi,indxj) list built
    the way step_therm1 builds it (aicen > puny, j outer / i inner, physical cells only;
    drivers/cice4/CICE_RunMod.F90:380-389).
    regime: 'winter' (cold, snow covered), 'summer' (melting, strong SW), 'mixed'.
    coherent: 0 = every yes/no property of a column (melting or cold, snow or bare, precipitation, ...)
    is drawn independently per cell (white noise: the harshest case for a SIMD machine and what the
    parity tests use); L > 0 = the same properties with the same frequencies, but as regions with
    correlation length L cells, the way weather and ice cover are organised (bench.py)."""
    rng = np.random.default_rng(seed + 1000 * ncat_index)
    shp = (ny, nx)
    # coin(p): boolean field, true with probability p
    if coherent:
        coin_rng = np.random.default_rng(seed + 77)   # the same regions for every category
        regions = {}

        def coin(p, key=None, own=False):
            r = rng if own else coin_rng
            if key is None or key not in regions:
                f = _smooth_uniform(r, shp, coherent)
                if key is not None:
                    regions[key] = f
            else:
                f = regions[key]
            return f < p
    else:
        def coin(p, key=None, own=False):
            return rng.uniform(0, 1, shp) < p
    hmax = hin_max()
    n = ncat_index
    _, Tmlt = salinity_profile()
    U = lambda lo, hi: rng.uniform(lo, hi, shp)
    present = coin(ice_frac, own=True)
    present[0, :] = present[-1, :] = False
    present[:, 0] = present[:, -1] = False
    aicen = np.where(present, U(0.02, 0.95), 0.0)
    hi_lo = max(hmax[n], 0.05)
    hi_hi = min(hmax[n + 1], hmax[n] + 2.5)
    hin = U(hi_lo, hi_hi)
    if regime == "winter":
        warm = np.zeros(shp, bool)
    elif regime == "summer":
        warm = np.ones(shp, bool)
    else:
        warm = coin(0.4, "warm")
    snowy = coin(np.where(warm, 0.3, 0.85), own=True)
    hsn = np.where(snowy, U(0.002, 0.45), 0.0)
    tiny_snow = coin(0.05, own=True)
    hsn = np.where(tiny_snow & snowy, U(1e-5, 2e-4), hsn)   # straddles hs_min = 1e-4
    Tsfc = np.where(warm, U(-2.0, 0.0), U(-32.0, -3.0))
    Tsfc = np.where(warm & coin(0.3, "melting"), 0.0, Tsfc)
    Tbot = np.full(shp, -1.8) + U(-0.05, 0.05)
    vicen = aicen * hin
    vsnon = aicen * hsn
    eicen = np.zeros((NILYR, ny, nx)); esnon = np.zeros((NSLYR, ny, nx))
    Tsn = np.minimum(Tsfc + U(0.0, 1.0) * (Tbot - Tsfc) * 0.15, 0.0)
    qsn = -rhos * (Lfresh - cp_ice * Tsn)
    esnon[0] = qsn * vsnon / NSLYR
    for k in range(NILYR):
        zc = (k + 0.5) / NILYR
        T = Tsfc + (Tbot - Tsfc) * (0.15 + 0.85 * zc) + U(-0.3, 0.3)
        T = np.minimum(T, Tmlt[k] - U(0.005, 0.4))
        eicen[k] = qin_from_T(T, Tmlt[k]) * vicen / NILYR
    trcrn = np.zeros((5, ny, nx))
    trcrn[0] = np.where(aicen > 0, Tsfc, 0.0)
    trcrn[1] = np.where(aicen > 0, U(0, 3e7), 0.0)   # ice age (s), passive
    a = dict(aicen=aicen, trcrn=trcrn, vicen=vicen, vsnon=vsnon, eicen=eicen, esnon=esnon)
    # forcing (ice_flux.F90:316-377 defaults +- spread)
    a["flw"] = np.where(warm, U(270, 330), U(150, 290))
    a["potT"] = np.where(warm, U(270, 277), U(238, 270))
    a["Qa"] = np.where(warm, U(0.002, 0.005), U(0.0002, 0.002))
    a["rhoa"] = U(1.25, 1.4)
    a["fsnow"] = np.where(coin(0.5, "precip"), 0.0, U(0, 4e-5))
    a["fbot"] = -U(0.0, 25.0)
    a["fbot"] = np.where(coin(0.2, "fbot0"), 0.0, a["fbot"])
    a["Tbot"] = Tbot
    wind = U(1.0, 12.0)
    a["shcoef"] = 1.2e-3 * 1005.0 * a["rhoa"] * wind
    a["lhcoef"] = 1.5e-3 * Lsub * a["rhoa"] * wind
    sw = np.where(warm, U(50, 350), U(0, 60)) * (~coin(0.2, "night"))
    alb = np.where(hsn > 0.01, U(0.7, 0.85), U(0.45, 0.65))
    absd = sw * (1 - alb)
    a["fswsfc"] = absd * np.where(hsn > 0.01, 0.9, 0.3)
    pen = absd - a["fswsfc"]
    ext = np.exp(-1.4 * hin[None] * (np.arange(NILYR + 1)[:, None, None] / NILYR))
    Isw = np.zeros((NILYR, ny, nx))
    for k in range(NILYR):
        Isw[k] = pen * (ext[k] - ext[k + 1])
    Ssw = np.zeros((NSLYR, ny, nx))
    Ssw[0] = np.where(hsn > 0.01, 0.05 * absd, 0.0)
    a["Iswabs"] = Isw
    a["Sswabs"] = Ssw
    a["fswint"] = Isw.sum(axis=0) + Ssw.sum(axis=0)
    a["fswthrun"] = pen * ext[NILYR]
    for nm in ("fsurfn", "fcondtopn", "fsensn", "flatn", "fswabsn", "flwoutn", "evapn", "freshn",
               "fsaltn", "fhocnn", "meltt", "melts", "meltb", "congel", "snoice"):
        a[nm] = U(-1, 1)      # intent(out): must be overwritten / zeroed by the routine
    a["mlt_onset"] = np.where(coin(0.5, "mlt"), 0.0, 120.0)
    a["frz_onset"] = np.where(coin(0.5, "frz"), 0.0, 250.0)
    for k in a:
        a[k] = np.ascontiguousarray(a[k], np.float64)
    jj, ii = np.nonzero(a["aicen"][1:-1, 1:-1] > puny)
    indxi = np.zeros(nx * ny, np.int32); indxj = np.zeros(nx * ny, np.int32)
    icells = len(ii)
    indxi[:icells] = ii + 2
    indxj[:icells] = jj + 2
    return a, icells, indxi, indxj


def known_tsfc_inputs(a, solved, seed=20261003, perturb=0.3):
    """Inputs for thermo_vertical with calc_Tsfc = F (surface fluxes handed over by a coupler,
    ice_therm_vertical.F90:213-217).  `a` = the columns, `solved` = the same columns after a
    calc_Tsfc = T call: its fsurfn, fcondtopn, flatn and surface temperature become the inputs.
    Where the solved surface is colder than -1 C, fsurfn and fcondtopn are scaled together by up to
    +-perturb (their difference keeps its sign, so no surface energy is lost) and flatn separately;
    melting surfaces and snow layers thinner than 2 cm keep the solved fluxes (an inconsistent
    flux drives such a layer to 0 C, where the energy cannot be conserved and the reference stops)."""
    b = {k: v.copy() for k, v in a.items()}
    rng = np.random.default_rng(seed)
    shape = solved["fsurfn"].shape
    hs = np.where(a["aicen"] > 0.0, a["vsnon"] / np.where(a["aicen"] > 0.0, a["aicen"], 1.0), 0.0)
    cold = (solved["trcrn"][0] < -1.0) & ((hs == 0.0) | (hs > 0.02))
    f = np.where(cold, rng.uniform(1.0 - perturb, 1.0 + perturb, shape), 1.0)
    b["fsurfn"] = solved["fsurfn"] * f
    b["fcondtopn"] = solved["fcondtopn"] * f
    b["flatn"] = solved["flatn"] * np.where(cold, rng.uniform(1.0 - perturb, 1.0 + perturb, shape), 1.0)
    b["trcrn"][0] = solved["trcrn"][0]
    return b
