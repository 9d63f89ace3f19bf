"""TEST INFRASTRUCTURE ONLY -- run the whole stand-alone model (oracle/build_driver.sh: the reference's
`program icemodel`, pure or with the three drop-in modules) in a scratch directory and read its restart
dump back.  Used by tests/ and tests/golden/make_golden_step.py; never by the product.

The run directory holds only data: our own `ice_in` (the namelist pin of SURVEY.md §8(c) = the values
of the reference's input_templates/gx3/ice_in that matter for the hot path, with default initial
condition and default forcing), and for gx3 the displaced-pole grid and land mask written from the
committed fixture tests/golden/gx3_grid_kmt.npz (numbers only).

Restart record order follows `dumpfile` (source/ice_restart.F90:166-252): header (istep1, time,
time_forc), then one nx_global x ny_global big-endian fp64 record per 2-d slab.
"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NCAT, NILYR, NSLYR = 5, 4, 1

SIG_ORDER = ("stressp_1", "stressp_3", "stressp_2", "stressp_4", "stressm_1", "stressm_3", "stressm_2",
             "stressm_4", "stress12_1", "stress12_3", "stress12_2", "stress12_4")


# whole-step cases (tests/golden/make_golden_step.py mints them, tests/test_gpu_step.py replays them)
STEP_CASES = {
    # name: (cfg, grid, nx, ny, npt, istep0, overrides, stride)
    "gx3_default25": ("gx3", "gx3", 100, 116, 25, 0, None, 1),
    "gx3_default3": ("gx3", "gx3", 100, 116, 3, 22, None, 1),
    "gx3_exact3": ("gx3", "gx3", 100, 116, 3, 22,
                   {"ice_nml": dict(krdg_partic=0, krdg_redist=0, calc_Tsfc=False)}, 1),
    "gx1_default3": ("gx1", "rect", 320, 384, 3, 22, None, 5),
}
STEP_DIMS = {"gx3": (100, 116, 100, 116, 1), "gx1": (320, 384, 320, 384, 1)}


def record_names(oceanmixed_ice=True):
    """Names of the 2-d records of a restart dump, in file order (ice_restart.F90:179-252)."""
    names = []
    for n in range(1, NCAT + 1):
        names += [f"aicen_{n}", f"vicen_{n}", f"vsnon_{n}", f"Tsfc_{n}"]
    names += [f"eicen_{k}" for k in range(1, NCAT * NILYR + 1)]
    names += [f"esnon_{k}" for k in range(1, NCAT * NSLYR + 1)]
    names += ["uvel", "vvel", "scale_factor", "swvdr", "swvdf", "swidr", "swidf", "strocnxT", "strocnyT"]
    names += list(SIG_ORDER) + ["iceumask"]
    if oceanmixed_ice:
        names += ["sst", "frzmlt"]
    return names


def _nml(name, d):
    def fmt(v):
        if isinstance(v, bool):
            return ".true." if v else ".false."
        if isinstance(v, str):
            return "'%s'" % v
        if isinstance(v, (tuple, list)):
            return ", ".join(fmt(x) for x in v)
        return repr(v)
    return "&%s\n" % name + "".join("  %s = %s\n" % (k, fmt(v)) for k, v in d.items()) + "/\n"


def write_rundir(path, grid="gx3", npt=25, istep0=0, nprocs=1, overrides=None):
    """Scratch run directory: ice_in + (gx3) grid and kmt files.  `overrides` = {namelist: {key: value}}."""
    os.makedirs(os.path.join(path, "restart"), exist_ok=True)
    os.makedirs(os.path.join(path, "history"), exist_ok=True)
    nml = {
        "setup_nml": dict(days_per_year=365, year_init=1997, istep0=istep0, dt=3600.0, npt=npt, ndyn_dt=1,
                          runtype="initial", ice_ic="default", restart=False, restart_dir="./restart/",
                          restart_file="iced", pointer_file="./restart/ice.restart_file", dumpfreq="d",
                          dumpfreq_n=1, diagfreq=24, diag_type="stdout", print_global=True,
                          print_points=False, dbug=False, histfreq=("x", "x", "x", "x", "x"),
                          histfreq_n=(1, 1, 1, 1, 1), hist_avg=True, history_dir="./history/",
                          history_file="iceh", history_format="bin", write_ic=False,
                          incond_dir="./history/", incond_file="iceh_ic"),
        "grid_nml": dict(grid_format="bin", grid_type="displaced_pole" if grid == "gx3" else "rectangular",
                         grid_file="grid", kmt_file="kmt", kcatbound=0),
        "domain_nml": dict(nprocs=nprocs, processor_shape="slenderX2", distribution_type="cartesian",
                           distribution_wght="latitude", ew_boundary_type="cyclic", ns_boundary_type="open"),
        "tracer_nml": dict(tr_iage=True, restart_age=False, tr_lvl=False, restart_lvl=False, tr_pond=False,
                           restart_pond=False),
        "ice_nml": dict(kitd=1, kdyn=1, ndte=120, kstrength=1, krdg_partic=1, krdg_redist=1, mu_rdg=4,
                        advection="remap", heat_capacity=True, conduct="MU71", shortwave="default",
                        albedo_type="default", albicev=0.78, albicei=0.36, albsnowv=0.98, albsnowi=0.70,
                        ahmax=0.5, R_ice=0.0, R_pnd=0.0, R_snw=0.0, atmbndy="default", fyear_init=1997,
                        ycycle=1, atm_data_format="bin", atm_data_type="default", atm_data_dir="none",
                        calc_strair=True, calc_Tsfc=True, precip_units="mks", Tfrzpt="linear_S",
                        ustar_min=0.05, update_ocn_f=False, oceanmixed_ice=True, ocn_data_format="bin",
                        sss_data_type="default", sst_data_type="default", ocn_data_dir="none",
                        oceanmixed_file="none", restore_sst=False, trestore=180, restore_ice=False),
        "icefields_nml": dict(f_tmask=False),
    }
    for grp, d in (overrides or {}).items():
        nml[grp].update(d)
    with open(os.path.join(path, "ice_in"), "w") as f:
        for grp in ("setup_nml", "grid_nml", "domain_nml", "tracer_nml", "ice_nml", "icefields_nml"):
            f.write(_nml(grp, nml[grp]) + "\n")
    if grid == "gx3":
        z = np.load(os.path.join(ROOT, "tests", "golden", "gx3_grid_kmt.npz"))
        # direct-access files, record = one nx_global x ny_global slab (ice_grid.F90:463-521: kmt i4,
        # then ULAT, ULON, HTN, HTE, HUS, HUW, ANGLE r8), big-endian
        with open(os.path.join(path, "grid"), "wb") as f:
            for k in ("ULAT", "ULON", "HTN", "HTE", "HUS", "HUW", "ANGLE"):
                f.write(z[k].astype(">f8").tobytes())
        with open(os.path.join(path, "kmt"), "wb") as f:
            f.write(z["kmt"].astype(">i4").tobytes())
    return path


def run(exe, rundir, timeout=1200, env=None, nprocs=1):
    """Run the model (nprocs > 1: under mpiexec, the MPI builds of build_driver.sh); returns its log.  Large automatic
    arrays need an unlimited stack."""
    import shutil
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    cmd = "ulimit -s unlimited; exec %s%s" % ("%s -n %d " % (mpiexec, nprocs) if nprocs > 1 else "", os.path.abspath(exe))
    p = subprocess.run(["bash", "-c", cmd], cwd=rundir, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    log = p.stdout + p.stderr
    with open(os.path.join(rundir, "ice.log"), "w") as f:
        f.write(log)
    if p.returncode != 0:
        raise RuntimeError("model run failed (rc %d):\n%s" % (p.returncode, log[-3000:]))
    return log


def read_restart(path, nx, ny, oceanmixed_ice=True):
    """Parse a sequential unformatted big-endian dump -> (header dict, {name: (ny, nx) array})."""
    raw = open(path, "rb").read()
    pos, recs = 0, []
    while pos < len(raw):
        n = int(np.frombuffer(raw, ">i4", 1, pos)[0])
        recs.append(raw[pos + 4: pos + 4 + n])
        assert int(np.frombuffer(raw, ">i4", 1, pos + 4 + n)[0]) == n
        pos += n + 8
    h = recs[0]
    hdr = dict(istep1=int(np.frombuffer(h, ">i4", 1, 0)[0]), time=float(np.frombuffer(h, ">f8", 1, 4)[0]),
               time_forc=float(np.frombuffer(h, ">f8", 1, 12)[0]))
    names = record_names(oceanmixed_ice)
    assert len(recs) == 1 + len(names), (len(recs), len(names))
    out = {}
    for name, r in zip(names, recs[1:]):
        out[name] = np.frombuffer(r, ">f8").astype(np.float64).reshape(ny, nx)
    return hdr, out


def restart_path(rundir):
    with open(os.path.join(rundir, "restart", "ice.restart_file")) as f:
        return os.path.join(rundir, f.read().strip())


def diagnostics(log):
    """Scalars printed by runtime_diags (ice_diagnostics.F90:105): {label: [(arctic, antarctic), ...]}."""
    out = {}
    for line in log.splitlines():
        if "=" in line:
            k, _, v = line.partition("=")
            parts = v.split()
            if len(parts) == 2:
                try:
                    out.setdefault(k.strip(), []).append((float(parts[0]), float(parts[1])))
                except ValueError:
                    pass
    return out
