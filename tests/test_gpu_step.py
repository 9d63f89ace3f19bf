"""Whole-driver drop-in run: the reference's own `program icemodel`
(drivers/cice4/CICE.F90:64-94 -> ice_step, CICE_RunMod.F90:164-242 -> step_therm1 :260-598 and
step_dynamics, source/ice_step_mod.F90:538-745 -> evp, transport_remap) compiled UNCHANGED, linked with the four drop-in
modules (cice4_amd/fortran/{ice_dyn_evp,ice_therm_vertical,ice_transport_driver,rccl/ice_boundary}.F90) and
libcice4_amd.so, runs its
time loop on the GPU box; its restart dump (source/ice_restart.F90:74-256) is compared with the dump of
the pure serial reference (tests/golden/step_*.npz, minted by tests/golden/make_golden_step.py).

Bound: BIT FOR BIT on all 69 records of the dump, with the default options too -- the device evaluates
exp() (ice_strength, saturation humidity) with glibc's own algorithm (cice4_amd/csrc/libm_exact.h).  With an
exp() that is merely accurate to an ulp the same runs differ from the reference by 5e-11 after 3 steps, 5e-7
after 4 and 6e-2 after 6 (stresses near the northern edge of the gx3 Arctic cap: profiles/r02_step_diag_*.log)."""
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import driver  # noqa: E402
from conftest import TOL_EXP  # noqa: E402
CASES = driver.STEP_CASES

GOLD = os.path.join(ROOT, "tests", "golden")
COMPARED = (["uvel", "vvel", "strocnxT", "strocnyT", "iceumask", "sst", "frzmlt", "scale_factor"]
            + list(driver.SIG_ORDER)
            + [f"{v}_{n}" for n in range(1, 6) for v in ("aicen", "vicen", "vsnon", "Tsfc")]
            + [f"eicen_{k}" for k in range(1, 21)] + [f"esnon_{k}" for k in range(1, 6)])


def _run_case(kind, name, extra_env=None):
    cfg, grid, nx, ny, npt, istep0, over, stride = CASES[name]
    exe = os.path.join(ROOT, "oracle", "_ref", "cice_%s_%s" % (kind, cfg))
    if not os.path.exists(exe):
        pytest.skip("%s not built (oracle/build_driver.sh needs /root/reference)" % exe)
    rd = tempfile.mkdtemp(prefix="cice_run_")
    try:
        driver.write_rundir(rd, grid=grid, npt=npt, istep0=istep0, overrides=over)
        log = driver.run(exe, rd, env=extra_env)
        hdr, rec = driver.read_restart(driver.restart_path(rd), nx, ny)
    finally:
        shutil.rmtree(rd, ignore_errors=True)
    gold = np.load(os.path.join(GOLD, "step_%s.npz" % name))
    assert hdr["istep1"] == int(gold["istep1"]) and hdr["time"] == float(gold["time"])
    return rec, gold, stride, log


def _compare(rec, gold, stride, tol):
    worst = ("", 0.0)
    for k in COMPARED:
        a, g = rec[k][::stride, ::stride], gold[k]
        if tol == 0.0:
            assert np.array_equal(a, g), (k, np.abs(a - g).max())
        else:
            err = np.abs(a - g).max() / max(np.abs(g).max(), 1e-300)
            if err > worst[1]:
                worst = (k, err)
            assert err <= tol, (k, err)
        if stride > 1:      # full-field statistics: every point takes part
            s = np.array([rec[k].sum(), (rec[k] ** 2).sum(), rec[k].min(), rec[k].max()])
            gs = gold["stats_" + k]
            scale = max(np.abs(gs[1]), 1e-300) ** 0.5
            assert np.all(np.abs(s[[0, 2, 3]] - gs[[0, 2, 3]]) <= max(tol, 1e-13) * max(scale, np.abs(gs[0]))), (k, s, gs)
    return worst


@pytest.mark.parametrize("name", ["gx3_default3", "gx3_exact3", "gx3_default25"])
def test_pure_reference_reproduces_its_golden_dump(name):
    """The harness itself: the pure reference run here gives the committed numbers again, bit for bit
    (build container only -- another host may select another libm `exp`)."""
    if not os.path.isdir("/root/reference/source"):
        pytest.skip("build container only")
    rec, gold, stride, _ = _run_case("ref", name)
    _compare(rec, gold, stride, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,tol", [("gx3_exact3", 0.0), ("gx3_default3", TOL_EXP), ("gx3_default25", TOL_EXP),
                                      ("gx1_default3", TOL_EXP)])
def test_reference_step_loop_with_dropin_modules(name, tol):
    # (gx3_default25 also runs with the evp -> transport chain: same dump; gx3_exact3 and gx1_default3 with evp's io state
    #  kept on the device between the steps -- the reference's driver is a caller of which both statements hold)
    extra = {"CICE4_AMD_CHAIN": "1"} if name == "gx3_default25" else {"CICE4_AMD_KEEP_STATE": "2"} if name in ("gx3_exact3", "gx1_default3") else {}
    rec, gold, stride, log = _run_case("dropin", name, {"CICE4_AMD_STATS": "1", **extra})
    assert "EVP dynamics on the GPU" in log and "Incremental remapping on the GPU" in log
    assert ("evp keeps uvel, vvel, the stresses and iceumask on the device" in log) == ("CICE4_AMD_KEEP_STATE" in extra)
    assert "on 1 block(s): 1 kernel launch(es)" in log
    assert ("transport_remap takes its state from the device after evp" in log) == (name == "gx3_default25")
    worst = _compare(rec, gold, stride, tol)
    print("whole-driver drop-in", name, "worst field-relative difference", worst)


@pytest.mark.gpu
def test_restart_round_trip_with_dropin_modules():
    """Checkpoint / resume through the reference's own dumpfile / restartfile (source/ice_restart.F90:74-256, :265)
    with the GPU modules in the loop: (A) 50 steps without interruption, dumps after 25 and 49 steps; (B) a new
    process restarted from A's first dump (runtype = 'continue': velocities, 12 stresses, iceumask, state read back
    into the module arrays the drop-in evp uploads).  B's dump must equal A's second dump bit for bit, and both
    must equal the PURE reference run on this host (cice_ref_gx3, 50 steps)."""
    import glob
    exe = {k: os.path.join(ROOT, "oracle", "_ref", "cice_%s_gx3" % k) for k in ("ref", "dropin")}
    for e in exe.values():
        if not os.path.exists(e):
            pytest.skip("%s not built" % e)
    dirs = {k: tempfile.mkdtemp(prefix="cice_rs_%s_" % k) for k in ("A", "B", "R")}
    try:
        dumps = {}
        for tag, kind in (("A", "dropin"), ("R", "ref")):
            driver.write_rundir(dirs[tag], npt=50)
            driver.run(exe[kind], dirs[tag])
            dumps[tag] = sorted(f for f in glob.glob(dirs[tag] + "/restart/iced.1997*"))
            assert [os.path.basename(f) for f in dumps[tag]] == ["iced.1997-01-02-00000", "iced.1997-01-03-00000"]
        driver.write_rundir(dirs["B"], npt=30, overrides={"setup_nml": dict(runtype="continue", restart=True)})
        for f in glob.glob(dirs["A"] + "/restart/iced*1997-01-02*"):      # the dump and its ice-age companion
            shutil.copy(f, dirs["B"] + "/restart/")
        with open(dirs["B"] + "/restart/ice.restart_file", "w") as f:
            f.write("./restart/iced.1997-01-02-00000\n")
        # (the restarted run also keeps evp's io state on the device: what the restart reader put into the module arrays
        #  travels up with the first call)
        log = driver.run(exe["dropin"], dirs["B"], env={"CICE4_AMD_KEEP_STATE": "2"})
        assert "Using restart dump" in log and "EVP dynamics on the GPU" in log and "evp keeps uvel, vvel" in log
        a2 = driver.read_restart(dumps["A"][1], 100, 116)
        b2 = driver.read_restart(dirs["B"] + "/restart/iced.1997-01-03-00000", 100, 116)
        r2 = driver.read_restart(dumps["R"][1], 100, 116)
        assert a2[0] == b2[0] == r2[0] and a2[0]["istep1"] == 48
        for k in a2[1]:
            assert np.array_equal(a2[1][k], b2[1][k]), ("restarted vs uninterrupted", k)
            if TOL_EXP == 0.0:
                assert np.array_equal(a2[1][k], r2[1][k]), ("drop-in vs pure reference, 49 steps", k)
        assert np.abs(a2[1]["uvel"]).max() > 0.05
    finally:
        for d in dirs.values():
            shutil.rmtree(d, ignore_errors=True)


@pytest.mark.gpu
def test_whole_model_with_upwind_advection():
    """namelist advection = 'upwind' (ice_step_mod.F90:581-582 calls transport_upwind instead of transport_remap): the
    drop-in transport module runs that scheme on the GPU too.  25 steps (the dump is written at the day boundary) of the
    whole model, pure reference and drop-in modules on this host, all records of the restart dump."""
    exe = {k: os.path.join(ROOT, "oracle", "_ref", "cice_%s_gx3" % k) for k in ("ref", "dropin")}
    for e in exe.values():
        if not os.path.exists(e):
            pytest.skip("%s not built" % e)
    dirs = {k: tempfile.mkdtemp(prefix="cice_upw_%s_" % k) for k in ("ref", "dropin")}
    try:
        rec = {}
        for kind in ("ref", "dropin"):
            driver.write_rundir(dirs[kind], npt=25, overrides={"ice_nml": dict(advection="upwind")})
            log = driver.run(exe[kind], dirs[kind])
            rec[kind] = driver.read_restart(driver.restart_path(dirs[kind]), 100, 116)
        assert "EVP dynamics on the GPU" in log and "Incremental remapping on the GPU" not in log
        assert rec["ref"][0] == rec["dropin"][0]
        for k in rec["ref"][1]:
            a, g = rec["dropin"][1][k], rec["ref"][1][k]
            if TOL_EXP == 0.0:
                assert np.array_equal(a, g), (k, np.abs(a - g).max())
            else:
                assert np.abs(a - g).max() <= TOL_EXP * max(np.abs(g).max(), 1e-300), k
        assert np.abs(rec["ref"][1]["uvel"]).max() > 0.05
    finally:
        for d in dirs.values():
            shutil.rmtree(d, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ns,ocean_at_fold", [("tripole", False), ("tripoleT", False), ("tripole", True), ("tripoleT", True)])
def test_whole_model_with_a_tripole_north_boundary(ns, ocean_at_fold):
    """namelist ns_boundary_type = 'tripole' / 'tripoleT' on the gx3 grid: every ice_HaloUpdate of the model (grid set-up
    with extrapolation, bound_state, the scalar and vector fields of the dynamics and the transport at their four
    locations) goes through the fold of the drop-in boundary module, evp(dt) runs the one-launch loop with the fold
    inside.  25 steps, pure reference and drop-in modules on this host, all records of the restart dump.
    ocean_at_fold: the gx3 land mask has land along the northern edge, so the fold moves nothing; with the last eight rows
    of kmt opened the initial ice reaches the fold and drifts across it (|u| ~ 0.1 m/s in the top rows): the symmetric
    averages of the degenerate row, the mirrored ghost row, the transport through the fold all carry real values."""
    exe = {k: os.path.join(ROOT, "oracle", "_ref", "cice_%s_gx3" % k) for k in ("ref", "dropin")}
    for e in exe.values():
        if not os.path.exists(e):
            pytest.skip("%s not built" % e)
    dirs = {k: tempfile.mkdtemp(prefix="cice_tri_%s_" % k) for k in ("ref", "dropin")}
    try:
        rec = {}
        for kind in ("ref", "dropin"):
            driver.write_rundir(dirs[kind], npt=25, overrides={"domain_nml": dict(ns_boundary_type=ns)})
            if ocean_at_fold:
                kmt = np.load(os.path.join(GOLD, "gx3_grid_kmt.npz"))["kmt"].copy()
                kmt[-8:, :] = np.maximum(kmt[-8:, :], 1)
                with open(os.path.join(dirs[kind], "kmt"), "wb") as f:
                    f.write(kmt.astype(">i4").tobytes())
            log = driver.run(exe[kind], dirs[kind])
            rec[kind] = driver.read_restart(driver.restart_path(dirs[kind]), 100, 116)
        assert "EVP dynamics on the GPU" in log
        assert rec["ref"][0] == rec["dropin"][0]
        if ocean_at_fold:
            assert np.abs(rec["ref"][1]["uvel"][-3:]).max() > 0.01, "no ice moves at the fold"
        for k in rec["ref"][1]:
            a, g = rec["dropin"][1][k], rec["ref"][1][k]
            if TOL_EXP == 0.0:
                assert np.array_equal(a, g), (ns, k, np.abs(a - g).max())
            else:
                assert np.abs(a - g).max() <= TOL_EXP * max(np.abs(g).max(), 1e-300), k
        assert np.abs(rec["ref"][1]["uvel"]).max() > 0.05
    finally:
        for d in dirs.values():
            shutil.rmtree(d, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,nprocs", [("gx3b4", 2), ("gx3b4", 4), ("gx3s2", 2), ("gx3b4", 1)])
def test_whole_model_as_an_mpi_job_on_one_gpu(cfg, nprocs):
    """The reference's whole model in its MPI build (mpi/ modules, MPICH) with the four drop-in modules, `mpiexec -n P` on
    the real gx3 grid in 2 x 2 blocks: the P tasks share the one GPU and exchange through the shared-memory link
    (CICE4_AMD_LINK=shm; RCCL refuses two ranks on one device).  Block distribution by the reference's create_distribution,
    ghost cells between tasks in every ice_HaloUpdate of the model and after every EVP subcycle, transport and
    thermodynamics per task, the restart dump gathered by the reference's own MPI gather: the dump after 25 steps equals
    the pure serial reference's (which the pure MPI reference reproduces bit for bit on the CPU).
    cfg gx3s2: two full-width slabs, one per task -- the drop-in dynamics connect the neighbouring task's exchange buffers
    (IPC handles over MPI) and evp(dt) subcycles in ONE launch per task with device-initiated exchange; since round 5 the
    same with gx3b4 on four tasks: one 50 x 58 block per task in a 2 x 2 layout, every task connected to the three others
    (its eastern and western neighbour are the same task, so are the two diagonal ones)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cice_dropinmpi_%s" % cfg)
    if not os.path.exists(exe):
        pytest.skip("%s not built (MPI=1 DROPIN=1 oracle/build_driver.sh)" % exe)
    rd = tempfile.mkdtemp(prefix="cice_mpi_")
    try:
        driver.write_rundir(rd, npt=25, nprocs=nprocs)
        # (gx3s2 and the four-task job also keep evp's io state on the device between the steps: CICE4_AMD_KEEP_STATE, same dump)
        keep = {"CICE4_AMD_KEEP_STATE": "2"} if (cfg == "gx3s2" or nprocs == 4) else {}
        log = driver.run(exe, rd, env={"CICE4_AMD_LINK": "shm", "CICE4_AMD_PEER_SHARE": str(nprocs), "CICE4_AMD_STATS": "1", **keep},
                         nprocs=nprocs)
        assert ("evp keeps uvel, vvel, the stresses and iceumask on the device" in log) == bool(keep)
        hdr, rec = driver.read_restart(driver.restart_path(rd), 100, 116)
    finally:
        shutil.rmtree(rd, ignore_errors=True)
    assert "EVP dynamics on the GPU" in log and "Incremental remapping on the GPU" in log
    # one block per task -- two full-width slabs, or (round 5) the 2 x 2 cartesian layout of comp_ice:34-46 with four tasks,
    # east-west and diagonal neighbours: the whole subcycle loop is one launch per task
    # ... and with two tasks of two blocks each (tiles numbered block by block, ghost cells between a task's own blocks
    # forwarded on the device, those of other tasks' blocks through the mapped buffers)
    one_launch = nprocs > 1
    assert ("EVP subcycling as one launch per task" in log) == one_launch
    if one_launch:
        assert "on %d block(s): 1 kernel launch(es)" % (4 // nprocs if cfg == "gx3b4" else 1) in log, log[-3000:]
    if nprocs == 1:     # all 2 x 2 blocks on one task: the one-launch loop on several blocks (round 4)
        assert "on 4 block(s): 1 kernel launch(es)" in log, log[-3000:]
    assert "resident EVP loop timed out" not in log
    gold = np.load(os.path.join(GOLD, "step_gx3_default25.npz"))
    assert hdr["istep1"] == int(gold["istep1"]) and hdr["time"] == float(gold["time"])
    worst = _compare(rec, gold, 1, TOL_EXP)
    print("whole model (%s), MPI job of %d tasks on one GPU: worst field-relative difference" % (cfg, nprocs), worst)


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nprocs,cfg", [("tripole", 2, "gx3b4"), ("tripole", 4, "gx3b4"), ("tripoleT", 4, "gx3b4"),
                                           ("tripole", 2, "gx3s2"), ("tripoleT", 2, "gx3s2")])
def test_whole_model_as_an_mpi_job_across_a_tripole_fold(ns, nprocs, cfg):
    """The same MPI job with a tripole north boundary and ocean up to the fold: the top rows of the two blocks of the top
    block row travel between tasks into every task's fold buffer (one more message pair per ice_HaloUpdate), the fold is
    applied after every EVP subcycle.  Against the pure SERIAL reference (one block) with the same namelist and mask.
    cfg gx3s2 (round 5): two full-width slabs -- every partner across the pole lies on the task of the top slab, which runs
    the cross-task one-launch loop with the fold inside; the task below the plain cross-task loop: ONE launch per task."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cice_dropinmpi_%s" % cfg)
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "cice_ref_gx3")
    for e in (exe, ref_exe):
        if not os.path.exists(e):
            pytest.skip("%s not built" % e)
    kmt = np.load(os.path.join(GOLD, "gx3_grid_kmt.npz"))["kmt"].copy()
    kmt[-8:, :] = np.maximum(kmt[-8:, :], 1)
    rec = {}
    for kind, e, n in (("ref", ref_exe, 1), ("mpi", exe, nprocs)):
        rd = tempfile.mkdtemp(prefix="cice_mpitri_")
        try:
            driver.write_rundir(rd, npt=25, nprocs=n, overrides={"domain_nml": dict(ns_boundary_type=ns)})
            with open(os.path.join(rd, "kmt"), "wb") as f:
                f.write(kmt.astype(">i4").tobytes())
            log = driver.run(e, rd, env={"CICE4_AMD_LINK": "shm", "CICE4_AMD_PEER_SHARE": str(n), "CICE4_AMD_STATS": "1"}, nprocs=n)
            if kind == "mpi":
                assert ("EVP subcycling as one launch per task" in log) == (cfg == "gx3s2"), log[-3000:]
                assert "resident EVP loop timed out" not in log
                if cfg == "gx3s2":
                    assert "on 1 block(s): 1 kernel launch(es)" in log, log[-3000:]
            rec[kind] = driver.read_restart(driver.restart_path(rd), 100, 116)
        finally:
            shutil.rmtree(rd, ignore_errors=True)
    assert rec["ref"][0] == rec["mpi"][0]
    assert np.abs(rec["ref"][1]["uvel"][-3:]).max() > 0.01
    for k in rec["ref"][1]:
        a, g = rec["mpi"][1][k], rec["ref"][1][k]
        if TOL_EXP == 0.0:
            assert np.array_equal(a, g), (ns, nprocs, k, np.abs(a - g).max())
        else:
            assert np.abs(a - g).max() <= TOL_EXP * max(np.abs(g).max(), 1e-300), k


@pytest.mark.gpu
def test_whole_model_on_120_blocks_with_eliminated_land_blocks():
    """The whole model on the real gx3 grid cut into 10 x 12 blocks of 10 x 10 cells (max_blocks = 120), the four
    all-land blocks eliminated by the reference's own create_distribution: multi-block EVP (per-subcycle halo updates
    between blocks), transport, thermodynamics and every ice_HaloUpdate of the model through the drop-in modules, 6
    steps, against the pure reference built for the same block layout and run on this host: restart dumps bit for bit."""
    import glob
    exe = {k: os.path.join(ROOT, "oracle", "_ref", "cice_%s_gx3e" % k) for k in ("ref", "dropin")}
    for e in exe.values():
        if not os.path.exists(e):
            pytest.skip("%s not built" % e)
    dirs = {k: tempfile.mkdtemp(prefix="cice_e_%s_" % k) for k in exe}
    try:
        rec = {}
        for kind in exe:
            driver.write_rundir(dirs[kind], npt=6, istep0=19)
            log = driver.run(exe[kind], dirs[kind], env={"CICE4_AMD_STATS": "1", "CICE4_AMD_CHAIN": "1"})
            if kind == "dropin":
                assert "EVP dynamics on the GPU" in log and "Incremental remapping on the GPU" in log
                # round 4: 116 blocks on one rank subcycle in ONE launch (was: ndte launches + halo kernels), and the
                # transport takes its state from the device after evp (cice_transport_chain)
                assert "on 116 block(s): 1 kernel launch(es)" in log, log[-3000:]
                assert "transport_remap takes its state from the device after evp" in log
            rec[kind] = driver.read_restart(driver.restart_path(dirs[kind]), 100, 116)
        assert rec["ref"][0] == rec["dropin"][0]
        for k in rec["ref"][1]:
            a, b = rec["dropin"][1][k], rec["ref"][1][k]
            if TOL_EXP == 0.0:
                assert np.array_equal(a, b), (k, np.abs(a - b).max())
            else:
                assert np.abs(a - b).max() <= TOL_EXP * max(np.abs(b).max(), 1e-300), k
        u = rec["ref"][1]["uvel"]
        assert (np.abs(u) > 1e29).any()                    # eliminated blocks are written with the reference's spval
        assert np.abs(u[np.abs(u) < 1e29]).max() > 0.05
    finally:
        for d in dirs.values():
            shutil.rmtree(d, ignore_errors=True)
