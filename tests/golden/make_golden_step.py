#!/usr/bin/env python3
"""Mint whole-step golden vectors FROM THE REFERENCE'S OWN STAND-ALONE MODEL (program icemodel built by
oracle/build_driver.sh from /root/reference, every module the reference's, serial backend): the restart
dump (`dumpfile`, source/ice_restart.F90:74-256) after N calls of ice_step
(drivers/cice4/CICE_RunMod.F90:164-242: prep_radiation, step_therm1, step_therm2, step_dynamics,
step_radiation, coupling_prep).  Only numbers are stored.

  step_gx3_default25.npz  gx3 real grid + land mask, default IC / default forcing, namelist pin of
                          SURVEY.md §8(c), 25 steps (the first dump the calendar allows from istep0=0)
  step_gx3_default3.npz   the same, istep0=22 -> the dump falls after 3 steps
  step_gx3_exact3.npz     3 steps with the transcendental-free options (krdg_partic=0, krdg_redist=0,
                          calc_Tsfc=F): the GPU drop-in must reproduce these BIT FOR BIT
  step_gx1_default3.npz   gx1-size 320x384 built-in rectangular grid, full ice cover, 3 steps; every 5th
                          point of every record + per-record sum / sum of squares / min / max

Run from the repo root in the build container:  python tests/golden/make_golden_step.py
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import driver  # noqa: E402

CASES, DIMS = driver.STEP_CASES, driver.STEP_DIMS


def meta(case):
    fc = subprocess.run(["/opt/rocm/bin/amdflang", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    return np.array([f"reference: COSIMA/cice4 @ /root/reference, program icemodel (drivers/cice4, serial/); "
                     f"compiler: {fc}; flags: -O2 -fdefault-real-8 -ffp-contract=off (oracle/build_driver.sh); "
                     f"case {case}: {CASES[case][2:]}; generator: tests/golden/make_golden_step.py"])


def stats(a):
    return np.array([a.sum(), (a * a).sum(), a.min(), a.max()])


def main():
    for name, (cfg, grid, nx, ny, npt, istep0, over, stride) in CASES.items():
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_driver.sh"), cfg,
                               *[str(d) for d in DIMS[cfg]]])
        rd = os.path.join(ROOT, "oracle", "_ref", "run_golden_" + name)
        shutil.rmtree(rd, ignore_errors=True)
        driver.write_rundir(rd, grid=grid, npt=npt, istep0=istep0, overrides=over)
        driver.run(os.path.join(ROOT, "oracle", "_ref", "cice_ref_" + cfg), rd)
        hdr, rec = driver.read_restart(driver.restart_path(rd), nx, ny)
        assert hdr["istep1"] == istep0 + npt - 1, hdr
        data = {"meta": meta(name), "istep1": np.array(hdr["istep1"]), "time": np.array(hdr["time"]),
                "stride": np.array(stride)}
        for k, v in rec.items():
            data[k] = v[::stride, ::stride].copy()
            if stride > 1:
                data["stats_" + k] = stats(v)
        np.savez_compressed(os.path.join(HERE, "step_%s.npz" % name), **data)
        shutil.rmtree(rd, ignore_errors=True)
        print(name, "ice u max", np.abs(rec["uvel"]).max(), "cells with ice", int((rec["aicen_1"] > 0).sum()))


if __name__ == "__main__":
    main()
