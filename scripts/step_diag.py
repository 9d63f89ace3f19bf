"""Diagnostic (GPU box): growth of the drop-in-vs-pure-reference difference with the number of steps."""
import os, sys, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import driver

def run(kind, cfg, grid, nx, ny, npt, istep0, over):
    rd = tempfile.mkdtemp(prefix="cice_diag_")
    driver.write_rundir(rd, grid=grid, npt=npt, istep0=istep0, overrides=over)
    driver.run(os.path.join(ROOT, "oracle", "_ref", "cice_%s_%s" % (kind, cfg)), rd)
    h, r = driver.read_restart(driver.restart_path(rd), nx, ny)
    shutil.rmtree(rd, ignore_errors=True)
    return r

exact = {"ice_nml": dict(krdg_partic=0, krdg_redist=0, calc_Tsfc=False)}
for label, over in (("default", None), ("exact", exact)):
    for n in (3, 4, 6, 9, 13, 25):
        a = run("ref", "gx3", "gx3", 100, 116, n, 25 - n, over)
        b = run("dropin", "gx3", "gx3", 100, 116, n, 25 - n, over)
        errs = {}
        for k in a:
            errs[k] = np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-300)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
        k0 = top[0][0]
        j, i = np.unravel_index(np.abs(a[k0] - b[k0]).argmax(), a[k0].shape)
        nd = sum(1 for k in a if errs[k] > 0)
        print(label, "steps", n, "fields differing", nd, "worst", [(k, float("%.2e" % v)) for k, v in top], "at", (i, j), flush=True)
