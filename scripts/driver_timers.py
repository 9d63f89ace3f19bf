"""The reference's own timers (ice_timers.F90) for the pure model and the drop-in build, N steps.
usage: driver_timers.py [N [dropin,ref [gx1|gx3]]]   (gx1: 320x384 rectangular, full cover; gx3: the real grid, default IC)"""
import os, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import driver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ("dropin", "ref")
cfg = sys.argv[3] if len(sys.argv) > 3 else "gx1"
for kind in kinds:
    rd = tempfile.mkdtemp(prefix="cice_t_")
    driver.write_rundir(rd, grid="rect" if cfg == "gx1" else "gx3", npt=n, istep0=max(0, 25 - n))
    env = {"CICE4_AMD_PIN": os.environ["PIN"]} if "PIN" in os.environ else None    # PIN=0: nothing page-locked
    log = driver.run(os.path.join(ROOT, "oracle", "_ref", "cice_%s_%s" % (kind, cfg)), rd, env=env)
    print(kind, cfg, n, "steps")
    print(log[log.index("Timing information"):][:700])
    shutil.rmtree(rd, ignore_errors=True)
