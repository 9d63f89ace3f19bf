"""The reference's own timers (ice_timers.F90) for the pure model and the drop-in build, gx1 size, N steps."""
import os, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import driver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ("dropin", "ref")
for kind in kinds:
    rd = tempfile.mkdtemp(prefix="cice_t_")
    driver.write_rundir(rd, grid="rect", npt=n, istep0=25 - n)
    env = {"CICE4_AMD_PIN": os.environ["PIN"]} if "PIN" in os.environ else None    # PIN=0: nothing page-locked
    log = driver.run(os.path.join(ROOT, "oracle", "_ref", "cice_%s_gx1" % kind), rd, env=env)
    print(kind, "gx1", n, "steps")
    print(log[log.index("Timing information"):][:700])
    shutil.rmtree(rd, ignore_errors=True)
