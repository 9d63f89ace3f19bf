"""What a step of the one-launch loop costs the host beyond its kernel (gx1, full cover): cice_evp_subcycles(1, ndte) with and
without the event bracket, against the kernel time of the bracket.  usage: python scripts/step_overhead.py [library.so]
(measured at the end of round 5: 584.6 us per step around a 573-us kernel; without the read-back of the abort word 582.9: the rest is
the latency of one launch and of one synchronisation)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
if len(sys.argv) > 1:
    lib.LIBPATH = os.path.abspath(sys.argv[1])
nxg, nyg, ndte, DT = 320, 384, 120, 3600.0
ctx = lib.Context(); ctx.sync()
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
s = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_upload(s); ctx.evp_prepare(DT); ctx.sync()
for _ in range(50):
    ctx.evp_subcycles(1, ndte)
ctx.sync()
for timed in (False, True, False, True):
    n = 400
    dev = 0.0
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n):
        dev += ctx.evp_subcycles(1, ndte, timed=timed)
    ctx.sync(); t = (time.perf_counter() - t0) / n * 1e6
    print("timed=%s: %.1f us per step (%.3f us per subcycle = %.0f subcycles/s)%s" % (timed, t, t / ndte, ndte / t * 1e6,
          "; kernel bracket %.1f us" % (dev / n * 1e3) if timed else ""), flush=True)
