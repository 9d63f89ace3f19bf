"""EVP subcycle rate on a gx1-size grid by north-south boundary type: what the fold costs the loop.
usage: python scripts/tripole_rate.py [nxg nyg]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
nxg = int(sys.argv[1]) if len(sys.argv) > 1 else 320
nyg = int(sys.argv[2]) if len(sys.argv) > 2 else 384
NDTE, DT = 120, 3600.0
for name, ns in (("open", 0), ("tripole", 3), ("tripoleT", 4)):
    ctx = lib.Context()
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=8, land_rows=0), dom, ew_cyclic=True, north_ocean=(ns != 0))
    s = synth.evp_state(grid, dom, seed=8, cover="full")
    ctx.evp_init(grid, ndte=NDTE)
    if os.environ.get("SKEW_MIN_CELLS"):
        ctx.evp_set_option("skew_min_cells", int(os.environ["SKEW_MIN_CELLS"]))
    ctx.evp_upload(s); ctx.evp_prepare(DT)
    for _ in range(10 if nxg * nyg < 1000000 else 1):
        ctx.evp_subcycles(1, NDTE)
    ctx.sync(); t0 = time.perf_counter()
    n = 50 if nxg * nyg < 1000000 else 3
    for _ in range(n):
        ctx.evp_subcycles(1, NDTE)
    ctx.sync(); t = (time.perf_counter() - t0) / n
    print("%-9s %8.2f us per subcycle  (%.0f subcycles/s; resident %d fused %d skew %d)" %
          (name, 1e6 * t / NDTE, NDTE / t, ctx.evp_get_info("resident"), ctx.evp_get_info("fused"), ctx.evp_get_info("skew")), flush=True)
    ctx.close()
