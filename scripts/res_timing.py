"""gx1 size: one launch per pair of subcycles against the whole loop in one launch (RES_LIB=<other .so>: A/B builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
if os.environ.get("RES_LIB"):          # A/B against another build of the library
    lib.LIBPATH = os.environ["RES_LIB"]
nxg, nyg, ndte = 320, 384, 120
ctx = lib.Context()
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.0, seed=1)
grid = synth.block_fields(gg, dom, ew_cyclic=True)
s = synth.evp_state(grid, dom, seed=1, cover="full")
def run(**opts):
    ctx.evp_init(grid, ndte=ndte)
    for k, v in opts.items():
        try:
            ctx.evp_set_option(k, v)
        except lib.CiceError:
            if v:
                return float("nan")
    ctx.evp_upload({k: v.copy() for k, v in s.items()}); ctx.evp_prepare(3600.0)
    ts = [ctx.evp_subcycles(1, ndte, timed=True) for _ in range(6)]
    return min(ts[1:]) * 1e3 / ndte
variants = [("launch per pair", dict(resident=0))]
for W in (int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "11,12".split(","))):
    variants.append(("W=%d whole loop in one launch" % W, dict(resident=2, resident_waves=W)))
for name, o in variants:
    print("%-40s %.3f us per subcycle" % (name, run(**o)), flush=True)
