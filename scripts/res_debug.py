import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
nxg, nyg, ew, ns = (int(x) for x in sys.argv[1:5])
ctx = lib.Context()
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=ns)
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg)
grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1))
s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
for ndte in (2, 3, 7, 120):
    for W in (4, 6, 8, 11, 12):
        sg = {k: v.copy() for k, v in s.items()}
        ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
        ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_waves", W)
        r0 = ctx.evp_get_info("resident")
        ctx.evp(3600.0, sg)
        print("ndte", ndte, "W", W, "resident before/after", r0, ctx.evp_get_info("resident"), flush=True)
