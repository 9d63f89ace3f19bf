"""Which columns / rows / fields of a K-subcycle sweep differ from one launch per subcycle?  usage: skew_debug.py nxg nyg [ew]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib, synth
nxg, nyg = int(sys.argv[1]), int(sys.argv[2])
ew = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=0)
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg)
grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1))
s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
def run(ndte, **opts):
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
    for k, v in opts.items():
        ctx.evp_set_option(k, v)
    ctx.evp(3600.0, sg)
    return sg
for ndte in (4, 8, 120):
    ref = run(ndte, fuse=0, resident=0, skew=0)
    for K in (4, 2, 3, 5, 6, 8):
        if ndte < K:
            continue
        got = run(ndte, resident=0, skew=1, skew_min_cells=0, skew_levels=K, use_graph=0)
        bad = {}
        for k in ("uvel", "vvel", "stressp_1", "stress12_4", "divu"):
            d = np.argwhere(got[k][0] != ref[k][0])
            if len(d):
                bad[k] = (len(d), sorted(set(d[:, 1].tolist()))[:12], sorted(set(d[:, 0].tolist()))[:6])
        print(f"ndte {ndte} K {K} strips {ctx.evp_get_info('skew_strips')}:", "OK" if not bad else bad, flush=True)
