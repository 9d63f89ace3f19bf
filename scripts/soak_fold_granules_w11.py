"""The one-launch granule loop under a tripole fold at 11 wavefronts per workgroup, many evp(dt) calls from one state against the
per-subcycle path: the case that showed the store-data hazard of profiles/r05_resident_granules.txt section 11 (18 of 40 calls
wrong at ndte = 32 before st_gran kept its registers live behind the store; none since).
usage: soak_fold_granules_w11.py <library.so> <ndte,ndte,...>   (scripts/gpu_r5_24.sh)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
lib.LIBPATH = os.path.abspath(sys.argv[1])
nxg, nyg, ns, W, reps = 96, 70, 3, 11, 30
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=nxg + nyg, land_rows=0)
grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True)
s = synth.evp_state(grid, dom, seed=nxg, cover="patchy")
def run(ndte, **opts):
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, krdg_partic=0, krdg_redist=0)
    for k, v in opts.items():
        ctx.evp_set_option(k, v)
    ctx.evp(3600.0, sg)
    return sg
for ndte in [int(x) for x in sys.argv[2].split(",")]:
    ref = run(ndte, resident=0, resident_fold=0)
    nbad = 0
    for rep in range(reps):
        got = run(ndte, resident=2, resident_fold=1, resident_waves=W, resident_granules=1)
        msgs = []
        for key in ("uvel", "vvel", "stressp_1", "stressm_1", "stress12_1", "stressp_2", "stressp_3", "stressp_4"):
            d = np.argwhere(got[key][0] != ref[key][0])
            if len(d):
                rows = sorted(set(d[:, 0].tolist()))
                msgs.append("%s n=%d rows=%s cols=%s" % (key, len(d), rows[:6], sorted(set(d[:, 1].tolist()))[:24]))
        if msgs:
            nbad += 1
            if nbad <= 4:
                print("ndte", ndte, "rep", rep, " | ".join(msgs), flush=True)
    print("ndte", ndte, "bad", nbad, "of", reps, flush=True)
