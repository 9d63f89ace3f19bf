"""Soak of the K-level sweep (k_subcycle_skew): REPS whole evp(dt) calls (ndte subcycles, ndte/K sweeps each) from one
state on a grid of more than a million cells; every call has to return the bits of the first one -- which are checked
against one launch per subcycle.  A hand-off through LDS, a forwarded east-west ghost or a row prefetched one step early
that once in a while let a stale value through would show here (one wrong ulp grows to 1e-2 within a step).
usage: python scripts/soak_sweep.py [reps] [nxg] [nyg]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nxg = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
nyg = int(sys.argv[3]) if len(sys.argv) > 3 else 600
NDTE, DT = 240, 3600.0
KEYS = ("uvel", "vvel") + synth.SIG_NAMES
ctx = lib.Context()
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=8), dom, ew_cyclic=True)
s = synth.evp_state(grid, dom, seed=8, cover="patchy")
ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
for k, v in (("skew", 0), ("resident", 0), ("fuse", 0)):
    ctx.evp_set_option(k, v)
ref = {k: v.copy() for k, v in s.items()}
ctx.evp(DT, ref)
assert np.abs(ref["uvel"]).max() > 0.01
t0 = time.time()
done = 0
for K in (4, 3):
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("skew", 1); ctx.evp_set_option("skew_levels", K); ctx.evp_set_option("resident", 0)
    assert ctx.evp_get_info("skew") == 1
    ctx.evp_upload({k: v.copy() for k, v in s.items()})
    out = {k: np.empty_like(v) for k, v in s.items()}
    for rep in range(reps if K == 4 else reps // 4):
        ctx.evp_prepare(DT); ctx.evp_subcycles(1, NDTE); ctx.evp_finish()
        ctx.evp_download(out)
        for k in KEYS:
            if not np.array_equal(out[k], ref[k]):
                bad = np.argwhere(out[k] != ref[k])
                raise SystemExit("SOAK FAILED K=%d rep %d field %s: %d cells differ, first %s" % (K, rep, k, len(bad), bad[:4].tolist()))
        ctx.evp_upload({k: v.copy() for k, v in s.items()})
        done += 1
        if done % 50 == 0:
            print("  %d calls, %.0f s" % (done, time.time() - t0), flush=True)
print("SOAK-OK %d evp(dt) calls = %d sweeps of K levels on %d x %d, %d subcycles each, %.0f s" % (done, done * NDTE // 4, nxg, nyg, NDTE, time.time() - t0))
