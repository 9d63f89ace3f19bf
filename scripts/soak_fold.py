"""Soak of the one-launch loop with a tripole fold inside: REPS whole evp(dt) calls from one state on a gx1-size grid, each
the bits of the per-subcycle path (launch per subcycle + halo update with the fold).
usage: python scripts/soak_fold.py [reps] [boundaries, default tripole,tripoleT; "open" = no fold: the plain granule loop]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nxg, nyg, NDTE, DT = 320, 384, 120, 3600.0
KEYS = ("uvel", "vvel") + synth.SIG_NAMES
t0 = time.time()
CODES = {"open": 0, "tripole": 3, "tripoleT": 4}
for name, ns in [(x, CODES[x]) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("tripole", "tripoleT"))]:
    ctx = lib.Context()
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=8, **({"land_rows": 0} if ns else {}))
    grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True) if ns else synth.block_fields(gg, dom)
    s = synth.evp_state(grid, dom, seed=8, cover="patchy")
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 0)
    ref = {k: v.copy() for k, v in s.items()}
    ctx.evp(DT, ref)
    ctx.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    ctx.evp_set_option("resident", 2)
    assert ctx.evp_get_info("resident") == 1
    ctx.evp_upload({k: v.copy() for k, v in s.items()})
    out = {k: np.empty_like(v) for k, v in s.items()}
    for rep in range(reps):
        ctx.evp_prepare(DT); ctx.evp_subcycles(1, NDTE); ctx.evp_finish()
        ctx.evp_download(out)
        for k in KEYS:
            if not np.array_equal(out[k], ref[k]):
                raise SystemExit("SOAK FAILED %s rep %d field %s: %s" % (name, rep, k, np.argwhere(out[k] != ref[k])[:4].tolist()))
        assert ctx.evp_get_info("resident") == 1, "fell back at rep %d" % rep
        ctx.evp_upload({k: v.copy() for k, v in s.items()})
    ctx.close()
print("SOAK-OK one-launch loop: %d evp(dt) calls of %d subcycles with each boundary on %d x %d, every one the bits of the per-subcycle path, %.0f s"
      % (reps, NDTE, nxg, nyg, time.time() - t0))
