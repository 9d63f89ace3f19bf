"""Where the time of one subcycle of the one-launch loop goes (gx1, dense shape): cycles wavefront 0 of every workgroup spends
between marked points of a subcycle, summed over a launch -- DIAGNOSTIC build only (scripts/build_ab.sh stamps
-DCICE4_AMD_STAMPS); the product build holds no stamp.  usage: resident_phases.py <lib_stamps.so> [out.csv]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
from cice4_amd import synth
NAMES = ["stress (LDS reads, wave shifts, 460 fp64 instructions)", "barrier (C): str rows of the wavefront above", "momentum",
         "edge stores issued (agent scope)", "stores drained (s_waitcnt vmcnt(0))", "barrier (D)",
         "progress word + poll of the producers' words + barrier (E)", "agent-scope loads of the exchanged velocities (drained)"]
NAMES_G = ["row below seen in LDS + stress", "str row of the wavefront above seen in LDS", "momentum, granule stores issued, row posted",
           "first pass of the poll", "PASSES of the poll (a count, not cycles)", "TICKS of 10 ns from the store of lane 0's granules to their arrival (not cycles)",
           "rest of the poll of the granules this wavefront needs", "(top of the loop)"]
rows = []
for W, dense, gran in ((0, 1, 1), (0, 1, 0), (11, 0, 1), (12, 0, 1), (11, 0, 0)):
    ctx = lib.Context(device=0)
    nxg, nyg, ndte = 320, 384, 120
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
    state = synth.evp_state(grid, dom, cover="full")
    ctx.evp_init(grid, ndte=ndte)
    ctx.evp_set_option("use_graph", 0); ctx.evp_set_option("resident_waves", W); ctx.evp_set_option("resident_dense", dense)
    ctx.evp_set_option("resident_granules", gran)
    ctx.evp_upload(state); ctx.evp_prepare(3600.0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20):
            ctx.evp_subcycles(1, ndte)
        ctx.sync()
    ctx.evp_set_option("stamps", 1)
    ms = ctx.evp_subcycles(1, ndte, timed=True)
    raw = ctx.evp_debug("stamps")
    g = len(raw) // 12
    st = raw[:4 * g].reshape(-1, 4).astype(np.float64)
    ph = raw[4 * g:12 * g].reshape(-1, 8).astype(np.float64)
    ok = st[:, 1] > st[:, 0]
    ghz = np.median((st[ok, 1] - st[ok, 0]) / (st[ok, 3] - st[ok, 2]) * 0.1)
    per = ph[ok] / ndte                      # cycles per subcycle
    tot = per.sum(axis=1)
    print(f"gx1, W = {ctx.evp_get_info('resident_waves')}, dense {ctx.evp_get_info('resident_dense')}, granules {gran}: {ok.sum()} workgroups, step {ms * 1e3:.1f} us "
          f"= {ms * 1e3 / ndte:.2f} us per subcycle (with the stamps and the extra drain), clock {ghz:.3f} GHz; per subcycle, median over workgroups:")
    for i, n in enumerate(NAMES_G if gran else NAMES):
        med = np.median(per[:, i])
        print(f"   {i}: {med:8.0f} cycles = {med / ghz / 1e3:6.3f} us  (p10 {np.percentile(per[:, i], 10):7.0f}, p90 {np.percentile(per[:, i], 90):7.0f})  {n}")
        rows.append((ctx.evp_get_info('resident_waves'), dense, i, n, med, med / ghz / 1e3))
    print(f"   sum {np.median(tot):8.0f} cycles = {np.median(tot) / ghz / 1e3:.3f} us")
    del ctx
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write("W,dense,phase,what,cycles_per_subcycle_median,us_per_subcycle\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},{r[2]},\"{r[3]}\",{r[4]:.0f},{r[5]:.4f}\n")
