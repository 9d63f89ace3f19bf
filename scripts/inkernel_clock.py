"""In-kernel shader clock of the two EVP loops (MI355X_MICROARCH.md, DVFS item 6).  Needs a GPU and the DIAGNOSTIC build
(scripts/build_ab.sh stamps -DCICE4_AMD_STAMPS -> build/ab/lib_stamps.so): that build stamps s_memtime / s_memrealtime once
before and once after the subcycle loop of k_evp_resident and the row loop of k_subcycle_skew; the product build has no stamp.
clock = d(cycles) / d(100 MHz ticks) x 100 MHz per workgroup, after >= 2 s of back-to-back launches; median over workgroups.
usage: inkernel_clock.py <lib.so> <out.csv> [seconds]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library: see bench.py)
torch.cuda.is_available()
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
from cice4_amd import synth
out_csv = sys.argv[2]
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
try:
    commit = open(os.path.join(ROOT, "commit.txt")).read().strip()
except OSError:
    commit = "unknown"
rows = []
for name, nxg, nyg, ndte, kernel in (("gx1", 320, 384, 120, "k_evp_resident (the shape the library picks, 120 subcycles per launch)"),
                                     ("tenth", 3600, 2400, 240, "k_subcycle_skew<4> (4 subcycles per sweep)")):
    ctx = lib.Context(device=0)
    dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
    grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
    state = synth.evp_state(grid, dom, cover="full")
    ctx.evp_init(grid, ndte=ndte)
    ctx.evp_set_option("use_graph", 0)
    ctx.evp_upload(state)
    ctx.evp_prepare(3600.0)
    ctx.evp_subcycles(1, ndte)
    ctx.sync()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:           # back-to-back launches, no stamp buffer yet
        for _ in range(20 if name == "gx1" else 1):
            ctx.evp_subcycles(1, ndte)
        ctx.sync()
        n += 1
    ctx.evp_set_option("stamps", 1)
    ms = ctx.evp_subcycles(1, ndte, timed=True)           # the stamped step follows at once
    raw = ctx.evp_debug("stamps")
    # [4 g] stamps come first; behind them the phase sums of the diagnostic build ([8 g] + a trace of 2400 words for the
    # one-launch loop, [8 K g] for the sweep): only the stamps are read here
    g = (len(raw) - 2400) // 12 if ctx.evp_get_info("resident") else len(raw) // (4 * (1 + 2 * ctx.evp_get_info("skew_levels")))
    st = raw[:4 * g].reshape(-1, 4).astype(np.float64)
    ok = (st[:, 1] > st[:, 0]) & (st[:, 3] > st[:, 2])
    if not ok.any():
        raise SystemExit(f"{name}: no stamps -- is {sys.argv[1]} the -DCICE4_AMD_STAMPS build?")
    cyc, ticks = st[ok, 1] - st[ok, 0], st[ok, 3] - st[ok, 2]
    ghz = cyc / ticks * 0.1
    rows.append((name, kernel, int(ok.sum()), float(np.median(ghz)), float(ghz.min()), float(ghz.max()),
                 float(np.median(ticks)) / 100.0, ms, ctx.evp_get_info("resident"), ctx.evp_get_info("skew")))
    print(f"{name}: {ok.sum()} workgroups, in-kernel clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}); "
          f"loop {np.median(ticks) / 100.0:.1f} us per workgroup; step {ms:.3f} ms; resident={rows[-1][8]} skew={rows[-1][9]}", flush=True)
    del ctx
with open(out_csv, "w") as f:
    import importlib
    sha = importlib.import_module("bench").kernel_source_sha()   # (the stamps are outside the loop: the diagnostic build times the product's code)
    f.write("workload,kernel,workgroups,clock_ghz_median,clock_ghz_min,clock_ghz_max,loop_us_median,step_ms,commit,source_sha,method\n")
    for r in rows:
        f.write(f"{r[0]},\"{r[1]}\",{r[2]},{r[3]:.4f},{r[4]:.4f},{r[5]:.4f},{r[6]:.2f},{r[7]:.4f},{commit},{sha},"
                f"\"s_memtime / s_memrealtime stamped once around the loop by thread 0 of every workgroup after {seconds} s of back-to-back launches; diagnostic build -DCICE4_AMD_STAMPS\"\n")
