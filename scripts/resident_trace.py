"""Who runs when in the free-running one-launch loop (DIAGNOSTIC build, scripts/build_ab.sh stamps -DCICE4_AMD_STAMPS): wall-clock
ticks (10 ns) of every wavefront of a few tiles at six points of subcycles 60 .. 63.  usage: resident_trace.py <lib_stamps.so> [W] [prio]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
from cice4_amd import synth
W = int(sys.argv[2]) if len(sys.argv) > 2 else 11
prio = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = lib.Context(device=0)
nxg, nyg, ndte = 320, 384, 120
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_set_option("use_graph", 0); ctx.evp_set_option("resident_waves", W); ctx.evp_set_option("resident_dense", 1 if W == 4 else 0)
ctx.evp_set_option("resident_granules", 1); ctx.evp_set_option("resident_prio", prio)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    for _ in range(20):
        ctx.evp_subcycles(1, ndte)
    ctx.sync()
ctx.evp_set_option("stamps", 1)
ms = ctx.evp_subcycles(1, ndte, timed=True)
raw = ctx.evp_debug("stamps")
g = (len(raw) - 2400) // 12
tr = raw[12 * g:12 * g + 2304].reshape(8, 12, 4, 6).astype(np.int64)
print(f"gx1, W = {W}, prio mode {prio}: {ms * 1e3 / ndte:.2f} us per subcycle (diagnostic build)")
names = ("top", "S go", "S end", "M go", "M end", "poll end")
for slot in range(8):
    t = tr[slot]
    if not t.any():
        continue
    base = t[t > 0].min()
    print(f"tile {slot * 29 + 7}: ticks of 10 ns since the first stamp; per wavefront, subcycles 60..63: " + " | ".join(names))
    for w in range(12):
        if not t[w].any():
            continue
        print(f"  w{w:2d}: " + "   ".join(" ".join(f"{(int(x) - base) if x else -1:4d}" for x in t[w, k]) for k in range(4)))
    break
