"""Where a step of the K-subcycle sweep goes, level by level (0.1 degree): cycles every wavefront spends between marked points
of a step, summed over a sweep -- DIAGNOSTIC build only (scripts/build_ab.sh stamps -DCICE4_AMD_STAMPS).
usage: sweep_phases.py <lib_stamps.so> [nxg nyg]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
from cice4_amd import synth
nxg = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
nyg = int(sys.argv[3]) if len(sys.argv) > 3 else 2400
K, ndte = 4, 240
NAMES = ["hand-off written + this step's loads issued", "stress (incl. the wait for its stresses / inputs)",
         "momentum, stores, hand-off of u, v", "barrier"]
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_set_option("use_graph", 0)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    ctx.evp_subcycles(1, ndte); ctx.sync()
ctx.evp_set_option("stamps", 1)
ms = ctx.evp_subcycles(1, K, timed=True)
raw = ctx.evp_debug("stamps")
g = len(raw) // (4 + 8 * K)
st = raw[:4 * g].reshape(-1, 4).astype(np.float64)
ph = raw[4 * g:].reshape(g, K, 8).astype(np.float64)
ok = st[:, 1] > st[:, 0]
ghz = np.median((st[ok, 1] - st[ok, 0]) / (st[ok, 3] - st[ok, 2]) * 0.1)
seg = ctx.evp_get_info("skew_seg_rows")
steps = seg + 1 + 2 * (K - 1) + 1
print(f"{nxg} x {nyg}: sweep of {K} subcycles {ms * 1e3:.0f} us (with the stamps), {ok.sum()} workgroups, {seg} rows per segment = {steps} steps, clock {ghz:.3f} GHz")
print("per STEP, median over workgroups (us); a wavefront's level is dealt by (wavefront + tile) mod K, so the rows below are levels:")
for k in range(K):
    row = []
    for i in range(4):
        row.append(np.median(ph[ok, k, i]) / steps / ghz / 1e3)
    print(f"  level {k}: " + "  ".join(f"{NAMES[i][:28]:28s} {row[i]:5.2f}" for i in range(4)) + f"   sum {sum(row):5.2f}")
