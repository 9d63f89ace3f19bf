"""Raw material for balancing the sweep: per workgroup (blockIdx) where it ran and how long, for several sweeps in a row --
DIAGNOSTIC build (scripts/build_ab.sh stamps -DCICE4_AMD_STAMPS).  usage: sweep_wg_times.py <lib_stamps.so> out.npz [nsweeps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
from cice4_amd import synth
nxg, nyg, K, ndte = 3600, 2400, 4, 120
nsw = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_set_option("use_graph", 0)
if os.environ.get("GEN_PCT"):
    ctx.evp_set_option("skew_gen_pct", int(os.environ["GEN_PCT"]))
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:      # sustained clocks
    ctx.evp_subcycles(1, ndte); ctx.sync()
ctx.evp_set_option("stamps", 1)
out = []
for i in range(nsw):
    ctx.evp_subcycles(1 + 4 * i, K)
    raw = ctx.evp_debug("stamps")
    g = len(raw) // (4 + 8 * K)
    st = raw[:4 * g].reshape(-1, 4).copy()
    ph = raw[4 * g:].reshape(g, K, 8)
    out.append(np.concatenate([st, ph[:, :, 7], ph[:, :, 6]], axis=1))   # 4 stamps, 4 HW_ID (by level), 4 XCC_ID
np.savez_compressed(sys.argv[2], wg=np.stack(out), seg=ctx.evp_get_info("skew_seg_rows"), strips=ctx.evp_get_info("skew_strips"))
a = np.stack(out).astype(np.float64)
dur = (a[:, :, 3] - a[:, :, 2]) * 0.01
ok = dur[0] > 0
print("sweeps", nsw, "workgroups", int(ok.sum()), "kernel (max end - min start) us:",
      [round(float((a[i, ok, 3].max() - a[i, ok, 2].min()) * 0.01)) for i in range(nsw)])
c = np.corrcoef(dur[:, ok])
print("correlation of a workgroup's duration between sweeps (systematic part): min %.2f median %.2f" % (c[np.triu_indices(nsw, 1)].min(), np.median(c[np.triu_indices(nsw, 1)])))
