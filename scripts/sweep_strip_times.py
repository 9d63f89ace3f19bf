"""After the sweep's tuning phase (0.1 degree, full cover): how long the workgroups of every STRIP take, by strip -- what is left
for a balance across strips.  usage: python scripts/sweep_strip_times.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth
nxg, nyg, K, ndte = 3600, 2400, 4, 240
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
for _ in range(3):
    ctx.evp_subcycles(1, ndte)
ctx.sync()
ctx.evp_set_option("use_graph", 0); ctx.evp_set_option("skew_debug", 1)
acc = None
for i in range(8):
    ctx.evp_subcycles(1 + 4 * i, K); ctx.sync()
    tm = ctx.evp_debug("skew_times").reshape(-1, 2).astype(np.float64)
    acc = tm if acc is None else acc + tm
tab = ctx.evp_debug("skew_rows").reshape(-1, 3)
nt = len(tab); chunk = (nt + 7) // 8
d = np.zeros(nt)
for p in range(nt):
    b = ((p % chunk) << 3) | (p // chunk)
    d[p] = (acc[b, 1] - acc[b, 0]) * 0.01 / 8
strips = int(tab[:, 0].max()) + 1
mean = np.array([d[tab[:, 0] == s].mean() for s in range(strips)]); mx = np.array([d[tab[:, 0] == s].max() for s in range(strips)])
cnt = np.array([(tab[:, 0] == s).sum() for s in range(strips)])
print("tiles", nt, "launch (slowest workgroup) %.0f us, mean workgroup %.0f us" % (d.max(), d.mean()))
for c in sorted(set(cnt.tolist())):
    m = cnt == c
    print("strips with %d tiles: %d; mean of their workgroups %.0f us (%.0f .. %.0f by strip), slowest workgroup by strip %.0f .. %.0f" % (c, m.sum(), mean[m].mean(), mean[m].min(), mean[m].max(), mx[m].min(), mx[m].max()))
print("spread of the strips' means: %.1f %% (max / mean - 1)" % (100 * (mean.max() / mean.mean() - 1)))
