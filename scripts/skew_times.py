"""When do the workgroups of one K-subcycle sweep start and end?  (tuning aid; needs a GPU)
usage: skew_times.py [K] [prio] [seg_rows]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library: see bench.py)
torch.cuda.is_available()
from cice4_amd import lib, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = lib.Context(device=0)
nxg, nyg, ndte = 3600, 2400, 240
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
gg = synth.global_grid(nxg, nyg)
grid = synth.block_fields(gg, dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
for k, v in (("use_graph", 0), ("skew_levels", K), ("skew_prio", prio), ("skew_seg_rows", seg), ("skew_debug", 1)):
    ctx.evp_set_option(k, v)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
for _ in range(3):
    ctx.evp_subcycles(1, 4 * K)
ms = ctx.evp_subcycles(1 + 4 * K, K, timed=True)
t = ctx.evp_debug("skew_times").reshape(-1, 2).astype(np.float64)
ok = t[:, 1] > 0
t0 = t[ok, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # us
print(f"K={K} prio={prio} launch {ms * 1e3:.1f} us, workgroups {ok.sum()}")
n = len(st)
for lo in range(0, n, max(1, n // 12)):
    sl = slice(lo, min(n, lo + max(1, n // 12)))
    m = ok[sl]
    if m.any():
        print(f"  blockIdx {lo:5d}..: start {st[sl][m].mean():8.1f}  end {en[sl][m].mean():8.1f}  (min {en[sl][m].min():8.1f} max {en[sl][m].max():8.1f})  dur {(en[sl][m]-st[sl][m]).mean():8.1f}")
print("  end-time percentiles (us):", np.percentile(en[ok], [0, 10, 25, 50, 75, 90, 100]).round(1))
tiles_x = ctx.evp_get_info("skew_strips")
segr = ctx.evp_get_info("skew_seg_rows")
tiles_y = -(-2400 // segr)
nt = tiles_x * tiles_y
chunk = (nt + 7) >> 3
order = np.argsort(-en)
print("  slowest workgroups: blockIdx (strip, segment) end us")
for b in order[:24]:
    tl = (b & 7) * chunk + (b >> 3)
    print(f"    {b:5d} ({tl % tiles_x:3d},{tl // tiles_x:3d}) {en[b]:8.1f}", end="")
print()
# mean end time by strip and by segment
strip = np.array([(((b & 7) * chunk + (b >> 3)) % tiles_x) for b in range(n)]); segi = np.array([(((b & 7) * chunk + (b >> 3)) // tiles_x) for b in range(n)])
for s_ in (0, 1, 2, tiles_x // 2, tiles_x - 2, tiles_x - 1):
    m = ok & (strip == s_)
    print(f"  strip {s_:3d}: mean end {en[m].mean():8.1f} max {en[m].max():8.1f}")
for g in range(tiles_y):
    m = ok & (segi == g)
    print(f"  segment {g:3d}: mean end {en[m].mean():8.1f} max {en[m].max():8.1f}")
