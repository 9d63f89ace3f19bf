"""How the sweep's segment table evolves under the measured-cost balancing: usage: sweep_balance_trace.py [cover] [nxg nyg]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib, synth
cover = sys.argv[1] if len(sys.argv) > 1 else "caps"
nxg = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
nyg = int(sys.argv[3]) if len(sys.argv) > 3 else 2400
K, ndte = 4, 240
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover=cover)
ctx.evp_init(grid, ndte=ndte)
ctx.evp_set_option("skew_debug", 1)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
strips = ctx.evp_get_info("skew_strips")
print("rowact", ctx.evp_get_info("skew_rowact"), "balance", ctx.evp_get_info("skew_balance"), "strips", strips, flush=True)
for call in range(6):
    ms = ctx.evp_subcycles(1, ndte, timed=True)
    t = ctx.evp_debug("skew_rows").reshape(-1, 3)        # per tile (place in the launch): strip, first / last row
    tm = ctx.evp_debug("skew_times").reshape(-1, 2)
    nt = len(t); chunk = (nt + 7) >> 3
    p = np.arange(nt); b = ((p % chunk) << 3) | (p // chunk)
    d = (tm[b, 1] - tm[b, 0]) * 0.01
    print(f"call {call}: {ms * 1e3 / ndte:.1f} us per subcycle; measured sweeps so far {ctx.evp_get_info('skew_balanced')}; {int((t[:, 2] >= t[:, 1]).sum())} tiles on {nt} places; "
          f"last sweep: slowest workgroup {d.max():.0f} us, mean of the non-empty {d[d > 0].mean():.0f} us, {int((d > 0).sum())} with rows to do")
    ms_ = np.array([d[t[:, 0] == sx].mean() for sx in range(strips)])
    cnt = np.array([(t[:, 0] == sx).sum() for sx in range(strips)])
    order = np.argsort(-ms_)
    print("   strips by mean workgroup time (strip, us, tiles): slowest", [(int(i), int(round(ms_[i])), int(cnt[i])) for i in order[:5]],
          "median", int(round(np.median(ms_))), "fastest", [(int(i), int(round(ms_[i])), int(cnt[i])) for i in order[-3:]],
          "| tiles per strip:", dict(zip(*[x.tolist() for x in np.unique(cnt, return_counts=True)])), flush=True)
    for sx in (0, 7, strips - 1):
        ts = np.where(t[:, 0] == sx)[0]
        ts = ts[np.argsort(t[ts, 1], kind="stable")]
        print(f"   strip {sx}: rows per segment {(t[ts, 2] - t[ts, 1] + 1).tolist()}  last sweep us {np.round(d[ts]).astype(int).tolist()}", flush=True)
