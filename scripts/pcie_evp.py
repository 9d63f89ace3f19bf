"""Where the 2.8 ms of a host-array evp(dt) go at gx1 size: upload / prepare+loop+finish / download, page-locked arrays.
usage: python scripts/pcie_evp.py [calls]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = lib.Context()
nxg, nyg = 320, 384
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=5), dom, ew_cyclic=True)
s = synth.evp_state(grid, dom, seed=3, cover="full")
for v in s.values():
    if isinstance(v, np.ndarray):
        ctx.host_register(v)
ctx.evp_init(grid, ndte=120)
s0 = {k: v.copy() for k, v in s.items() if isinstance(v, np.ndarray)}
whole, up, mid, down = [], [], [], []
for it in range(calls):
    for k, v in s0.items():
        s[k][...] = v
    t0 = time.perf_counter(); ctx.evp(3600.0, s); whole.append(time.perf_counter() - t0)
for it in range(calls):
    for k, v in s0.items():
        s[k][...] = v
    t0 = time.perf_counter(); ctx.evp_upload(s); t1 = time.perf_counter()
    ctx.evp_prepare(3600.0); ctx.evp_subcycles(1, 120); ctx.evp_finish(); ctx.sync(); t2 = time.perf_counter()
    ctx.evp_download(s); t3 = time.perf_counter()
    up.append(t1 - t0); mid.append(t2 - t1); down.append(t3 - t2)
nbytes = lambda names: sum(s[k].nbytes for k in names if k in s)
print("evp(dt) over PCIe: min %.3f ms  median %.3f ms" % (1e3 * min(whole[2:]), 1e3 * np.median(whole[2:])))
print("  upload   %.3f ms   compute %.3f ms   download %.3f ms  (min over %d)" % (1e3 * min(up[2:]), 1e3 * min(mid[2:]), 1e3 * min(down[2:]), calls - 2))
