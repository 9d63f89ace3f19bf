"""Which time levels of the K-subcycle sweep share a SIMD (0.1 degree): every wavefront reports the XCC / SE / CU / SIMD it
ran on -- DIAGNOSTIC build only (scripts/build_ab.sh stamps -DCICE4_AMD_STAMPS).  CICE4_AMD_SKEW_DEAL selects the deal.
usage: sweep_placement.py <lib_stamps.so> [nxg nyg]"""
import collections, os, sys
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
K, ndte = 4, 8
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
ctx.evp_init(grid, ndte=ndte)
ctx.evp_set_option("use_graph", 0)
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
ctx.evp_subcycles(1, ndte); ctx.sync()
ctx.evp_set_option("stamps", 1)
ctx.evp_subcycles(1, K)
raw = ctx.evp_debug("stamps")
g = len(raw) // (4 + 8 * K)
ph = raw[4 * g:].reshape(g, K, 8)
simd = collections.defaultdict(list)   # (xcc, se, cu, simd) -> levels
cus = collections.defaultdict(set)
for w in range(g):
    for k in range(K):
        hw, xcc = int(ph[w, k, 7]), int(ph[w, k, 6]) & 15
        if hw == 0 and xcc == 0 and ph[w, k, 0] == 0:
            continue
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        simd[key + ((hw >> 4) & 3,)].append(k)
        cus[key].add(w)
print(f"deal {os.environ.get('CICE4_AMD_SKEW_DEAL', '0')}: {g} workgroups on {len(cus)} CUs; workgroups per CU:",
      dict(collections.Counter(len(v) for v in cus.values())))
mix = collections.Counter(tuple(sorted(v)) for v in simd.values())
print("levels sharing a SIMD (sorted) -> number of SIMDs:")
for m, c in sorted(mix.items(), key=lambda x: -x[1])[:16]:
    print("  ", m, c)
ex = sorted(cus)[0]
print("example CU", ex, {s: sorted(simd[ex + (s,)]) for s in range(4)}, "workgroups", sorted(cus[ex]))
