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
if os.environ.get("GEN_PCT"):
    ctx.evp_set_option("skew_gen_pct", int(os.environ["GEN_PCT"]))
ctx.evp_upload(state); ctx.evp_prepare(3600.0)
ctx.evp_subcycles(1, ndte); ctx.sync()
ctx.evp_set_option("stamps", 1)
ctx.evp_subcycles(1, K)
raw = ctx.evp_debug("stamps")
g = len(raw) // (4 + 8 * K)
ph = raw[4 * g:].reshape(g, K, 8)
st = raw[:4 * g].reshape(-1, 4).astype(np.float64)   # cycles at start / end, 100 MHz ticks at start / end
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

# how long a workgroup runs, by the number of workgroups its CU holds (CICE4_AMD_SKEW_FILL gives the ones on emptier CUs more rows)
t0 = min(st[w, 2] for v in cus.values() for w in v)
for n in sorted(set(len(v) for v in cus.values())):
    ws = [w for v in cus.values() if len(v) == n for w in v]
    dur = np.array([(st[w, 3] - st[w, 2]) * 0.01 for w in ws])
    end = np.array([(st[w, 3] - t0) * 0.01 for w in ws])
    print(f"CUs with {n} workgroups: {len(ws)} workgroups run {np.median(dur):.0f} us (min {dur.min():.0f}, max {dur.max():.0f}); "
          f"they end {np.median(end):.0f} us after the first start (max {end.max():.0f})")
# ... and by the order in which the workgroups of a CU were dispatched (blockIdx: b, b + 256, b + 512 share a CU)
for n in sorted(set(len(v) for v in cus.values())):
    for gen in range(n):
        ws = [sorted(v)[gen] for v in cus.values() if len(v) == n]
        dur = np.array([(st[w, 3] - st[w, 2]) * 0.01 for w in ws])
        print(f"  CUs with {n}: workgroup no. {gen} of its CU runs {np.median(dur):.0f} us (min {dur.min():.0f}, max {dur.max():.0f}, 10 % {np.percentile(dur, 10):.0f}, 90 % {np.percentile(dur, 90):.0f})")
last = np.array([max(st[w, 3] for w in v) - t0 for v in cus.values()]) * 0.01
print(f"a CU is done after {np.median(last):.0f} us (min {last.min():.0f}, max {last.max():.0f}, 10 % {np.percentile(last, 10):.0f}, 90 % {np.percentile(last, 90):.0f})")
print("gen_pct", os.environ.get("GEN_PCT", "0"))
print("fill", os.environ.get("CICE4_AMD_SKEW_FILL", "0"), "info", ctx.evp_get_info("skew_fill"))
