"""Solver iterations per column in the bench's thermo workload (tuning aid; needs a GPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
import bench
from cice4_amd import lib
coh = int(sys.argv[1]) if len(sys.argv) > 1 else bench.THERMO_COHERENCE
ctx = lib.Context(device=0)
dom = ctx.domain_create(320, 384, 320, 384, ew=1, ns=0)
ctx.thermo_init()
tb, _ = bench.thermo_case(dom, coherent=coh)
ctx.thermo_batch_alloc(dom["nx"], dom["ny"], dom["nblocks"])
ctx.thermo_batch_upload(tb)
st = ctx.thermo_batch_step(3600.0, yday=150.0, timed=True)
nit = ctx.evp_debug("thermo_niter").view(np.uint8)[:5 * dom["ny"] * dom["nx"]].reshape(5, dom["ny"], dom["nx"])
act = tb["aicen"][0] > 1e-11
act[:, [0, -1], :] = False; act[:, :, [0, -1]] = False
v = nit[act]
print("coherence", coh, "ms", st["ms"], "columns", act.sum(), "mean iterations", v.mean())
print("histogram:", {int(k): int(c) for k, c in zip(*np.unique(v, return_counts=True))})
# wave-level: rows of 64 consecutive cells (the dense kernel's wavefronts)
flat = np.where(act, nit, 0).reshape(5, -1)
n64 = flat.shape[1] // 64
w = flat[:, :n64 * 64].reshape(5, n64, 64)
mx, sm, cnt = w.max(axis=2), w.sum(axis=2), (w > 0).sum(axis=2)
print("lane-iterations used / issued (max over the wave x 64): %.3f" % (sm.sum() / (mx.sum() * 64)))
for G in (8, 16):
    g = flat[:, :(flat.shape[1] // G) * G].reshape(5, -1, G)
    gm = g.max(axis=2)
    # sort groups by their max, then cut into waves of 64/G groups
    tot_used, tot_issued = 0, 0
    for n in range(5):
        o = np.argsort(gm[n], kind="stable")
        gs = g[n][o]
        per = 64 // G
        k = (len(gs) // per) * per
        ww = gs[:k].reshape(-1, per * G)
        tot_used += ww.sum(); tot_issued += ww.max(axis=1).sum() * 64
    print(f"groups of {G} adjacent columns sorted by their maximum: used / issued = {tot_used / tot_issued:.3f}")
for C in (256, 512, 2048):
    tot_used, tot_issued = 0, 0
    for n in range(5):
        f = flat[n][: (flat.shape[1] // C) * C].reshape(-1, C)
        fs = np.sort(f, axis=1).reshape(-1, 64)
        tot_used += fs.sum(); tot_issued += fs.max(axis=1).sum() * 64
    print(f"columns sorted inside chunks of {C}: used / issued = {tot_used / tot_issued:.3f}")
# the wavefronts the sorted kernel really formed in a second pass (keys = the first pass's iteration counts)
ctx.thermo_batch_upload(tb)
st2 = ctx.thermo_batch_step(3600.0, yday=150.0, timed=True)
perm = ctx.evp_debug("thermo_perm").view(np.int32)
if len(perm):
    npl = len(perm) // 5
    nit2 = ctx.evp_debug("thermo_niter").view(np.uint8)[:5 * dom["ny"] * dom["nx"]].reshape(5, -1)
    assert np.array_equal(nit2.reshape(nit.shape), nit), "iteration counts differ between two passes over the same state"
    used = issued = 0
    worst = []
    for n in range(5):
        p = perm[n * npl:(n + 1) * npl]
        q = p & 0x7fffffff
        a_ = (p >= 0)
        it = np.where(a_, flat[n][np.minimum(q, flat.shape[1] - 1)], 0)
        w = it[: (len(it) // 64) * 64].reshape(-1, 64)
        used += w.sum(); issued += w.max(axis=1).sum() * 64
        worst.append(w.max(axis=1))
    print("second pass ms", st2["ms"], "sorted kernel: lane-iterations used / issued = %.3f" % (used / issued),
          "mean of wave maxima", np.concatenate(worst).mean())
else:
    print("second pass ms", st2["ms"], "(not sorted)")
