"""Soak of the one-launch loop ACROSS ranks (k_evp_resident<.., PEER>: stores into the neighbour's exchange copies, remote
progress words) with R contexts of this process on one GPU: REPS whole evp(dt) calls from one state; every rank's result
has to be the bits of the first call, which are checked against the single-domain run.
(The R loops have to be on the chip at the same time, one CU per workgroup, and the dispatcher deals the workgroups of
a launch over 8 XCDs x 4 shader engines of 8 CUs: Evp::resident_waves picks a shape that fits every engine for R launches
together and refuses where there is none -- R = 3 on 320 x 384 -- which this script reports at set-up.  One process per
GPU, the deployment this loop is for, has the chip to itself.)
usage: python scripts/soak_peer.py [R] [reps] [nyg]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nyg = int(sys.argv[3]) if len(sys.argv) > 3 else (384 if R == 2 else 192)
nxg, nyg, NDTE, DT = 320, nyg // R * R, 120, 3600.0
KEYS = ("uvel", "vvel") + synth.SIG_NAMES
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.03, seed=8)
# the single-domain answer
c1 = lib.Context(device=0)
d1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
g1 = synth.block_fields(gg, d1, ew_cyclic=True)
s1 = synth.evp_state(g1, d1, seed=8, cover="patchy")
c1.evp_init(g1, ndte=NDTE, krdg_partic=0, krdg_redist=0)
c1.evp_set_option("resident", 0)
ref = {k: v.copy() for k, v in s1.items()}
c1.evp(DT, ref)
c1.close()      # its streams go back to the runtime: the rank contexts below should not have to share hardware queues
del c1
bar = threading.Barrier(R)
exports, errs, counts = [None] * R, [], [0] * R


def rank_fn(r):
    try:
        c = lib.Context(device=0); c.sync()
        dom = c.domain_create(nxg, nyg, nxg, nyg // R, ew=1, ns=0, rank=r, npx=1, npy=R)
        c.comm_init_local(91, r, R)
        grid = synth.block_fields(gg, dom, ew_cyclic=True)
        j0, nloc = r * (nyg // R), nyg // R
        s = synth.evp_state(grid, dom, seed=8, cover="patchy")     # a function of the global coordinates
        c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
        c.evp_set_option("resident_peer_share", R)
        exports[r] = c.evp_peer_export()
        bar.wait()
        if r > 0: c.evp_peer_connect(0, exports[r - 1])
        if r < R - 1: c.evp_peer_connect(1, exports[r + 1])
        assert c.evp_get_info("resident_peer") == 1
        want = {k: np.ascontiguousarray(ref[k][:, j0:j0 + nloc + 2]) for k in KEYS}
        for rep in range(REPS):
            sg = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.items()}
            bar.wait()
            c.evp(DT, sg)
            own = slice(1, -1)       # owned rows (ghost rows of the slab are the neighbour's)
            for k in KEYS:
                if not np.array_equal(sg[k][:, own], want[k][:, own]):
                    bad = np.argwhere(sg[k][:, own] != want[k][:, own])
                    raise AssertionError("rank %d rep %d field %s: %d cells differ, first %s" % (r, rep, k, len(bad), bad[:4].tolist()))
            assert c.evp_get_info("resident_peer") == 1, "fell back at rep %d" % rep
            counts[r] += 1
    except BaseException as e:  # noqa: BLE001
        errs.append((r, repr(e))); bar.abort()


t0 = time.time()
th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(R)]
[t.start() for t in th]; [t.join(1200) for t in th]
if errs:
    print("SOAK FAILED", errs); sys.exit(1)
print("SOAK-OK peer loop: %d slabs of %d x %d on one GPU, %d evp(dt) calls of %d subcycles each per rank, every one the single-domain bits, %.0f s"
      % (R, nxg, nyg // R, min(counts), NDTE, time.time() - t0))
