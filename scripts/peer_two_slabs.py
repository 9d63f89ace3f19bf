"""The cross-rank resident loop on ONE GPU: R contexts (threads) of this process, one slab each, neighbours' exchange
copies connected by plain device pointers.  Times the ndte-subcycle loop; for DESIGN.md section 7.
usage: peer_two_slabs.py [R] [nx] [ny] [ndte]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
torch.cuda.is_available()
from cice4_amd import lib, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nxg = int(sys.argv[2]) if len(sys.argv) > 2 else 320
nyg = int(sys.argv[3]) if len(sys.argv) > 3 else 384
ndte = int(sys.argv[4]) if len(sys.argv) > 4 else 120
REPS = 200
gg = synth.global_grid(nxg, nyg)
bar = threading.Barrier(R)
exports, res, errs = [None] * R, [None] * R, []

def rank_fn(r, peer):
    try:
        c = lib.Context(device=0); c.sync()
        dom = c.domain_create(nxg, nyg, nxg, nyg // R, ew=1, ns=0, rank=r, npx=1, npy=R)
        c.comm_init_local(77 + int(peer), r, R)
        grid = synth.block_fields(gg, dom)
        s = synth.evp_state(grid, dom, cover="full")
        c.evp_init(grid, ndte=ndte)
        c.evp_set_option("resident_peer_share", R)
        c.evp_set_option("resident_peer_agree", 0)     # timing: no host round trip after the launch
        if peer:
            exports[r] = c.evp_peer_export()
            bar.wait()
            if r > 0: c.evp_peer_connect(0, exports[r - 1])
            if r < R - 1: c.evp_peer_connect(1, exports[r + 1])
            assert c.evp_get_info("resident_peer") == 1
        else:
            c.evp_set_option("resident", 0)
        c.evp_upload(s); c.evp_prepare(3600.0)
        W = c.evp_get_info("resident_waves") if peer else 0
        for _ in range(5):
            bar.wait(); c.evp_subcycles(1, ndte); c.sync()
        bar.wait()
        t0 = time.perf_counter()
        n = REPS if peer else 3
        for _ in range(n):
            c.evp_subcycles(1, ndte)
        c.sync()
        bar.wait()
        dt = time.perf_counter() - t0
        ok = (c.evp_get_info("resident_peer") == 1) if peer else True
        res[r] = (dt / n, W, ok)
        bar.wait()
    except BaseException as e:  # noqa: BLE001
        errs.append((r, repr(e))); bar.abort()

for peer in (True, False):
    th = [threading.Thread(target=rank_fn, args=(r, peer)) for r in range(R)]
    [t.start() for t in th]; [t.join(600) for t in th]
    if errs:
        print("FAILED", errs); sys.exit(1)
    t = max(x[0] for x in res)
    print(f"{'peer loop (one launch per rank, device-initiated exchange)' if peer else 'one launch per subcycle + messages through the in-process link (host mailbox: not a performance path)'}: "
          f"{R} slabs of {nxg}x{nyg // R} on one GPU, {ndte} subcycles in {t * 1e3:.3f} ms = {t / ndte * 1e6:.2f} us per subcycle"
          + (f", W = {res[0][1]}, fell back: {not all(x[2] for x in res)}" if peer else ""))
# one context, whole grid, for comparison
c = lib.Context(device=0)
dom = c.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
grid = synth.block_fields(gg, dom); s = synth.evp_state(grid, dom, cover="full")
c.evp_init(grid, ndte=ndte)
c.evp_upload(s); c.evp_prepare(3600.0)
for dense in (1, 0):
    c.evp_set_option("resident", 2); c.evp_set_option("resident_dense", dense)
    for _ in range(20): c.evp_subcycles(1, ndte)
    c.sync(); t0 = time.perf_counter()
    for _ in range(REPS): c.evp_subcycles(1, ndte)
    c.sync(); t = (time.perf_counter() - t0) / REPS
    print(f"one context, whole grid, resident loop ({'dense' if dense else 'one workgroup per CU'}, W = {c.evp_get_info('resident_waves')}): {t / ndte * 1e6:.2f} us per subcycle")
