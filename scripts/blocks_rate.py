"""EVP subcycle rate of a ONE-RANK domain cut into several blocks: the one-launch loop (round 4) against one launch per
subcycle + on-rank halo copies (what such domains ran before).  usage: blocks_rate.py nxg nyg bsx bsy [ndte]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library: see bench.py)
torch.cuda.is_available()
from cice4_amd import lib, synth
nxg, nyg, bsx, bsy = (int(x) for x in sys.argv[1:5])
ndte = int(sys.argv[5]) if len(sys.argv) > 5 else 120
ctx = lib.Context(device=0)
dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
grid = synth.block_fields(synth.global_grid(nxg, nyg), dom)
state = synth.evp_state(grid, dom, cover="full")
for label, opts in (("one launch per evp(dt) (k_evp_resident on %d blocks)" % dom["nblocks"], {"resident": 2}),
                    ("one launch per subcycle (k_subcycle + on-rank halo)", {"resident": 0})):
    ctx.evp_init(grid, ndte=ndte)
    for k, v in opts.items():
        ctx.evp_set_option(k, v)
    ctx.evp_upload(state); ctx.evp_prepare(3600.0)
    for _ in range(5):
        ctx.evp_subcycles(1, ndte)
    ctx.sync()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(10):
            ctx.evp_subcycles(1, ndte)
        ctx.sync(); n += 10
    dt = (time.perf_counter() - t0) / n
    print(f"{nxg}x{nyg} in {dom['nblocks']} blocks of {bsx}x{bsy}: {label}: {dt / ndte * 1e6:.2f} us per subcycle "
          f"({ndte / dt:.0f} subcycles/s), launches per call {ctx.evp_get_info('last_launches')}, resident {ctx.evp_get_info('resident')}"
          + (f", W = {ctx.evp_get_info('resident_waves')}, dense {ctx.evp_get_info('resident_dense')}" if ctx.evp_get_info('resident') else ""), flush=True)
