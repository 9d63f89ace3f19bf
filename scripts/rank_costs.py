"""What ONE rank of an N-rank job costs per subcycle, rank by rank, through the REAL slab code path.

Rank r of N is set up exactly as `bench.py --gpus N` sets it up -- wide-halo j-slab with bench.auto_overlap's H, K-subcycle
sweeps over the tile lists of Evp::tiles_for where the slab is large enough, pack / unpack of the refresh every H subcycles --
and runs ALONE on the GPU: its messages come back to it through the mirror link (cice_comm_init_mirror: device-to-device, no
partner), so the time is kernels + pack + unpack of that rank with nobody else on the chip, i.e. what one GPU of an N-GPU
node spends, minus the link.  The fields computed are those of a mirror boundary and mean nothing; parity of the same
decompositions is tests/test_gpu_fullsize.py::test_*_on_eight_ranks.

usage: rank_costs.py [--workload gx1|tenth] [--ranks 2,4,8] [--steps S] [--out file]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cice4_amd import lib, synth  # noqa: E402

DT = 3600.0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workload", default="tenth")
    p.add_argument("--ranks", default="8")
    p.add_argument("--only", default="", help="comma-separated ranks to run (default: all)")
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--overlap", type=int, default=-1)
    p.add_argument("--split", type=int, default=0)
    p.add_argument("--balance", type=int, default=-1)
    p.add_argument("--out", default="")
    a = p.parse_args()
    bench = importlib.import_module("bench")
    nxg, nyg, ndte, _ = bench.workload(a.workload)
    gg = synth.global_grid(nxg, nyg)
    rows_out = []
    for N in [int(x) for x in a.ranks.split(",")]:
        rows = nyg // N
        H = a.overlap if a.overlap >= 0 else bench.auto_overlap(nxg, rows)
        for r in range(N):
            if a.only and str(r) not in a.only.split(","):
                continue
            ctx = lib.Context(device=0)
            if N == 1:
                dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
            else:
                dom = ctx.domain_create_slabs(nxg, nyg, N, ew=1, ns=0, rank=r, nranks=N, overlap=H)
                ctx.comm_init_mirror(r, N)
            grid = synth.block_fields(gg, dom)
            state = synth.evp_state(grid, dom, cover="full")
            ctx.evp_init(grid, ndte=ndte)
            if N > 1:
                ctx.evp_set_option("skew_split", a.split)
            if a.balance >= 0:
                ctx.evp_set_option("skew_balance", a.balance)
            ctx.evp_upload(state)
            ctx.evp_prepare(DT)
            ctx.sync()
            for _ in range(3):                     # graph / tables / (where the library balances) its measuring loops
                ctx.evp_subcycles(1, ndte)
            ctx.sync()
            t0 = time.perf_counter()
            n = 0
            while n < a.steps or time.perf_counter() - t0 < 0.5:
                ctx.evp_subcycles(1, ndte)
                n += 1
            ctx.sync()
            us = (time.perf_counter() - t0) / (n * ndte) * 1e6
            rec = dict(workload=a.workload, ranks=N, rank=r, overlap=H, owned_rows=rows,
                       slab_rows=int(dom["jhi"][0] - dom["jlo"][0] + 1), sweeps=bool(ctx.evp_get_info("skew")),
                       K=ctx.evp_get_info("skew_levels") if ctx.evp_get_info("skew") else 0,
                       resident=bool(ctx.evp_get_info("resident")), launches_per_step=ctx.evp_get_info("last_launches"),
                       balanced_sweeps=ctx.evp_get_info("skew_balanced"), split=a.split,
                       us_per_subcycle=round(us, 2))
            rows_out.append(rec)
            print(json.dumps(rec), flush=True)
            del ctx, state, grid
    if a.out:
        with open(a.out, "w") as f:
            for rec in rows_out:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
