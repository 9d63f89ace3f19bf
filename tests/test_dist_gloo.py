"""N > 1 path on CPU: two processes (gloo) each own one j-slab of the grid, use the library's
own per-rank block/halo topology (cice_domain_create with rank / npy = 2) and exchange the
ghost rows named by its send/recv address lists over torch.distributed -- the exchange the GPU
path performs with RCCL.  The subcycle arithmetic is the CPU checker's.  The two slabs
together must reproduce the single-domain run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT, NDTE, NSUB = 3600.0, 120, 12
NXG, NYG = 48, 40


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def prepared_case(ctx_dom, rank, world):
    from cice4_amd import synth
    gg = synth.global_grid(NXG, NYG, perturb=0.12, land_frac=0.04, seed=3)
    grid = synth.block_fields(gg, ctx_dom)
    s = synth.evp_state(grid, ctx_dom, seed=3, cover="patchy")
    return grid, s


def exchange(c, dom, fields, rank, world):
    """ghost update of `fields` (flat views): on-rank copies + p2p messages in the library's order."""
    sends = c.halo_msgs(0); recvs = c.halo_msgs(1)
    reqs, bufs = [], []
    for peer, addr in recvs:
        b = torch.empty(len(fields) * len(addr), dtype=torch.float64)
        bufs.append((addr, b)); reqs.append(dist.irecv(b, src=peer))
    for peer, addr in sends:
        t = torch.from_numpy(np.concatenate([f[addr] for f in fields]))
        reqs.append(dist.isend(t, dst=peer))
    for f in fields:
        f[dom["hdst"]] = f[dom["hsrc"]]
    for r in reqs:
        r.wait()
    for addr, b in bufs:
        v = b.numpy().reshape(len(fields), len(addr))
        for k, f in enumerate(fields):
            f[addr] = v[k]


def run_rank(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cice4_amd import lib, synth
    from oracle import oracle
    orc = oracle.Oracle()
    orc.set_evp_parameters(DT, NDTE)
    c = lib.Context()
    dom = c.domain_create(NXG, NYG, NXG, NYG // world, ew=1, ns=0, rank=rank, npx=1, npy=world)
    grid, s = prepared_case(dom, rank, world)
    ny, nx = dom["ny"], dom["nx"]
    # per-block prepared inputs of the subcycle loop (masks/lists from a fixed rule, not evp_prep)
    tm = (s["aice"][0] > 0.01) & (grid["tmask"][0] > 0); tm[0, :] = False; tm[:, 0] = False
    um = (grid["umask"][0] > 0) & (s["aice"][0] > 0.01); um[0, :] = um[-1, :] = False; um[:, 0] = um[:, -1] = False
    def lists(m):
        jj, ii = np.nonzero(m); n = len(ii)
        a = np.zeros(nx * ny, np.int32); b = np.zeros(nx * ny, np.int32); a[:n] = ii + 1; b[:n] = jj + 1
        return n, a, b
    icellt, ti, tj = lists(tm); icellu, ui, uj = lists(um)
    g = {k: np.ascontiguousarray(grid[k][0]) for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
                                                         "tarear", "tinyarea", "uarear")}
    u = np.ascontiguousarray(s["uvel"][0]); v = np.ascontiguousarray(s["vvel"][0])
    sig = [np.ascontiguousarray(s[k][0]) for k in synth.SIG_NAMES]
    strength = np.ascontiguousarray(2.0e4 * s["aice"][0])
    diag = {k: np.zeros((ny, nx)) for k in ("shear", "divu", "prs_sig", "rdg_conv", "rdg_shear")}
    str8 = np.zeros((8, ny, nx))
    aiu = np.ascontiguousarray(np.maximum(s["aice"][0], 0.05))
    umd = np.ascontiguousarray(300.0 * aiu / 30.0); fm = np.ascontiguousarray(1e-4 * 300.0 * aiu)
    io = [np.zeros((ny, nx)) for _ in range(4)]
    exchange(c, dom, [u.reshape(-1), v.reshape(-1), strength.reshape(-1)], rank, world)
    for ksub in range(1, NSUB + 1):
        orc.stress(ksub, icellt, ti, tj, u, v, g, strength, sig, diag, str8)
        orc.stepu(icellu, ui, uj, aiu, str8, np.ascontiguousarray(s["uocn"][0]), np.ascontiguousarray(s["vocn"][0]),
                  np.ascontiguousarray(s["uocn"][0]), np.ascontiguousarray(s["vocn"][0]),
                  np.ascontiguousarray(s["strairxT"][0]), np.ascontiguousarray(s["strairyT"][0]), umd, fm,
                  g["uarear"], *io, u, v)
        exchange(c, dom, [u.reshape(-1), v.reshape(-1)], rank, world)
    j0 = int(dom["j0"][0])
    np.savez(os.path.join(outdir, f"r{world}_{rank}.npz"), u=u[1:-1, 1:-1], v=v[1:-1, 1:-1],
             s0=sig[0][1:-1, 1:-1], j0=j0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_slabs_reproduce_one_domain(tmp_path):
    outdir = str(tmp_path)
    for world in (1, 2):
        mp.spawn(run_rank, args=(world, free_port(), outdir), nprocs=world, join=True)
    one = np.load(os.path.join(outdir, "r1_0.npz"))
    parts = [np.load(os.path.join(outdir, f"r2_{r}.npz")) for r in range(2)]
    for k in ("u", "v", "s0"):
        two = np.concatenate([p[k] for p in sorted(parts, key=lambda p: int(p["j0"]))], axis=0)
        assert np.array_equal(two, one[k]), k
    assert np.abs(one["u"]).max() > 1e-3
