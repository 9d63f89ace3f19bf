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
    """ghost update of `fields` (flat views): on-rank copies + p2p messages in the library's order
    (Halo::update: wrap list, pack, send/recv, on-rank refresh rows, unpack)."""
    sends = c.halo_msgs(0); recvs = c.halo_msgs(1)
    reqs, bufs = [], []
    for f in fields:          # the wrap list first: whole rows are sent INCLUDING their E/W ghost columns
        f[dom["hdst"]] = f[dom["hsrc"]]
    for peer, addr in recvs:
        b = torch.empty(len(fields) * len(addr), dtype=torch.float64)
        bufs.append((addr, b)); reqs.append(dist.irecv(b, src=peer))
    for peer, addr in sends:
        t = torch.from_numpy(np.concatenate([f[addr] for f in fields]))
        reqs.append(dist.isend(t, dst=peer))
    for r in reqs:
        r.wait()
    if len(dom.get("rsrc", ())):
        for f in fields:
            f[dom["rdst"]] = f[dom["rsrc"]]
    for addr, b in bufs:
        v = b.numpy().reshape(len(fields), len(addr))
        for k, f in enumerate(fields):
            f[addr] = v[k]


def run_rank(rank, world, port, outdir, overlap=-1):
    """overlap < 0: classic decomposition (cice_domain_create), ghost rows of u, v after every subcycle.
    overlap = H >= 0: the decomposition bench.py --gpus N uses (cice_domain_create_slabs(..., overlap=H)):
    every slab extended by H rows that are recomputed; u, v and the 12 stresses of the outer rows refreshed
    from their owner in ONE 14-field message per neighbour every H subcycles and after the last one
    (Evp::launch_subcycle), the E-W wrap every subcycle."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cice4_amd import lib, synth
    from oracle import oracle
    orc = oracle.Oracle()
    orc.set_evp_parameters(DT, NDTE)
    c = lib.Context()
    if overlap >= 0:
        dom = c.domain_create_slabs(NXG, NYG, world, ew=1, ns=0, rank=rank, nranks=world, overlap=overlap)
        assert dom["nblocks"] == 1 and (world == 1 or dom["nsend"] >= 1)
    else:
        dom = c.domain_create(NXG, NYG, NXG, NYG // world, ew=1, ns=0, rank=rank, npx=1, npy=world)
    grid, s = prepared_case(dom, rank, world)
    ny, nx = dom["ny"], dom["nx"]
    # per-block prepared inputs of the subcycle loop (masks/lists from a fixed rule, not evp_prep)
    # T-cells on jlo..jhi+1, U-cells on jlo..jhi of the block's physical extent (evp_prep2, :850-859); a
    # wide-halo slab is padded to a common height, rows beyond jhi+1 take no part
    jlo, jhi = int(dom["jlo"][0]), int(dom["jhi"][0])
    tm = (s["aice"][0] > 0.01) & (grid["tmask"][0] > 0); tm[:jlo - 1, :] = False; tm[jhi + 1:, :] = False; tm[:, 0] = False
    um = (grid["umask"][0] > 0) & (s["aice"][0] > 0.01); um[:jlo - 1, :] = False; um[jhi:, :] = False
    um[:, 0] = um[:, -1] = False
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
    for ksub in range(1, NSUB + 1) if overlap <= 0 else ():
        orc.stress(ksub, icellt, ti, tj, u, v, g, strength, sig, diag, str8)
        orc.stepu(icellu, ui, uj, aiu, str8, np.ascontiguousarray(s["uocn"][0]), np.ascontiguousarray(s["vocn"][0]),
                  np.ascontiguousarray(s["uocn"][0]), np.ascontiguousarray(s["vocn"][0]),
                  np.ascontiguousarray(s["strairxT"][0]), np.ascontiguousarray(s["strairyT"][0]), umd, fm,
                  g["uarear"], *io, u, v)
        exchange(c, dom, [u.reshape(-1), v.reshape(-1)], rank, world)
    uocn = np.ascontiguousarray(s["uocn"][0]); vocn = np.ascontiguousarray(s["vocn"][0])
    sax = np.ascontiguousarray(s["strairxT"][0]); say = np.ascontiguousarray(s["strairyT"][0])
    for ksub in range(1, NSUB + 1) if overlap > 0 else ():
        orc.stress(ksub, icellt, ti, tj, u, v, g, strength, sig, diag, str8)
        orc.stepu(icellu, ui, uj, aiu, str8, uocn, vocn, uocn, vocn, sax, say, umd, fm, g["uarear"], *io, u, v)
        if ksub % overlap == 0 or ksub == NSUB:       # wide-halo refresh: 14 fields, one message per neighbour
            exchange(c, dom, [u.reshape(-1), v.reshape(-1)] + [x.reshape(-1) for x in sig], rank, world)
        else:                                         # E-W wrap only (the kernel writes these ghosts itself)
            for f in (u.reshape(-1), v.reshape(-1)):
                f[dom["hdst"]] = f[dom["hsrc"]]
    j0 = int(dom["j0"][0])
    if overlap >= 0:      # owned rows of the extended slab
        r0, r1 = int(dom["own_jlo"][0]) - 1, int(dom["own_jhi"][0])
        j0 += int(dom["own_jlo"][0] - dom["jlo"][0])
    else:
        r0, r1 = 1, ny - 1
    tag = f"h{overlap}_" if overlap >= 0 else ""
    np.savez(os.path.join(outdir, f"{tag}r{world}_{rank}.npz"), u=u[r0:r1, 1:-1], v=v[r0:r1, 1:-1],
             s0=sig[0][r0:r1, 1:-1], s11=sig[11][r0:r1, 1:-1], j0=j0)
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [4, 3])
def test_two_wide_halo_slabs_reproduce_one_domain(tmp_path, overlap):
    """The decomposition `bench.py --gpus N` runs (domain_create_slabs(..., overlap=H)), world_size 2 over gloo:
    owned rows equal the one-domain run bit for bit (H = 4: pairs of subcycles between refreshes on the GPU;
    H = 3: odd)."""
    outdir = str(tmp_path)
    mp.spawn(run_rank, args=(1, free_port(), outdir, -1), nprocs=1, join=True)
    mp.spawn(run_rank, args=(2, free_port(), outdir, overlap), nprocs=2, join=True)
    one = np.load(os.path.join(outdir, "r1_0.npz"))
    parts = sorted([np.load(os.path.join(outdir, f"h{overlap}_r2_{r}.npz")) for r in range(2)], key=lambda p: int(p["j0"]))
    assert int(parts[0]["j0"]) == 0 and int(parts[1]["j0"]) == NYG // 2
    for k in ("u", "v", "s0", "s11"):
        two = np.concatenate([p[k] for p in parts], axis=0)
        assert np.array_equal(two, one[k]), k
