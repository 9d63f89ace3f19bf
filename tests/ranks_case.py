"""R ranks of one job as R contexts of THIS process on the one GPU, one host thread each (the GPU boxes admit at most six
processes on the card, so an 8-rank job cannot be eight processes there): messages through the in-process link
(cice_comm_init_local: pack kernel -> host mailbox -> unpack kernel), the cross-rank one-launch loop through plain device
pointers.  Used by tests/test_gpu_evp.py (2-3 ranks, small grids) and tests/test_gpu_fullsize.py (8 ranks: BASELINE.json
configs[3] and configs[4], /root/reference/mpi/ice_boundary.F90:1028-1417 is what the exchange replaces)."""
import os
import threading

import numpy as np

from cice4_amd import lib, synth

_LINK = [5000]


def owned(dom, f):
    """global physical field from the OWNED rows of (possibly overlapping) slab blocks"""
    g = np.zeros((dom["nyg"], dom["nxg"]))
    for b in range(dom["nblocks"]):
        r0 = dom["j0"][b] + (dom["own_jlo"][b] - dom["jlo"][b])
        nr = dom["own_jhi"][b] - dom["own_jlo"][b] + 1
        g[r0:r0 + nr, :] = f[b, dom["own_jlo"][b] - 1:dom["own_jhi"][b], dom["ilo"][b] - 1:dom["ihi"][b]]
    return g


def run_ranks(gg, R, mode, ndte, dt, ns=0, seed=31, cover="patchy", overlap=0, skew_k=0, split=None, strength_args=None,
              min_cells=None, timeout=600, info=None, npx=1, blocks=(1, 1), block_map=None):
    """Run evp(dt) on R ranks (threads) and return [(dom, state)] per rank.
    mode: 'classic' (one block per rank, ghost cells after every subcycle), 'peer' (the whole loop in one launch per rank,
    neighbours' exchange copies mapped), 'slabs' (wide-halo slabs with `overlap` rows; skew_k > 0: K-subcycle sweeps
    between the refreshes).  npx: task columns of the cartesian layout (classic / peer; R / npx task rows): 1 = j-slabs;
    blocks = (bx, by): every task holds bx x by blocks of its part of the grid.  block_map = (bsx, bsy, owner): any block -> rank
    map instead (owner[g] = rank of global block g, -1: an eliminated land block; cice_domain_create_map).
    info: optional dict that receives what rank 0 reports (evp_get_info)."""
    nxg, nyg = gg["nxg"], gg["nyg"]
    _LINK[0] += 1
    link = _LINK[0]
    bar = threading.Barrier(R)
    exports, out, errs = [None] * R, [None] * R, []

    def rank_fn(r):
        try:
            c = lib.Context(device=0); c.sync()
            # every rank's MAIN stream first, one after the other: the runtime deals its streams to the hardware queues in
            # the order they are created, and two loops that wait for each other must not share a queue (a copy stream that
            # another rank creates in between would shift the count)
            bar.wait(timeout=120)
            if mode == "slabs":
                dom = c.domain_create_slabs(nxg, nyg, R, ew=1, ns=0, rank=r, nranks=R, overlap=overlap)
            elif block_map is not None:
                bsx, bsy, owner = block_map
                dom = c.domain_create_map(nxg, nyg, bsx, bsy, owner, ew=1, ns=ns, rank=r, nranks=R)
            else:
                npy = R // npx
                assert npx * npy == R and nxg % (npx * blocks[0]) == 0 and nyg % (npy * blocks[1]) == 0
                dom = c.domain_create(nxg, nyg, nxg // (npx * blocks[0]), nyg // (npy * blocks[1]), ew=1, ns=ns, rank=r, npx=npx, npy=npy)
            assert (block_map is not None or dom["nblocks"] == blocks[0] * blocks[1]) and dom["nsend"] >= 1
            c.comm_init_local(link, r, R)
            grid = synth.block_fields(gg, dom, ew_cyclic=True, north_ocean=True) if ns in (3, 4) else synth.block_fields(gg, dom, ns_cyclic=(ns == 1))
            s = synth.evp_state(grid, dom, seed=seed, cover=cover)
            kw = dict(krdg_partic=0, krdg_redist=0) if strength_args is None else strength_args
            c.evp_init(grid, ndte=ndte, **kw)
            if mode == "peer":
                c.evp_set_option("resident_peer_share", R)
                exports[r] = c.evp_peer_export()
                bar.wait(timeout=120)
                if npx == 1 and R <= 3 and blocks == (1, 1) and block_map is None:   # the older call: by side (0 = the rank to the south, 1 = to the north)
                    if r > 0 or ns == 1:
                        c.evp_peer_connect(0, exports[(r - 1) % R])
                    if r < R - 1 or ns == 1:
                        c.evp_peer_connect(1, exports[(r + 1) % R])
                else:                            # any cartesian layout: by rank
                    nbrs = c.evp_peer_ranks()
                    assert r not in nbrs and 1 <= len(nbrs) <= 8
                    for nr in nbrs:
                        c.evp_peer_connect_rank(nr, exports[nr])
                assert c.evp_get_info("resident_peer") == 1
                assert c.evp_get_info("resident_peer_fine") == (0 if os.environ.get("CICE4_AMD_PEER_COARSE") == "1" else 1)
                bar.wait(timeout=120)
            else:
                c.evp_set_option("resident", 0)
                if skew_k:
                    if min_cells is not None:
                        c.evp_set_option("skew_min_cells", min_cells)
                    c.evp_set_option("skew_levels", skew_k)
                    assert c.evp_get_info("skew") == 1
                    if split is not None:
                        c.evp_set_option("skew_split", split)
                    assert c.evp_get_info("skew_trim_ext") == 1
                else:
                    c.evp_set_option("skew", 0)
            if mode == "peer":
                # R loops on ONE device wait for each other, and a copy stream of one rank may share a hardware queue with the
                # main stream of another: no rank's loop may start while another rank's uploads are still queued (on a node
                # every rank has a device, and queues, of its own).  Same entry points as cice_evp, a barrier in between.
                c.evp_upload(s)
                bar.wait(timeout=120)
                c.evp_step(dt)
                bar.wait(timeout=timeout)
                c.evp_download(s)
                assert c.evp_get_info("resident_peer") == 1, "the cross-rank loop timed out and fell back"
            else:
                c.evp(dt, s)
            if info is not None and r == 0:
                for k in ("fused", "skew", "skew_levels", "last_launches"):
                    info[k] = c.evp_get_info(k)
            out[r] = (dom, s)
            bar.wait(timeout=timeout)        # nobody frees buffers a neighbour may still be writing to
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))
            bar.abort()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    return out


def assemble_blocks(out, key, nxg, nyg):
    """every rank's block (classic / peer layouts: no overlap rows) of one field, as one global array"""
    g = np.zeros((nyg, nxg))
    for dom, s in out:
        for b in range(dom["nblocks"]):
            ni = dom["ihi"][b] - dom["ilo"][b] + 1; nj = dom["jhi"][b] - dom["jlo"][b] + 1
            g[dom["j0"][b]:dom["j0"][b] + nj, dom["i0"][b]:dom["i0"][b] + ni] = s[key][b, dom["jlo"][b] - 1:dom["jhi"][b],
                                                                                      dom["ilo"][b] - 1:dom["ihi"][b]]
    return g


def assemble(out, key, nxg, nyg):
    """every rank's owned rows of one field, as one global array"""
    got = np.zeros((nyg, nxg))
    for dom, s in out:
        part = owned(dom, s[key])
        rows = slice(int(dom["j0"][0] + dom["own_jlo"][0] - dom["jlo"][0]),
                     int(dom["j0"][0] + dom["own_jhi"][0] - dom["jlo"][0]) + 1)
        got[rows] = part[rows]
    return got
