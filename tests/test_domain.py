"""Host-side block decomposition / halo topology (cice4_amd/csrc/domain.cpp) against a
brute-force global-index model, for one and several ranks, cyclic / open / closed edges and
block sizes that do not divide the grid."""
import numpy as np
import pytest

from cice4_amd import lib


def brute_halo(nxg, nyg, bsx, bsy, ew, ns, dom):
    """Expected ghost values for a field holding each cell's global id."""
    nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
    f = -np.ones((nb, ny, nx), np.int64)
    for b in range(nb):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                ig = dom["i0"][b] + (i - dom["ilo"][b]); jg = dom["j0"][b] + (j - dom["jlo"][b])
                phys = dom["ilo"][b] <= i <= dom["ihi"][b] and dom["jlo"][b] <= j <= dom["jhi"][b]
                if not phys:
                    if not (dom["ilo"][b] - 1 <= i <= dom["ihi"][b] + 1 and dom["jlo"][b] - 1 <= j <= dom["jhi"][b] + 1):
                        continue
                    if ig < 0 or ig >= nxg:
                        if ew != 1:
                            continue
                        ig %= nxg
                    if jg < 0 or jg >= nyg:
                        if ns != 1:
                            continue
                        jg %= nyg
                f[b, j - 1, i - 1] = jg * nxg + ig
    return f


@pytest.mark.parametrize("nxg,nyg,bsx,bsy,ew,ns", [(24, 20, 24, 20, 1, 0), (24, 20, 12, 10, 1, 0),
                                                   (24, 20, 8, 5, 1, 1), (25, 19, 8, 5, 1, 0),
                                                   (24, 20, 12, 10, 0, 0), (24, 20, 6, 20, 2, 2)])
def test_local_halo_lists(nxg, nyg, bsx, bsy, ew, ns):
    c = lib.Context()
    dom = c.domain_create(nxg, nyg, bsx, bsy, ew=ew, ns=ns)
    want = brute_halo(nxg, nyg, bsx, bsy, ew, ns, dom)
    f = want.copy()
    ghost = np.ones_like(f, bool)
    for b in range(dom["nblocks"]):
        ghost[b, dom["jlo"][b] - 1:dom["jhi"][b], dom["ilo"][b] - 1:dom["ihi"][b]] = False
    f[ghost] = -1
    f = f.reshape(-1)
    # sources are physical cells, destinations ghost cells, no destination twice
    assert not ghost.reshape(-1)[dom["hsrc"]].any() and ghost.reshape(-1)[dom["hdst"]].all()
    assert len(np.unique(dom["hdst"])) == len(dom["hdst"])
    f[dom["hdst"]] = f[dom["hsrc"]]
    assert np.array_equal(f.reshape(want.shape), want)


@pytest.mark.parametrize("npx,npy", [(1, 2), (2, 2), (1, 4), (2, 1)])
def test_multi_rank_messages_reproduce_single_rank_halo(npx, npy):
    nxg, nyg, bsx, bsy = 24, 20, 6, 5
    nr = npx * npy
    ctxs = [lib.Context() for _ in range(nr)]
    doms = [c.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0, rank=r, npx=npx, npy=npy) for r, c in enumerate(ctxs)]
    assert sum(d["nblocks"] for d in doms) == doms[0]["nblocks_tot"]
    fields, wants = [], []
    for d in doms:
        w = brute_halo(nxg, nyg, bsx, bsy, 1, 0, d)
        f = w.copy()
        for b in range(d["nblocks"]):
            g = np.ones(f[b].shape, bool); g[d["jlo"][b] - 1:d["jhi"][b], d["ilo"][b] - 1:d["ihi"][b]] = False
            f[b][g] = -1
        fields.append(f.reshape(-1)); wants.append(w)
    sends = [dict(c.halo_msgs(0)) for c in ctxs]
    recvs = [dict(c.halo_msgs(1)) for c in ctxs]
    for r in range(nr):
        for peer, addr in recvs[r].items():
            src = sends[peer][r]
            assert len(src) == len(addr)
            fields[r][addr] = fields[peer][src]      # both ends list elements in the same order
    for r, d in enumerate(doms):
        fields[r][d["hdst"]] = fields[r][d["hsrc"]]
        assert np.array_equal(fields[r].reshape(wants[r].shape), wants[r]), r
    # 1xN j-slabs: two peers at most, full rows incl. E/W ghost columns are NOT needed
    if npx == 1:
        for r in range(nr):
            assert len(sends[r]) <= 2


@pytest.mark.parametrize("ns", [3, 4])
@pytest.mark.parametrize("nb,nr,ov", [(2, 2, 0), (2, 2, 4), (3, 3, 6), (4, 2, 2)])
def test_slabs_under_a_tripole_fold(nb, nr, ov, ns):
    """A folded grid cut into (wide-halo) slabs: only the rank with the top slab has fold lists, and applied to the slab's
    array they leave its top row and the ghost row beyond exactly as the one-block domain's lists leave them -- for every
    field location, scalars and vectors.  Slabs whose overlap would reach the fold rows are refused."""
    nxg, nyg = 16, 48
    rng = np.random.default_rng(5 + ns)
    g = rng.standard_normal((nyg, nxg))
    one = lib.Context()
    d1 = one.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    for r in range(nr):
        c = lib.Context()
        d = c.domain_create_slabs(nxg, nyg, nb, ew=1, ns=ns, rank=r, nranks=nr, overlap=ov)
        top = [b for b in range(d["nblocks"]) if d["j0"][b] + (d["jhi"][b] - d["jlo"][b]) == nyg - 1]
        assert (len(c.domain_list("fold_lsrc")) > 0) == bool(top), r
        if not top:
            continue
        b = top[0]
        for loc in (1, 2, 3, 4):
            for kind in (1, 2):
                a1 = np.zeros((1, d1["ny"], d1["nx"])); a1[0, 1:-1, 1:-1] = g
                one.apply_halo_lists(a1, loc=loc, kind=kind)
                a = np.zeros((d["nblocks"], d["ny"], d["nx"]))
                for bb in range(d["nblocks"]):
                    rows = d["jhi"][bb] - d["jlo"][bb] + 1
                    a[bb, d["jlo"][bb] - 1:d["jhi"][bb], 1:-1] = g[d["j0"][bb]:d["j0"][bb] + rows]
                c.apply_halo_lists(a, loc=loc, kind=kind)
                jt = d["jhi"][b]                      # 1-based top physical row; the ghost row is jt + 1
                assert np.array_equal(a[b, jt - 3:jt + 1], a1[0, -4:]), (r, loc, kind)
    with pytest.raises(lib.CiceError):
        lib.Context().domain_create_slabs(nxg, nyg, 2, ew=1, ns=ns, rank=0, nranks=2, overlap=nyg // 2 - 3)
    with pytest.raises(lib.CiceError):
        lib.Context().domain_create_slabs(nxg, nyg, 2, ew=0, ns=ns, rank=0, nranks=2, overlap=2)


@pytest.mark.parametrize("nb,nr,ov", [(4, 1, 0), (4, 1, 3), (4, 2, 3), (8, 4, 2), (2, 2, 10)])
def test_slab_overlap_lists(nb, nr, ov):
    """Wide-halo slabs: after wrap + refresh every cell of every extended block (and its ghost
    ring inside the domain) holds the value of the global cell it mirrors."""
    nxg, nyg = 12, 40
    ctxs = [lib.Context() for _ in range(nr)]
    doms = [c.domain_create_slabs(nxg, nyg, nb, ew=1, ns=0, rank=r, nranks=nr, overlap=ov) for r, c in enumerate(ctxs)]
    fields, wants = [], []
    for d in doms:
        nbk, ny, nx = d["nblocks"], d["ny"], d["nx"]
        w = -np.ones((nbk, ny, nx), np.int64); f = -np.ones((nbk, ny, nx), np.int64)
        for b in range(nbk):
            for j in range(1, ny + 1):
                jg = d["j0"][b] + (j - d["jlo"][b])
                if not (d["jlo"][b] - 1 <= j <= d["jhi"][b] + 1) or jg < 0 or jg >= nyg:
                    continue
                for i in range(1, nx + 1):
                    ig = (i - 2) % nxg
                    w[b, j - 1, i - 1] = jg * nxg + ig
                    if d["own_jlo"][b] <= j <= d["own_jhi"][b] and 2 <= i <= nx - 1:
                        f[b, j - 1, i - 1] = jg * nxg + ig
        fields.append(f.reshape(-1)); wants.append(w)
    # the order the device uses: wrap is done by the producing kernel, then refresh
    for r, d in enumerate(doms):
        fields[r][d["hdst"]] = fields[r][d["hsrc"]]
    sends = [dict(c.halo_msgs(0)) for c in ctxs]; recvs = [dict(c.halo_msgs(1)) for c in ctxs]
    staged = [{p: fields[p][sends[p][r]].copy() for p in recvs[r]} for r in range(nr)]
    for r, d in enumerate(doms):
        fields[r][d["rdst"]] = fields[r][d["rsrc"]]
        for p, addr in recvs[r].items():
            fields[r][addr] = staged[r][p]
        fields[r][d["hdst"]] = np.where(fields[r][d["hdst"]] < 0, fields[r][d["hsrc"]], fields[r][d["hdst"]])
        got = fields[r].reshape(wants[r].shape)
        for b in range(d["nblocks"]):     # every extended physical row is complete
            rows = slice(d["jlo"][b] - 1, d["jhi"][b])
            assert np.array_equal(got[b, rows], wants[r][b, rows]), (r, b)
            for jj in (d["jlo"][b] - 2, d["jhi"][b]):       # ghost rows inside the domain
                if (wants[r][b, jj] >= 0).any():
                    assert np.array_equal(got[b, jj], wants[r][b, jj]), (r, b, jj)


def _halo_multi_rank(ctxs, fields, loc, kind, fill, nxg):
    """Emulates Halo::update for several ranks on host arrays (one flat float64 array per rank): wrap list, fill
    list, ghost messages, then the tripole fold with its own messages into each top-row rank's buffer."""
    nr = len(ctxs)
    doms = [c.domain() for c in ctxs]
    sends = [dict(c.halo_msgs(0)) for c in ctxs]; recvs = [dict(c.halo_msgs(1)) for c in ctxs]
    staged = [{p: fields[p][sends[p][r]].copy() for p in recvs[r]} for r in range(nr)]
    for r in range(nr):
        f = fields[r]
        f[doms[r]["hdst"]] = f[doms[r]["hsrc"]]
        f[doms[r]["hfill"]] = fill
        for p, addr in recvs[r].items():
            f[addr] = staged[r][p]
    fs = [dict(c.halo_msgs(2)) for c in ctxs]; fr = [dict(c.halo_msgs(3)) for c in ctxs]
    fstaged = [{p: fields[p][fs[p][r]].copy() for p in fr[r]} for r in range(nr)]
    sgn = 1.0 if kind == 1 else -1.0
    for r, c in enumerate(ctxs):
        lsrc, bidx = c.domain_list("fold_lsrc"), c.domain_list("fold_bidx")
        dst, src = c.domain_list("fold_dst", loc), c.domain_list("fold_src", loc)
        if not len(dst):
            assert not fr[r]
            continue
        buf = np.full(2 * nxg, fill, np.float64)
        buf[bidx] = fields[r][lsrc]
        for p, b in fr[r].items():
            buf[b] = fstaged[r][p]
        lo, hi = c.domain_list("fold_lo", loc), c.domain_list("fold_hi", loc)
        x = 0.5 * (buf[lo] + sgn * buf[hi])
        buf[lo] = x; buf[hi] = sgn * x
        fields[r][dst] = sgn * buf[src]


@pytest.mark.parametrize("ns", [0, 3])
@pytest.mark.parametrize("nr,seed", [(2, 1), (3, 2), (4, 3)])
def test_any_block_map_tripole_and_eliminated_blocks_on_several_ranks(nr, seed, ns):
    """cice_domain_create_map: a RANDOM block->task map (what rake / space-curve distributions produce) with
    eliminated blocks, on nr ranks, open or tripole north boundary: ghost messages, fill cells and fold messages
    give every rank the ghost values the single-rank domain with the same eliminated blocks has."""
    nxg, nyg, bsx, bsy = 24, 20, 6, 5
    nbx, nby = 4, 4
    rng = np.random.default_rng(seed)
    owner = rng.integers(0, nr, nbx * nby).astype(np.int32)
    owner[rng.choice(nbx * nby - nbx, 3, replace=False)] = -1          # eliminated (never in the top block row)
    owner[nbx * nby - 1] = 0; owner[nbx * nby - 2] = nr - 1             # top row spread over ranks
    one = lib.Context()
    own1 = np.where(owner >= 0, 0, -1).astype(np.int32)
    d1 = one.domain_create_map(nxg, nyg, bsx, bsy, own1, ew=1, ns=ns)
    ctxs = [lib.Context() for _ in range(nr)]
    doms = [c.domain_create_map(nxg, nyg, bsx, bsy, owner, ew=1, ns=ns, rank=r, nranks=nr) for r, c in enumerate(ctxs)]
    assert sum(d["nblocks"] for d in doms) == d1["nblocks"] == int((owner >= 0).sum())
    np_ = d1["ny"] * d1["nx"]
    for loc in ((1, 2, 3, 4) if ns == 3 else (1,)):
        for kind in ((1, 2) if ns == 3 else (1,)):
            g = rng.uniform(-1, 1, (d1["nblocks"], d1["ny"], d1["nx"]))
            want = g.copy()
            one.apply_halo_lists(want, loc, kind, fill=7.0)
            fields = []
            for d in doms:
                f = np.zeros((d["nblocks"], d["ny"], d["nx"]))
                for lb, gid in enumerate(d["gid"]):
                    f[lb] = g[list(d1["gid"]).index(gid)]
                fields.append(f.reshape(-1))
            _halo_multi_rank(ctxs, fields, loc, kind, 7.0, nxg)
            for d, f in zip(doms, fields):
                f = f.reshape(d["nblocks"], d["ny"], d["nx"])
                for lb, gid in enumerate(d["gid"]):
                    assert np.array_equal(f[lb], want[list(d1["gid"]).index(gid)]), (nr, ns, loc, kind, gid)
            assert not np.array_equal(want, g)
