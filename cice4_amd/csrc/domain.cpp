#include "domain.h"

#include <algorithm>
#include <map>

namespace cice {

namespace {

// global index of a ghost position, or -1 when it falls outside an open/closed edge
inline int wrap(int g, int n, int bnd) {
  if (g >= 0 && g < n) return g;
  if (bnd != BND_CYCLIC) return -1;
  return (g + n) % n;
}

}  // namespace

const char* Domain::create(int nx_global, int ny_global, int block_size_x, int block_size_y,
                           int ew_bnd, int ns_bnd, int rank_, int npx_, int npy_) {
  if (nx_global < 1 || ny_global < 1) return "domain size < 1";
  if (block_size_x < 1 || block_size_y < 1) return "block size < 1";
  if (npx_ < 1 || npy_ < 1) return "process grid < 1";
  nxg = nx_global; nyg = ny_global; bsx = block_size_x; bsy = block_size_y;
  nx_block = bsx + 2; ny_block = bsy + 2;
  nbx = (nxg - 1) / bsx + 1;  // ice_blocks.F90:158-160
  nby = (nyg - 1) / bsy + 1;
  npx = npx_; npy = npy_; nranks = npx * npy; rank = rank_;
  ew = ew_bnd; ns = ns_bnd;
  if (rank < 0 || rank >= nranks) return "rank out of range";
  if (npx > nbx || npy > nby) return "more ranks than blocks along an axis";
  all.clear(); local.clear(); hsrc.clear(); hdst.clear(); send.clear(); recv.clear();
  rsrc.clear(); rdst.clear(); overlap = 0;

  std::vector<int> nlocal(nranks, 0);
  for (int jb = 0; jb < nby; ++jb)
    for (int ib = 0; ib < nbx; ++ib) {
      Block b;
      b.gid = jb * nbx + ib; b.ib = ib; b.jb = jb;
      b.i0 = ib * bsx; b.j0 = jb * bsy;
      b.ilo = 2; b.jlo = 2;
      b.ihi = 1 + std::min(bsx, nxg - b.i0);  // padded last block: ice_blocks.F90:171-178
      b.jhi = 1 + std::min(bsy, nyg - b.j0);
      // contiguous rectangles of blocks per rank, dealt the way create_distrb_cart does
      // (ice_distribution.F90:719-732): ceil(nblocks / nprocs) block columns (rows) per process
      // column (row), the last ones taking what is left -- possibly nothing
      const int per_x = (nbx - 1) / npx + 1, per_y = (nby - 1) / npy + 1;
      const int px = ib / per_x, py = jb / per_y;
      b.owner = py * npx + px;
      b.local_id = nlocal[b.owner]++;
      b.own_jlo = b.jlo; b.own_jhi = b.jhi;
      all.push_back(b);
    }
  for (const Block& b : all)
    if (b.owner == rank) local.push_back(b.gid);

  const long long np = (long long)nx_block * ny_block;
  if (np * (long long)std::max<size_t>(local.size(), 1) > 0x7fffffffLL) return "local array too large for int32 addressing";
  auto addr = [&](const Block& b, int i, int j) {  // 1-based (i,j)
    return (int32_t)((long long)b.local_id * np + (long long)(j - 1) * nx_block + (i - 1));
  };

  std::map<int, HaloMsg> smap, rmap;
  // Visit every ghost cell of every block in one global order; both ends of a message
  // therefore agree on element order.
  for (const Block& d : all) {
    for (int j = d.jlo - 1; j <= d.jhi + 1; ++j)
      for (int i = d.ilo - 1; i <= d.ihi + 1; ++i) {
        const bool ghost = (i < d.ilo || i > d.ihi || j < d.jlo || j > d.jhi);
        if (!ghost) continue;
        int ig = wrap(d.i0 + (i - d.ilo), nxg, ew);
        int jg = wrap(d.j0 + (j - d.jlo), nyg, ns);
        if (ig < 0 || jg < 0) continue;  // beyond an open/closed edge: never written
        const Block& s = all[(jg / bsy) * nbx + (ig / bsx)];
        int is = s.ilo + (ig - s.i0), js = s.jlo + (jg - s.j0);
        if (d.owner == rank && s.owner == rank && !(self_comm && s.gid != d.gid)) {
          hsrc.push_back(addr(s, is, js));
          hdst.push_back(addr(d, i, j));
        } else if (d.owner == rank && s.owner == rank) {  // self_comm: both ends are this rank
          HaloMsg& mr = rmap[rank]; mr.peer = rank;
          mr.addr.push_back(addr(d, i, j));
          HaloMsg& ms = smap[rank]; ms.peer = rank;
          ms.addr.push_back(addr(s, is, js));
        } else if (d.owner == rank) {
          HaloMsg& m = rmap[s.owner]; m.peer = s.owner;
          m.addr.push_back(addr(d, i, j));
        } else if (s.owner == rank) {
          HaloMsg& m = smap[d.owner]; m.peer = d.owner;
          m.addr.push_back(addr(s, is, js));
        }
      }
  }
  for (auto& kv : smap) send.push_back(std::move(kv.second));
  for (auto& kv : rmap) recv.push_back(std::move(kv.second));
  return "";
}

const char* Domain::create_slabs(int nx_global, int ny_global, int nblocks_y, int ew_bnd, int ns_bnd,
                                 int rank_, int nranks_, int overlap_rows) {
  if (nx_global < 1 || ny_global < 1) return "domain size < 1";
  if (nblocks_y < 1 || nranks_ < 1 || nblocks_y % nranks_) return "nblocks_y must be a positive multiple of nranks";
  if (ny_global % nblocks_y) return "ny_global must be divisible by nblocks_y";
  if (rank_ < 0 || rank_ >= nranks_) return "rank out of range";
  if (overlap_rows < 0) return "overlap < 0";
  if (ns_bnd == BND_CYCLIC) return "cyclic north-south boundary is not supported with slabs";
  nxg = nx_global; nyg = ny_global; bsx = nxg; bsy = nyg / nblocks_y;
  if (overlap_rows > bsy) return "overlap larger than a slab";
  overlap = overlap_rows;
  nx_block = bsx + 2; ny_block = bsy + 2 * overlap + 2;
  nbx = 1; nby = nblocks_y; npx = 1; npy = nranks_; nranks = nranks_; rank = rank_;
  ew = ew_bnd; ns = ns_bnd;
  all.clear(); local.clear(); hsrc.clear(); hdst.clear(); send.clear(); recv.clear();
  rsrc.clear(); rdst.clear();
  const int per_rank = nby / nranks;
  std::vector<int> e0(nby);
  for (int jb = 0; jb < nby; ++jb) {
    Block b;
    b.gid = jb; b.ib = 0; b.jb = jb;
    const int o0 = jb * bsy, o1 = o0 + bsy - 1;
    e0[jb] = std::max(0, o0 - overlap);
    const int e1 = std::min(nyg - 1, o1 + overlap);
    b.i0 = 0; b.j0 = e0[jb];
    b.ilo = 2; b.ihi = 1 + bsx; b.jlo = 2; b.jhi = 1 + (e1 - e0[jb] + 1);
    b.own_jlo = b.jlo + (o0 - e0[jb]); b.own_jhi = b.jlo + (o1 - e0[jb]);
    b.owner = jb / per_rank; b.local_id = jb % per_rank;
    all.push_back(b);
    if (b.owner == rank) local.push_back(b.gid);
  }
  const long long np = (long long)nx_block * ny_block;
  if (np * (long long)local.size() > 0x7fffffffLL) return "local array too large for int32 addressing";
  auto addr = [&](const Block& b, int i, int j) {
    return (int32_t)((long long)b.local_id * np + (long long)(j - 1) * nx_block + (i - 1));
  };
  // E-W wrap of every physical row (every subcycle; folded into the producing kernel)
  if (ew == BND_CYCLIC)
    for (int gid : local) {
      const Block& b = all[gid];
      for (int j = b.jlo; j <= b.jhi; ++j) {
        hsrc.push_back(addr(b, b.ihi, j)); hdst.push_back(addr(b, b.ilo - 1, j));
        hsrc.push_back(addr(b, b.ilo, j)); hdst.push_back(addr(b, b.ihi + 1, j));
      }
    }
  // refresh: every row outside the owned range (overlap rows and the two ghost rows), whole
  // rows including the E/W ghost columns, from the block that owns that global row
  std::map<int, HaloMsg> smap, rmap;
  for (const Block& d : all)
    for (int j = d.jlo - 1; j <= d.jhi + 1; ++j) {
      if (j >= d.own_jlo && j <= d.own_jhi) continue;
      const int jg = d.j0 + (j - d.jlo);
      if (jg < 0 || jg >= nyg) continue;  // beyond the open/closed edge: never written
      const Block& s = all[jg / bsy];
      const int js = s.jlo + (jg - s.j0);
      for (int i = 1; i <= nx_block; ++i) {
        if (d.owner == rank && s.owner == rank && !self_comm) {
          rsrc.push_back(addr(s, i, js)); rdst.push_back(addr(d, i, j));
        } else if (d.owner == rank && s.owner == rank) {
          HaloMsg& mr = rmap[rank]; mr.peer = rank; mr.addr.push_back(addr(d, i, j));
          HaloMsg& ms = smap[rank]; ms.peer = rank; ms.addr.push_back(addr(s, i, js));
        } else if (d.owner == rank) {
          HaloMsg& m = rmap[s.owner]; m.peer = s.owner; m.addr.push_back(addr(d, i, j));
        } else if (s.owner == rank) {
          HaloMsg& m = smap[d.owner]; m.peer = d.owner; m.addr.push_back(addr(s, i, js));
        }
      }
    }
  for (auto& kv : smap) send.push_back(std::move(kv.second));
  for (auto& kv : rmap) recv.push_back(std::move(kv.second));
  return "";
}

}  // namespace cice
