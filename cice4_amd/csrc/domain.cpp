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
  nbx = (nxg - 1) / bsx + 1;  // ice_blocks.F90:158-160
  nby = (nyg - 1) / bsy + 1;
  npx = npx_; npy = npy_; nranks = npx * npy; rank = rank_;
  ew = ew_bnd; ns = ns_bnd;
  from_map = false; map_owner.clear(); map_lid.clear();
  if (rank < 0 || rank >= nranks) return "rank out of range";
  if (npx > nbx || npy > nby) return "more ranks than blocks along an axis";
  // contiguous rectangles of blocks per rank, dealt the way create_distrb_cart does
  // (ice_distribution.F90:719-732): ceil(nblocks / nprocs) block columns (rows) per process
  // column (row), the last ones taking what is left -- possibly nothing
  std::vector<int> owner((size_t)nbx * nby), lid((size_t)nbx * nby), nlocal(nranks, 0);
  const int per_x = (nbx - 1) / npx + 1, per_y = (nby - 1) / npy + 1;
  for (int jb = 0; jb < nby; ++jb)
    for (int ib = 0; ib < nbx; ++ib) {
      const int o = (jb / per_y) * npx + ib / per_x;
      owner[jb * nbx + ib] = o;
      lid[jb * nbx + ib] = nlocal[o]++;
    }
  return build(owner, lid);
}

const char* Domain::create_map(int nx_global, int ny_global, int block_size_x, int block_size_y,
                               int ew_bnd, int ns_bnd, int rank_, int nranks_, const int* owner_in,
                               const int* lid_in) {
  if (nx_global < 1 || ny_global < 1) return "domain size < 1";
  if (block_size_x < 1 || block_size_y < 1) return "block size < 1";
  if (nranks_ < 1 || rank_ < 0 || rank_ >= nranks_) return "rank out of range";
  if (!owner_in) return "NULL block map";
  nxg = nx_global; nyg = ny_global; bsx = block_size_x; bsy = block_size_y;
  nbx = (nxg - 1) / bsx + 1;
  nby = (nyg - 1) / bsy + 1;
  npx = nranks_; npy = 1; nranks = nranks_; rank = rank_;
  ew = ew_bnd; ns = ns_bnd;
  const size_t nb = (size_t)nbx * nby;
  std::vector<int> owner(owner_in, owner_in + nb), lid(nb, -1), nlocal(nranks, 0);
  for (size_t g = 0; g < nb; ++g) {
    if (owner[g] < -1 || owner[g] >= nranks) return "block map: owner out of range";
    if (owner[g] < 0) continue;
    lid[g] = lid_in ? lid_in[g] : nlocal[owner[g]];
    nlocal[owner[g]]++;
  }
  for (size_t g = 0; g < nb; ++g)   // local ids of a rank must be a permutation of 0 .. n-1
    if (owner[g] >= 0 && (lid[g] < 0 || lid[g] >= nlocal[owner[g]])) return "block map: local id out of range";
  from_map = true;
  map_owner = owner;
  map_lid = lid;
  return build(owner, lid);
}

// Tripole fold lists of the blocks as they stand in `all` / `local` (block map or slabs).
void Domain::build_fold() {
  fold = false; fold_lsrc.clear(); fold_bidx.clear(); fold_send.clear(); fold_recv.clear();
  for (int l = 0; l < 4; ++l) {
    fold_out[l].dst.clear(); fold_out[l].src.clear(); fold_lo[l].clear(); fold_hi[l].clear();
  }
  if (!tripole()) return;
  const long long np = (long long)nx_block * ny_block;
  auto addr = [&](const Block& b, int i, int j) {  // 1-based (i,j)
    return (int32_t)((long long)b.local_id * np + (long long)(j - 1) * nx_block + (i - 1));
  };
  // The fold buffer holds the top R physical rows of the whole grid: R = 2 for a fold through U points, 3 for one
  // through T points (tripoleRows).  Ranks that own a block of the top block row each assemble the whole buffer.
  const int R = fold_rows();
  const bool tfold = ns == BND_TRIPOLET;
  std::vector<char> top_rank(nranks, 0);
  for (const Block& b : all)
    if (b.jb == nby - 1 && b.owner >= 0) top_rank[b.owner] = 1;
  fold = top_rank[rank] != 0;
  std::map<int, HaloMsg> fs, fr;
  for (const Block& s : all) {   // 1. top R physical rows of every top-row block -> buffer (:3702-3722)
    if (s.jb != nby - 1 || s.owner < 0) continue;
    for (int r = 0; r < R; ++r)
      for (int i = s.ilo; i <= s.ihi; ++i) {
        const int32_t b = (int32_t)(r * nxg + s.i0 + (i - s.ilo));
        // U-fold: rows jhi-1, jhi.  T-fold: the 'north' message fills the three buffer rows with jhi-2, jhi-1, jhi
        // (:3702-3722), then the 'northeast' / 'northwest' messages of the same block, which kept the U-fold's two
        // rows, overwrite rows 1 and 2 with jhi-1, jhi (:3813-3826, :3859-3872): what the update works on is
        // jhi-1, jhi, jhi -- reproduced here as the reference runs, not as its comments describe it
        const int j = tfold ? s.jhi - 1 + std::min(r, 1) : s.jhi - 1 + r;
        if (s.owner == rank) {
          fold_lsrc.push_back(addr(s, i, j));
          fold_bidx.push_back(b);
          for (int p = 0; p < nranks; ++p)
            if (p != rank && top_rank[p]) {
              HaloMsg& m = fs[p]; m.peer = p; m.addr.push_back(addr(s, i, j));
            }
        } else if (fold) {
          HaloMsg& m = fr[s.owner]; m.peer = s.owner; m.addr.push_back(b);
        }
      }
  }
  for (auto& kv : fs) fold_send.push_back(std::move(kv.second));
  for (auto& kv : fr) fold_recv.push_back(std::move(kv.second));
  if (fold) {
    // 2. symmetry of the degenerate top row of the buffer (serial/ice_boundary.F90:725-823); 1-based i as there.
    // U-fold: NE-corner and N-face fields lie on the fold; T-fold: centre and E-face fields do.
    const int top = (R - 1) * nxg;
    auto pair = [&](int loc, int i, int idst) {
      fold_lo[loc - 1].push_back(top + i - 1);
      fold_hi[loc - 1].push_back(top + idst - 1);
    };
    if (tfold) {
      for (int i = 2; i <= nxg / 2; ++i) pair(LOC_CENTER, i, nxg - i + 2);      // :735-743
      for (int i = 1; i <= nxg / 2; ++i) pair(LOC_EFACE, i, nxg + 1 - i);       // :757-765
    } else {
      for (int i = 1; i <= nxg / 2 - 1; ++i) pair(LOC_NECORNER, i, nxg - i);    // :792-800
      for (int i = 1; i <= nxg / 2; ++i) pair(LOC_NFACE, i, nxg + 1 - i);       // :814-822
    }
    // 3. copy out (:3726-3750 list, :831-866 offsets): rows jhi (jj = 1) and jhi+1 (jj = 2) of every
    // top-row block of this rank, columns 1 .. ihi+1.  Offsets in the order centre, NE corner, N face, E face.
    const int ioffU[4] = {0, 1, 0, 1}, joffU[4] = {0, 1, 1, 0};
    const int ioffT[4] = {-1, 0, -1, 0}, joffT[4] = {0, 1, 1, 0};
    const int* ioff = tfold ? ioffT : ioffU;
    const int* joff = tfold ? joffT : joffU;
    for (int gid : local) {
      const Block& d = all[gid];
      if (d.jb != nby - 1) continue;
      for (int jj = 1; jj <= 2; ++jj)
        for (int i = 1; i <= d.ihi + 1; ++i) {
          const int ig1 = ((d.i0 + (i - d.ilo)) % nxg + nxg) % nxg + 1;  // i_glob(i), cyclic
          for (int l = 0; l < 4; ++l) {
            int iSrc = nxg - ig1 + 1 - ioff[l];
            const int jSrc = 4 - jj - joff[l];
            if (iSrc == 0) iSrc = nxg;
            if (iSrc > nxg) iSrc -= nxg;
            if (jSrc > R || jSrc < 1) continue;
            fold_out[l].dst.push_back(addr(d, i, d.jhi + jj - 1));
            fold_out[l].src.push_back((int32_t)((jSrc - 1) * nxg + iSrc - 1));
          }
        }
    }
  }
}

const char* Domain::build(const std::vector<int>& owner, const std::vector<int>& lid) {
  nx_block = bsx + 2; ny_block = bsy + 2;
  if (ew < 0 || ew > BND_CLOSED) return "east-west boundary must be open, cyclic or closed";
  if (ns < 0 || ns > BND_TRIPOLET) return "unknown north-south boundary";
  if (tripole()) {
    if (ew != BND_CYCLIC) return "a tripole north boundary needs a cyclic east-west boundary";
    if (nxg % 2) return "a tripole north boundary needs an even nx_global";
    if (nyg - (nby - 1) * bsy < fold_rows() && nby > 1) return "tripole: the top block row has fewer physical rows than the fold needs";
    if (nyg < fold_rows()) return "tripole: fewer rows than the fold needs";
  }
  all.clear(); local.clear(); hsrc.clear(); hdst.clear(); hfill.clear(); send.clear(); recv.clear();
  rsrc.clear(); rdst.clear(); overlap = 0;

  for (int jb = 0; jb < nby; ++jb)
    for (int ib = 0; ib < nbx; ++ib) {
      Block b;
      b.gid = jb * nbx + ib; b.ib = ib; b.jb = jb;
      b.i0 = ib * bsx; b.j0 = jb * bsy;
      b.ilo = 2; b.jlo = 2;
      b.ihi = 1 + std::min(bsx, nxg - b.i0);  // padded last block: ice_blocks.F90:171-178
      b.jhi = 1 + std::min(bsy, nyg - b.j0);
      b.owner = owner[b.gid];
      b.local_id = lid[b.gid];
      b.own_jlo = b.jlo; b.own_jhi = b.jhi;
      all.push_back(b);
    }
  {  // this rank's blocks in local-id order
    std::vector<std::pair<int, int>> mine;
    for (const Block& b : all)
      if (b.owner == rank) mine.push_back({b.local_id, b.gid});
    std::sort(mine.begin(), mine.end());
    for (size_t k = 0; k < mine.size(); ++k) {
      if (mine[k].first != (int)k) return "block map: local ids of this rank are not 0 .. n-1";
      local.push_back(mine[k].second);
    }
  }

  const long long np = (long long)nx_block * ny_block;
  if (np * (long long)std::max<size_t>(local.size(), 1) > 0x7fffffffLL) return "local array too large for int32 addressing";
  auto addr = [&](const Block& b, int i, int j) {  // 1-based (i,j)
    return (int32_t)((long long)b.local_id * np + (long long)(j - 1) * nx_block + (i - 1));
  };
  const int ns_wrap = tripole() ? BND_OPEN : ns;   // south edge of a tripole grid is open (:246)

  std::map<int, HaloMsg> smap, rmap;
  // Visit every ghost cell of every block in one global order; both ends of a message
  // therefore agree on element order.
  for (const Block& d : all) {
    if (d.owner < 0) continue;
    for (int j = d.jlo - 1; j <= d.jhi + 1; ++j)
      for (int i = d.ilo - 1; i <= d.ihi + 1; ++i) {
        const bool ghost = (i < d.ilo || i > d.ihi || j < d.jlo || j > d.jhi);
        if (!ghost) continue;
        int ig = wrap(d.i0 + (i - d.ilo), nxg, ew);
        int jg = wrap(d.j0 + (j - d.jlo), nyg, ns_wrap);
        if (ig < 0 || jg < 0) continue;  // beyond an open/closed edge (or the fold, below): never written here
        const Block& s = all[(jg / bsy) * nbx + (ig / bsx)];
        if (s.owner < 0) {               // eliminated land block: fill value
          if (d.owner == rank) hfill.push_back(addr(d, i, j));
          continue;
        }
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

  build_fold();
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
  if (ns_bnd < 0 || ns_bnd > BND_TRIPOLET) return "unknown north-south boundary";
  nxg = nx_global; nyg = ny_global; bsx = nxg; bsy = nyg / nblocks_y;
  if (overlap_rows > bsy) return "overlap larger than a slab";
  if (ns_bnd == BND_TRIPOLE || ns_bnd == BND_TRIPOLET) {
    // the fold works on the top rows of the top slab, every subcycle; no other slab's extension may reach them
    if (ew_bnd != BND_CYCLIC) return "a tripole north boundary needs a cyclic east-west boundary";
    if (nxg % 2) return "a tripole north boundary needs an even nx_global";
    if (bsy < 4) return "tripole: slabs of fewer than 4 rows";
    if (nblocks_y > 1 && overlap_rows > bsy - 4) return "tripole: the overlap must leave the top 4 rows of the top slab alone";
  }
  overlap = overlap_rows;
  nx_block = bsx + 2; ny_block = bsy + 2 * overlap + 2;
  nbx = 1; nby = nblocks_y; npx = 1; npy = nranks_; nranks = nranks_; rank = rank_;
  ew = ew_bnd; ns = ns_bnd;
  from_map = false; map_owner.clear(); map_lid.clear();
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
  build_fold();
  return "";
}

}  // namespace cice
