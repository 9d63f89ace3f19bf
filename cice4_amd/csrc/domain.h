// Host-side block decomposition and halo topology (no device code).
// Counterpart of the reference's create_blocks (source/ice_blocks.F90:133-330),
// the cartesian block->task map (source/ice_distribution.F90:78) and the address
// lists ice_HaloCreate precomputes (mpi/ice_boundary.F90:153-1021, type ice_halo
// :51-76): ghost width 1, E/W/N/S edges and the four corners, sources are always
// physical cells, ghost cells beyond an open or closed domain edge are never
// written.  No tripole fold, no land-block elimination.
#pragma once
#include <cstdint>
#include <vector>

namespace cice {

enum Boundary { BND_OPEN = 0, BND_CYCLIC = 1, BND_CLOSED = 2 };

struct Block {
  int gid;                 // global block id, 0-based, i fastest (ice_blocks.F90:163-172)
  int ib, jb;              // cartesian block position
  int ilo, ihi, jlo, jhi;  // 1-based physical range inside the (nx_block,ny_block) array
  int i0, j0;              // 0-based global index of local cell (ilo), (jlo)
  int owner;               // rank
  int local_id;            // position among the owner's blocks
  int own_jlo, own_jhi;    // rows this block owns (== jlo..jhi unless the domain has overlap rows)
};

struct HaloMsg {
  int peer;
  std::vector<int32_t> addr;  // linear addresses into the local (nblocks,ny_block,nx_block) array
};

struct Domain {
  int nxg = 0, nyg = 0, bsx = 0, bsy = 0;
  int nx_block = 0, ny_block = 0;
  int nbx = 0, nby = 0, npx = 1, npy = 1, rank = 0, nranks = 1;
  int ew = BND_CYCLIC, ns = BND_OPEN;
  // Test aid: route copies between DIFFERENT blocks of this rank through the message path
  // (send to / receive from the own rank), so that pack / RCCL / unpack run on a single GPU.
  bool self_comm = false;
  std::vector<Block> all;
  std::vector<int> local;            // gids of this rank's blocks, ascending
  std::vector<int32_t> hsrc, hdst;   // on-rank ghost copies: a[hdst[n]] = a[hsrc[n]]
  std::vector<HaloMsg> send, recv;   // per peer, ascending peer; element order agrees on both ends
  // Wide-halo ("overlap") mode, create_slabs(): every block is a j-slab extended by `overlap`
  // rows into its neighbours; the extension rows are recomputed redundantly and refreshed from
  // their owner only every `overlap` subcycles.  Then hsrc/hdst hold only the E-W wrap (needed
  // every subcycle), rsrc/rdst the on-rank part of the refresh, send/recv its off-rank part.
  int overlap = 0;
  std::vector<int32_t> rsrc, rdst;

  int nblocks() const { return (int)local.size(); }
  // Returns empty string on success, else an error message.
  const char* create(int nx_global, int ny_global, int block_size_x, int block_size_y, int ew_bnd,
                     int ns_bnd, int rank_, int npx_, int npy_);
  // nblocks_y j-slabs of full width dealt to nranks ranks (contiguous runs), each extended by
  // overlap_rows rows on both sides (clipped at the domain edge).  overlap_rows = 0 gives the
  // same blocks as create(nx, ny, nx, ny/nblocks_y, ...).
  const char* create_slabs(int nx_global, int ny_global, int nblocks_y, int ew_bnd, int ns_bnd,
                           int rank_, int nranks_, int overlap_rows);
};

}  // namespace cice
