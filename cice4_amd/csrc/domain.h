// Host-side block decomposition and halo topology (no device code).
// Counterpart of the reference's create_blocks (source/ice_blocks.F90:133-330),
// the cartesian block->task map (source/ice_distribution.F90:78) and the address
// lists ice_HaloCreate precomputes (mpi/ice_boundary.F90:153-1021, type ice_halo
// :51-76): ghost width 1, E/W/N/S edges and the four corners, sources are always
// physical cells, ghost cells beyond an open or closed domain edge are never
// written.  Any block->task map can be given (create_map: rake / space-curve distributions, and
// LAND-BLOCK ELIMINATION: blocks owned by nobody; ghost cells that face one take the fill value,
// mpi/ice_boundary.F90:5108-5111).  North boundary 'tripole' (U-fold, ice_blocks.F90:457-467,
// serial/ice_boundary.F90:705-869): the top rows of the top block row pass through a global
// buffer (the reference's bufTripole) and come back mirrored, see the fold_* members.
#pragma once
#include <cstdint>
#include <vector>

namespace cice {

// BND_TRIPOLE: fold through U points ('tripole'), BND_TRIPOLET: through T points ('tripoleT', source/ice_blocks.F90:228-233)
enum Boundary { BND_OPEN = 0, BND_CYCLIC = 1, BND_CLOSED = 2, BND_TRIPOLE = 3, BND_TRIPOLET = 4 };
// field location / kind codes of ice_HaloUpdate (drivers/cice4/ice_constants.F90:185-205); they only matter on
// a tripole boundary
enum FieldLoc { LOC_CENTER = 1, LOC_NECORNER = 2, LOC_NFACE = 3, LOC_EFACE = 4 };
enum FieldKind { KIND_SCALAR = 1, KIND_VECTOR = 2, KIND_ANGLE = 3 };

struct Block {
  int gid;                 // global block id, 0-based, i fastest (ice_blocks.F90:163-172)
  int ib, jb;              // cartesian block position
  int ilo, ihi, jlo, jhi;  // 1-based physical range inside the (nx_block,ny_block) array
  int i0, j0;              // 0-based global index of local cell (ilo), (jlo)
  int owner;               // rank; -1: eliminated (all-land block, owned by nobody)
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
  bool from_map = false;               // created from an explicit block -> task map (create_map): kept for rebuilding a
  std::vector<int> map_owner, map_lid; // neighbour's view of the same decomposition
  int ew = BND_CYCLIC, ns = BND_OPEN;
  bool tripole() const { return ns == BND_TRIPOLE || ns == BND_TRIPOLET; }
  int fold_rows() const { return ns == BND_TRIPOLET ? 3 : 2; }   // rows of the global fold buffer (tripoleRows, serial/ice_boundary.F90:199-204)
  // Test aid: route copies between DIFFERENT blocks of this rank through the message path
  // (send to / receive from the own rank), so that pack / RCCL / unpack run on a single GPU.
  bool self_comm = false;
  std::vector<Block> all;
  std::vector<int> local;            // gids of this rank's blocks, ascending
  std::vector<int32_t> hsrc, hdst;   // on-rank ghost copies: a[hdst[n]] = a[hsrc[n]]
  std::vector<int32_t> hfill;        // ghost cells facing an eliminated block: a[hfill[n]] = fillValue
  // Tripole fold.  Buffer index = r * nx_global + ig (ig 0-based; r = 0: global row ny-1, r = 1: top row ny).
  //   1. buffer = fillValue; buf[fold_bidx[n]] = a[fold_lsrc[n]] (+ fold_send/fold_recv for top-row blocks of
  //      other ranks; recv addresses are buffer indices);
  //   2. per field location, the degenerate top row is made symmetric: for the pairs (fold_lo, fold_hi)
  //      x = 0.5 * (buf[lo] + sign * buf[hi]); buf[lo] = x; buf[hi] = sign * x      (NE corner and N face only);
  //   3. a[fold_out[loc].dst[n]] = sign * buf[fold_out[loc].src[n]]: the ghost row and, for NE corner / N face,
  //      the top physical row itself, E/W ghost columns included.       sign = -1 for vectors and angles.
  bool fold = false;                 // this rank owns blocks on a tripole north boundary
  std::vector<int32_t> fold_lsrc, fold_bidx;
  std::vector<HaloMsg> fold_send, fold_recv;
  struct FoldOut { std::vector<int32_t> dst, src; };
  FoldOut fold_out[4];               // index = FieldLoc - 1
  std::vector<int32_t> fold_lo[4], fold_hi[4];
  std::vector<HaloMsg> send, recv;   // per peer, ascending peer; element order agrees on both ends
  // Wide-halo ("overlap") mode, create_slabs(): every block is a j-slab extended by `overlap`
  // rows into its neighbours; the extension rows are recomputed redundantly and refreshed from
  // their owner only every `overlap` subcycles.  Then hsrc/hdst hold only the E-W wrap (needed
  // every subcycle), rsrc/rdst the on-rank part of the refresh, send/recv its off-rank part.
  int overlap = 0;
  std::vector<int32_t> rsrc, rdst;

  int nblocks() const { return (int)local.size(); }
  const char* build(const std::vector<int>& owner, const std::vector<int>& lid);  // lists for the block map
  void build_fold();                                                             // tripole fold lists of the blocks in `all`
  // Returns empty string on success, else an error message.
  const char* create(int nx_global, int ny_global, int block_size_x, int block_size_y, int ew_bnd,
                     int ns_bnd, int rank_, int npx_, int npy_);
  // The same with an explicit block->task map: owner[gid] = rank or -1 (eliminated), local_id[gid] = position
  // among the owner's blocks (NULL: ascending gid).  gid = jb * nbx + ib.
  const char* create_map(int nx_global, int ny_global, int block_size_x, int block_size_y, int ew_bnd,
                         int ns_bnd, int rank_, int nranks_, const int* owner, const int* local_id);
  // nblocks_y j-slabs of full width dealt to nranks ranks (contiguous runs), each extended by
  // overlap_rows rows on both sides (clipped at the domain edge).  overlap_rows = 0 gives the
  // same blocks as create(nx, ny, nx, ny/nblocks_y, ...).
  const char* create_slabs(int nx_global, int ny_global, int nblocks_y, int ew_bnd, int ns_bnd,
                           int rank_, int nranks_, int overlap_rows);
};

}  // namespace cice
