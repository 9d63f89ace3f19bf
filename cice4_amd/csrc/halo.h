// Ghost-cell update on the device: on-rank copies by address list, off-rank
// rows/columns packed and exchanged with RCCL point-to-point over xGMI.
// Replaces ice_HaloUpdate2DR8 / 2DI4 (mpi/ice_boundary.F90:1028-1417, 1820-...)
// for non-tripole grids.  Several fields (e.g. uvel and vvel) travel in ONE
// message per neighbour: the exchange is latency-bound (2.6-29 KB per field).
#pragma once
#include <vector>

#include "common.h"
#include "domain.h"

struct ncclComm;

namespace cice {

// In-process stand-in for the RCCL communicator: the ranks of one "link" are contexts of ONE process (one host
// thread each), on one GPU or several; a message travels device -> host mailbox -> device.  What the reference's
// serial/ directory is to its mpi/ one: the whole multi-rank path -- message lists, pack / unpack kernels, wide-halo
// refresh, the cross-rank resident loop's agreement -- runs and can be checked on a box with a single GPU, where
// RCCL refuses two ranks on one device.  Slow by construction (every message synchronises the stream); tests only.
struct LocalLink;
LocalLink* local_link_get(int link_id, int nranks);   // the link of that id (created on first use; sizes must agree)
// the same between PROCESSES of one host through a file under /dev/shm (rank 0 creates it); owned by the caller
LocalLink* shm_link_open(const char* name, int rank, int nranks, size_t box_bytes);
void link_close(LocalLink* l);
// timing aid: no partner at all -- what a rank would send comes back as what it would receive (halo.hip: MirrorLink); owned by the caller
LocalLink* mirror_link_new(int nranks);

enum HaloParts { HALO_COPIES = 1, HALO_FOLD = 2, HALO_ALL = 3 };

class Halo {
 public:
  Halo() = default;
  ~Halo();
  void init(const Domain& d, hipStream_t s);
  // The RCCL communicator belongs to the CONTEXT (it outlives a change of the block decomposition); a Halo
  // borrows it.
  void set_comm(ncclComm* c, int rank, int nranks);
  void set_link(LocalLink* l, int rank, int nranks);
  bool multi_rank() const { return remote_; }  // any message to exchange (normally: nranks > 1)
  bool has_refresh() const { return remote_ || nrefresh_ > 0 || nfill_ > 0; }
  bool has_onrank_refresh() const { return nrefresh_ > 0; }   // block-to-block copies that are NOT in the every-subcycle list (create_slabs)
  // nfields fields of element type T, field k starting at base + k*stride (elements).
  // Always performs the refresh part (on-rank block-to-block rows of a wide-halo domain and all
  // off-rank messages); wrap = false skips the every-subcycle on-rank list (hsrc/hdst) because
  // the caller -- the subcycle kernel -- has written those ghosts itself.
  // loc / kind (FieldLoc, FieldKind) and fill only matter on a tripole north boundary / next to eliminated
  // land blocks (domain.h); the defaults are right for every other domain.
  // parts: HALO_COPIES = everything but the tripole fold, HALO_FOLD = the fold alone (a wide-halo slab domain folds u, v
  // after every subcycle but refreshes its overlap rows -- all 14 planes of the state, no fold -- only now and then).
  void update_r8(double* base, int nfields, size_t stride, bool wrap = true, int loc = LOC_CENTER,
                 int kind = KIND_SCALAR, double fill = 0.0, int parts = HALO_ALL);
  void update_i4(int32_t* base, int nfields, size_t stride, int loc = LOC_CENTER, int kind = KIND_SCALAR,
                 int32_t fill = 0);
  void update_r4(float* base, int nfields, size_t stride, int loc = LOC_CENTER, int kind = KIND_SCALAR,
                 float fill = 0.0f);
  bool has_fold() const { return fold_; }
  // in-place maximum of one 32-bit word over the ranks (control decisions every rank has to take alike); no-op
  // without a communicator
  void all_max_u32(unsigned* dev_word);
  // Device pointers to the on-rank copy list, for kernels that fold it in.
  const int32_t* d_src() const { return src_.p; }
  const int32_t* d_dst() const { return dst_.p; }
  int ncopy() const { return ncopy_; }
  // Forwarding form of the same list, for producers that write ghosts themselves:
  // ring_slot[cell] = slot or -1; fwd[3*slot + k] = k-th ghost address mirroring that cell or -1.
  const int32_t* d_ring_slot() const { return ring_slot_.p; }
  const int32_t* d_fwd() const { return fwd_.p; }
  bool fwd_ok() const { return fwd_ok_; }
  // bumped whenever the message buffers are re-allocated (a 65-level bound_state update after the EVP
  // loop was captured): whoever holds a hipGraph of update() calls has to re-capture it
  int generation() const { return generation_; }

 private:
  template <class T>
  void update(T* base, int nfields, size_t stride, bool wrap, int loc, int kind, T fill, int parts);
  template <class T>
  void update_copies(T* base, int nfields, size_t stride, bool wrap, T fill);
  template <class T>
  void update_fold(T* base, int nfields, size_t stride, int loc, int kind, T fill);
  template <class T>
  void exchange(const T* src_base, size_t src_stride, T* dst_base, size_t dst_stride, int nfields,
                const DevBuf<int32_t>& saddr, const std::vector<int>& speer, const std::vector<int>& soff,
                const std::vector<int>& scnt, int ns, const DevBuf<int32_t>& raddr,
                const std::vector<int>& rpeer, const std::vector<int>& roff, const std::vector<int>& rcnt, int nr);
  hipStream_t stream_ = nullptr;
  int ncopy_ = 0, nrefresh_ = 0, rank_ = 0, nranks_ = 1;
  bool fwd_ok_ = true, remote_ = false;
  DevBuf<int32_t> src_, dst_, rsrc_, rdst_, send_addr_, recv_addr_, ring_slot_, fwd_;
  std::vector<int> send_peer_, send_off_, send_cnt_, recv_peer_, recv_off_, recv_cnt_;
  int nsend_ = 0, nrecv_ = 0;
  // land-block fill list and tripole fold (domain.h)
  int fold_rows_ = 2;   // rows of the global fold buffer (2: fold through U points, 3: through T points)
  int nfill_ = 0, nxg_ = 0, nfold_src_ = 0, nfold_out_[4] = {0, 0, 0, 0}, nfold_pair_[4] = {0, 0, 0, 0};
  bool fold_ = false;
  DevBuf<int32_t> fill_, fold_lsrc_, fold_bidx_, fold_send_addr_, fold_recv_addr_, fold_dst_[4], fold_src_[4],
      fold_lo_[4], fold_hi_[4];
  std::vector<int> fsend_peer_, fsend_off_, fsend_cnt_, frecv_peer_, frecv_off_, frecv_cnt_;
  int nfsend_ = 0, nfrecv_ = 0, ftotal_s_ = 0, ftotal_r_ = 0;
  DevBuf<double> foldbuf_;            // 2 * nx_global elements per field, grown like the message buffers
  int fold_cap_ = 0;
  DevBuf<double> sendbuf_, recvbuf_;  // sized for cap_fields_ fields of 8-byte elements
  int cap_fields_ = 0, total_s_ = 0, total_r_ = 0, generation_ = 0;
  void reserve(int nfields);          // grows the message buffers (never shrinks)
  ncclComm* comm_ = nullptr;
  LocalLink* link_ = nullptr;
  template <class T>
  void link_exchange(const T* sb, T* rb, int nfields, const std::vector<int>& speer, const std::vector<int>& soff,
                     const std::vector<int>& scnt, int ns, const std::vector<int>& rpeer, const std::vector<int>& roff,
                     const std::vector<int>& rcnt, int nr);
  static constexpr int MINF = 14;     // u, v and the 12 stresses in one message: allocated up front
};

}  // namespace cice
