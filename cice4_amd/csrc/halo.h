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

class Halo {
 public:
  Halo() = default;
  ~Halo();
  void init(const Domain& d, hipStream_t s);
  // The RCCL communicator belongs to the CONTEXT (it outlives a change of the block decomposition); a Halo
  // borrows it.
  void set_comm(ncclComm* c, int rank, int nranks);
  bool multi_rank() const { return remote_; }  // any message to exchange (normally: nranks > 1)
  bool has_refresh() const { return remote_ || nrefresh_ > 0; }
  // nfields fields of element type T, field k starting at base + k*stride (elements).
  // Always performs the refresh part (on-rank block-to-block rows of a wide-halo domain and all
  // off-rank messages); wrap = false skips the every-subcycle on-rank list (hsrc/hdst) because
  // the caller -- the subcycle kernel -- has written those ghosts itself.
  void update_r8(double* base, int nfields, size_t stride, bool wrap = true);
  void update_i4(int32_t* base, int nfields, size_t stride);
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
  void update(T* base, int nfields, size_t stride, bool wrap);
  hipStream_t stream_ = nullptr;
  int ncopy_ = 0, nrefresh_ = 0, rank_ = 0, nranks_ = 1;
  bool fwd_ok_ = true, remote_ = false;
  DevBuf<int32_t> src_, dst_, rsrc_, rdst_, send_addr_, recv_addr_, ring_slot_, fwd_;
  std::vector<int> send_peer_, send_off_, send_cnt_, recv_peer_, recv_off_, recv_cnt_;
  int nsend_ = 0, nrecv_ = 0;
  DevBuf<double> sendbuf_, recvbuf_;  // sized for cap_fields_ fields of 8-byte elements
  int cap_fields_ = 0, total_s_ = 0, total_r_ = 0, generation_ = 0;
  void reserve(int nfields);          // grows the message buffers (never shrinks)
  ncclComm* comm_ = nullptr;
  static constexpr int MINF = 14;     // u, v and the 12 stresses in one message: allocated up front
};

}  // namespace cice
