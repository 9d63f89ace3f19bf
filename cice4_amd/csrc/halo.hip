#include "halo.h"

#include <rccl/rccl.h>

#include <cstring>

namespace cice {

#define CICE_NCCL(expr)                                                                         \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      throw ::cice::Error{CICE_ECOMM, std::string(#expr) + ": " + ncclGetErrorString(r_)};      \
  } while (0)

namespace {

// a[dst[n]] = a[src[n]] for every field; sources are physical cells and destinations
// ghost cells, so the copies are order-independent (serial/ice_boundary.F90:682-702).
template <class T>
__global__ __launch_bounds__(256) void k_halo_copy(T* __restrict__ base, int nfields, size_t stride,
                                                   const int32_t* __restrict__ src,
                                                   const int32_t* __restrict__ dst, int ncopy) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ncopy * nfields) return;
  int k = t / ncopy, e = t - k * ncopy;
  T* a = base + (size_t)k * stride;
  a[dst[e]] = a[src[e]];
}

// message m occupies buf[nfields*off[m] .. nfields*(off[m]+cnt[m])), field-major inside
template <class T>
__global__ __launch_bounds__(256) void k_pack(const T* __restrict__ base, int nfields, size_t stride,
                                              const int32_t* __restrict__ addr, int total,
                                              T* __restrict__ buf, const int* __restrict__ meta,
                                              int nmsg) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int m = 0;
  while (m + 1 < nmsg && e >= meta[2 * (m + 1)]) ++m;  // meta: off,cnt per message
  int off = meta[2 * m], cnt = meta[2 * m + 1];
  for (int k = 0; k < nfields; ++k)
    buf[(size_t)nfields * off + (size_t)k * cnt + (e - off)] = base[(size_t)k * stride + addr[e]];
}

template <class T>
__global__ __launch_bounds__(256) void k_unpack(T* __restrict__ base, int nfields, size_t stride,
                                                const int32_t* __restrict__ addr, int total,
                                                const T* __restrict__ buf,
                                                const int* __restrict__ meta, int nmsg) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int m = 0;
  while (m + 1 < nmsg && e >= meta[2 * (m + 1)]) ++m;
  int off = meta[2 * m], cnt = meta[2 * m + 1];
  for (int k = 0; k < nfields; ++k)
    base[(size_t)k * stride + addr[e]] = buf[(size_t)nfields * off + (size_t)k * cnt + (e - off)];
}

}  // namespace

// meta arrays live right behind the address lists (one allocation each)
void Halo::init(const Domain& d, hipStream_t s) {
  stream_ = s;
  rank_ = d.rank;
  nranks_ = d.nranks;
  ncopy_ = (int)d.hsrc.size();
  src_.alloc(ncopy_);
  dst_.alloc(ncopy_);
  if (ncopy_) {
    src_.upload(d.hsrc.data(), s);
    dst_.upload(d.hdst.data(), s);
  }
  nrefresh_ = (int)d.rsrc.size();
  rsrc_.alloc(nrefresh_);
  rdst_.alloc(nrefresh_);
  if (nrefresh_) {
    rsrc_.upload(d.rsrc.data(), s);
    rdst_.upload(d.rdst.data(), s);
  }
  {  // forwarding form: source cell -> the (at most 3: edge, edge, corner) ghosts mirroring it
    const size_t n = (size_t)d.nblocks() * d.nx_block * d.ny_block;
    std::vector<int32_t> slot(n, -1), fwd;
    for (int e = 0; e < ncopy_; ++e) {
      int32_t& sl = slot[d.hsrc[e]];
      if (sl < 0) {
        sl = (int32_t)(fwd.size() / 3);
        fwd.insert(fwd.end(), {-1, -1, -1});
      }
      int k = 0;
      while (k < 3 && fwd[3 * sl + k] >= 0) ++k;
      if (k == 3) {  // blocks only 1 cell wide: keep the separate copy kernel
        fwd_ok_ = false;
        break;
      }
      fwd[3 * sl + k] = d.hdst[e];
    }
    if (fwd.empty()) fwd.assign(3, -1);
    ring_slot_.alloc(n);
    ring_slot_.upload(slot.data(), s);
    fwd_.alloc(fwd.size());
    fwd_.upload(fwd.data(), s);
    CICE_HIP(hipStreamSynchronize(s));
  }
  auto flatten = [&](const std::vector<HaloMsg>& msgs, std::vector<int>& peer,
                     std::vector<int>& off, std::vector<int>& cnt, DevBuf<int32_t>& dev, int& nmsg) {
    std::vector<int32_t> flat;
    peer.clear(); off.clear(); cnt.clear();
    for (const HaloMsg& m : msgs) {
      peer.push_back(m.peer);
      off.push_back((int)flat.size());
      cnt.push_back((int)m.addr.size());
      flat.insert(flat.end(), m.addr.begin(), m.addr.end());
    }
    nmsg = (int)msgs.size();
    int total = (int)flat.size();
    // append meta (off,cnt pairs)
    for (int m = 0; m < nmsg; ++m) {
      flat.push_back(off[m]);
      flat.push_back(cnt[m]);
    }
    dev.alloc(flat.size());
    if (!flat.empty()) dev.upload(flat.data(), s);
    return total;
  };
  int ts = flatten(d.send, send_peer_, send_off_, send_cnt_, send_addr_, nsend_);
  int tr = flatten(d.recv, recv_peer_, recv_off_, recv_cnt_, recv_addr_, nrecv_);
  remote_ = nsend_ > 0 || nrecv_ > 0;
  total_s_ = ts;
  total_r_ = tr;
  reserve(MINF);
  CICE_HIP(hipStreamSynchronize(s));
}

void Halo::reserve(int nfields) {
  if (nfields <= cap_fields_) return;
  if (cap_fields_) CICE_HIP(hipStreamSynchronize(stream_));   // nothing in flight may still use the old buffers
  sendbuf_.alloc((size_t)total_s_ * nfields);
  recvbuf_.alloc((size_t)total_r_ * nfields);
  cap_fields_ = nfields;
  ++generation_;
}

Halo::~Halo() = default;

void Halo::set_comm(ncclComm* c, int rank, int nranks) {
  CICE_REQUIRE(rank == rank_ && nranks == nranks_, "cice_comm_init: rank/nranks differ from cice_domain_create");
  comm_ = c;
}

template <class T>
void Halo::update(T* base, int nfields, size_t stride, bool wrap) {
  CICE_REQUIRE(nfields >= 1, "halo: no field");
  if (remote_) reserve(nfields);    // any number of levels in one message per neighbour
  const int total_s = nsend_ ? send_off_.back() + send_cnt_.back() : 0;
  const int total_r = nrecv_ ? recv_off_.back() + recv_cnt_.back() : 0;
  // The wrap list goes first: a wide-halo refresh copies whole rows INCLUDING their E/W ghost
  // columns, so the owner's ghost columns must be current before they are packed or copied.
  if (ncopy_ && wrap) {
    int t = ncopy_ * nfields;
    hipLaunchKernelGGL(k_halo_copy<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, src_.p, dst_.p, ncopy_);
  }
  if (remote_) {
    CICE_REQUIRE(comm_ != nullptr, "halo: cice_comm_init has not been called on a multi-rank domain");
    T* sb = reinterpret_cast<T*>(sendbuf_.p);
    T* rb = reinterpret_cast<T*>(recvbuf_.p);
    if (total_s) {
      const int* meta = reinterpret_cast<const int*>(send_addr_.p + total_s);
      hipLaunchKernelGGL(k_pack<T>, dim3((total_s + 255) / 256), dim3(256), 0, stream_, base, nfields,
                         stride, send_addr_.p, total_s, sb, meta, nsend_);
    }
    const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclInt32;
    CICE_NCCL(ncclGroupStart());
    for (int m = 0; m < nrecv_; ++m)
      CICE_NCCL(ncclRecv(rb + (size_t)nfields * recv_off_[m], (size_t)nfields * recv_cnt_[m], dt,
                         recv_peer_[m], (ncclComm_t)comm_, stream_));
    for (int m = 0; m < nsend_; ++m)
      CICE_NCCL(ncclSend(sb + (size_t)nfields * send_off_[m], (size_t)nfields * send_cnt_[m], dt,
                         send_peer_[m], (ncclComm_t)comm_, stream_));
    CICE_NCCL(ncclGroupEnd());
  }
  if (nrefresh_) {  // sources are owned rows, destinations overlap/ghost rows: disjoint from the wrap list's sources
    int t = nrefresh_ * nfields;
    hipLaunchKernelGGL(k_halo_copy<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, rsrc_.p, rdst_.p, nrefresh_);
  }
  if (remote_ && total_r) {
    const int* meta = reinterpret_cast<const int*>(recv_addr_.p + total_r);
    hipLaunchKernelGGL(k_unpack<T>, dim3((total_r + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, recv_addr_.p, total_r, reinterpret_cast<const T*>(recvbuf_.p), meta,
                       nrecv_);
  }
  CICE_HIP(hipGetLastError());
}

void Halo::update_r8(double* base, int nfields, size_t stride, bool wrap) { update<double>(base, nfields, stride, wrap); }
void Halo::update_i4(int32_t* base, int nfields, size_t stride) { update<int32_t>(base, nfields, stride, true); }

}  // namespace cice
