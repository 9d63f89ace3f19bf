#include "halo.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>

namespace cice {

#define CICE_NCCL(expr)                                                                         \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      throw ::cice::Error{CICE_ECOMM, std::string(#expr) + ": " + ncclGetErrorString(r_)};      \
  } while (0)

// ---- in-process link (halo.h) ------------------------------------------------------------------------------------
struct LocalLink {
  int nranks = 0;
  std::mutex m;
  std::condition_variable cv;
  // box[src * nranks + dst]: the message src posted for dst, its sequence number, and the last one dst has taken
  std::vector<std::vector<char>> box;
  std::vector<long> posted, taken;
  // all-reduce of one word
  std::vector<unsigned> red_val;
  std::vector<long> red_seq;
};

LocalLink* local_link_get(int link_id, int nranks) {
  static std::mutex gm;
  static std::map<int, std::unique_ptr<LocalLink>> links;
  std::lock_guard<std::mutex> g(gm);
  auto& l = links[link_id];
  if (!l) {
    l.reset(new LocalLink());
    l->nranks = nranks;
    l->box.resize((size_t)nranks * nranks);
    l->posted.assign((size_t)nranks * nranks, 0);
    l->taken.assign((size_t)nranks * nranks, 0);
    l->red_val.assign(nranks, 0);
    l->red_seq.assign(nranks, 0);
  }
  if (l->nranks != nranks) throw Error{CICE_EINVAL, "cice_comm_init_local: this link exists with another number of ranks"};
  return l.get();
}

namespace {
constexpr int LINK_WAIT_S = 60;   // a rank whose partner never shows up fails instead of hanging
}

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

// ghost cells that face an eliminated land block (mpi/ice_boundary.F90:5108-5111)
template <class T>
__global__ __launch_bounds__(256) void k_halo_fill(T* __restrict__ base, int nfields, size_t stride,
                                                   const int32_t* __restrict__ addr, int n, T fill) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nfields) return;
  int k = t / n, e = t - k * n;
  base[(size_t)k * stride + addr[e]] = fill;
}

// ---- tripole fold (serial/ice_boundary.F90:705-869); buffer: nfields x (2 x nx_global) ----
template <class T>
__global__ __launch_bounds__(256) void k_fold_fill(T* __restrict__ buf, size_t n, T fill) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) buf[t] = fill;
}

template <class T>
__global__ __launch_bounds__(256) void k_fold_gather(const T* __restrict__ base, int nfields, size_t stride,
                                                     const int32_t* __restrict__ lsrc,
                                                     const int32_t* __restrict__ bidx, int n,
                                                     T* __restrict__ buf, int bstride) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nfields) return;
  int k = t / n, e = t - k * n;
  buf[(size_t)k * bstride + bidx[e]] = base[(size_t)k * stride + lsrc[e]];
}

__device__ __forceinline__ double fold_avg(double x1, double x2, int sgn) { return 0.5 * (x1 + sgn * x2); }
__device__ __forceinline__ float fold_avg(float x1, float x2, int sgn) { return 0.5f * (x1 + sgn * x2); }
__device__ __forceinline__ int32_t fold_avg(int32_t x1, int32_t x2, int sgn) {
  return (int32_t)round(0.5 * (double)(x1 + sgn * x2));   // nint(): halves away from zero
}

// "top row is degenerate, so must enforce symmetry": pairs are disjoint
template <class T>
__global__ __launch_bounds__(256) void k_fold_sym(T* __restrict__ buf, int nfields, int bstride,
                                                  const int32_t* __restrict__ lo,
                                                  const int32_t* __restrict__ hi, int n, int sgn) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nfields) return;
  int k = t / n, e = t - k * n;
  T* b = buf + (size_t)k * bstride;
  const T xavg = fold_avg(b[lo[e]], b[hi[e]], sgn);
  b[lo[e]] = xavg;
  b[hi[e]] = (T)sgn * xavg;
}

template <class T>
__global__ __launch_bounds__(256) void k_fold_out(T* __restrict__ base, int nfields, size_t stride,
                                                  const int32_t* __restrict__ dst,
                                                  const int32_t* __restrict__ src, int n,
                                                  const T* __restrict__ buf, int bstride, int sgn) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nfields) return;
  int k = t / n, e = t - k * n;
  base[(size_t)k * stride + dst[e]] = (T)sgn * buf[(size_t)k * bstride + src[e]];
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
  ftotal_s_ = flatten(d.fold_send, fsend_peer_, fsend_off_, fsend_cnt_, fold_send_addr_, nfsend_);
  ftotal_r_ = flatten(d.fold_recv, frecv_peer_, frecv_off_, frecv_cnt_, fold_recv_addr_, nfrecv_);
  remote_ = nsend_ > 0 || nrecv_ > 0 || nfsend_ > 0 || nfrecv_ > 0;
  total_s_ = std::max(ts, ftotal_s_);
  total_r_ = std::max(tr, ftotal_r_);
  auto put = [&](DevBuf<int32_t>& dev, const std::vector<int32_t>& h) {
    dev.alloc(h.size());
    if (!h.empty()) dev.upload(h.data(), s);
    return (int)h.size();
  };
  nfill_ = put(fill_, d.hfill);
  fold_ = d.fold;
  nxg_ = d.nxg;
  nfold_src_ = put(fold_lsrc_, d.fold_lsrc);
  put(fold_bidx_, d.fold_bidx);
  for (int l = 0; l < 4; ++l) {
    nfold_out_[l] = put(fold_dst_[l], d.fold_out[l].dst);
    put(fold_src_[l], d.fold_out[l].src);
    nfold_pair_[l] = put(fold_lo_[l], d.fold_lo[l]);
    put(fold_hi_[l], d.fold_hi[l]);
  }
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

void Halo::set_link(LocalLink* l, int rank, int nranks) {
  CICE_REQUIRE(rank == rank_ && nranks == nranks_, "cice_comm_init_local: rank/nranks differ from cice_domain_create");
  link_ = l;
}

// One exchange through the in-process link: the packed send buffer goes to the host and into the partners' mailboxes,
// theirs come back the same way.  Every rank of a link makes the same sequence of calls (the halo updates of the
// model are collective), so a per-Halo call counter pairs the messages up.
template <class T>
void Halo::link_exchange(const T* sb, T* rb, int nfields, const std::vector<int>& speer, const std::vector<int>& soff,
                         const std::vector<int>& scnt, int ns, const std::vector<int>& rpeer,
                         const std::vector<int>& roff, const std::vector<int>& rcnt, int nr) {
  LocalLink& L = *link_;
  const int R = L.nranks;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LINK_WAIT_S);
  CICE_HIP(hipStreamSynchronize(stream_));   // the pack kernel has run
  for (int m = 0; m < ns; ++m) {
    const size_t bytes = (size_t)nfields * scnt[m] * sizeof(T);
    std::vector<char> tmp(bytes);
    CICE_HIP(hipMemcpy(tmp.data(), sb + (size_t)nfields * soff[m], bytes, hipMemcpyDeviceToHost));
    std::unique_lock<std::mutex> lk(L.m);
    const size_t slot = (size_t)rank_ * R + speer[m];
    // the partner has taken the previous message of this pair
    if (!L.cv.wait_until(lk, deadline, [&] { return L.taken[slot] == L.posted[slot]; }))
      throw Error{CICE_ECOMM, "in-process link: the partner rank did not take the previous message (is it running?)"};
    L.box[slot].swap(tmp);
    L.posted[slot] += 1;       // messages of a (sender, receiver) pair are numbered by the pair
    L.cv.notify_all();
  }
  for (int m = 0; m < nr; ++m) {
    const size_t bytes = (size_t)nfields * rcnt[m] * sizeof(T);
    std::vector<char> tmp;
    {
      std::unique_lock<std::mutex> lk(L.m);
      const size_t slot = (size_t)rpeer[m] * R + rank_;
      if (!L.cv.wait_until(lk, deadline, [&] { return L.posted[slot] == L.taken[slot] + 1; }))
        throw Error{CICE_ECOMM, "in-process link: no message from the partner rank (is it running?)"};
      if (L.box[slot].size() != bytes)
        throw Error{CICE_ECOMM, "in-process link: the ranks are not making the same sequence of halo updates"};
      tmp.swap(L.box[slot]);
      L.taken[slot] += 1;
      L.cv.notify_all();
    }
    CICE_HIP(hipMemcpy(rb + (size_t)nfields * roff[m], tmp.data(), bytes, hipMemcpyHostToDevice));
  }
}

// pack -> grouped RCCL send/recv -> unpack for one set of message lists
template <class T>
void Halo::exchange(const T* src_base, size_t src_stride, T* dst_base, size_t dst_stride, int nfields,
                    const DevBuf<int32_t>& saddr, const std::vector<int>& speer, const std::vector<int>& soff,
                    const std::vector<int>& scnt, int ns, const DevBuf<int32_t>& raddr,
                    const std::vector<int>& rpeer, const std::vector<int>& roff, const std::vector<int>& rcnt,
                    int nr) {
  const int total_s = ns ? soff.back() + scnt.back() : 0;
  const int total_r = nr ? roff.back() + rcnt.back() : 0;
  if (!total_s && !total_r) return;
  CICE_REQUIRE(comm_ != nullptr || link_ != nullptr, "halo: cice_comm_init has not been called on a multi-rank domain");
  T* sb = reinterpret_cast<T*>(sendbuf_.p);
  T* rb = reinterpret_cast<T*>(recvbuf_.p);
  if (total_s) {
    const int* meta = reinterpret_cast<const int*>(saddr.p + total_s);
    hipLaunchKernelGGL(k_pack<T>, dim3((total_s + 255) / 256), dim3(256), 0, stream_, src_base, nfields,
                       src_stride, saddr.p, total_s, sb, meta, ns);
  }
  if (link_) {
    link_exchange<T>(sb, rb, nfields, speer, soff, scnt, ns, rpeer, roff, rcnt, nr);
    if (total_r) {
      const int* meta = reinterpret_cast<const int*>(raddr.p + total_r);
      hipLaunchKernelGGL(k_unpack<T>, dim3((total_r + 255) / 256), dim3(256), 0, stream_, dst_base, nfields,
                         dst_stride, raddr.p, total_r, reinterpret_cast<const T*>(recvbuf_.p), meta, nr);
    }
    return;
  }
  const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : (std::is_same<T, float>::value ? ncclFloat : ncclInt32);
  CICE_NCCL(ncclGroupStart());
  for (int m = 0; m < nr; ++m)
    CICE_NCCL(ncclRecv(rb + (size_t)nfields * roff[m], (size_t)nfields * rcnt[m], dt, rpeer[m],
                       (ncclComm_t)comm_, stream_));
  for (int m = 0; m < ns; ++m)
    CICE_NCCL(ncclSend(sb + (size_t)nfields * soff[m], (size_t)nfields * scnt[m], dt, speer[m],
                       (ncclComm_t)comm_, stream_));
  CICE_NCCL(ncclGroupEnd());
  if (total_r) {
    const int* meta = reinterpret_cast<const int*>(raddr.p + total_r);
    hipLaunchKernelGGL(k_unpack<T>, dim3((total_r + 255) / 256), dim3(256), 0, stream_, dst_base, nfields,
                       dst_stride, raddr.p, total_r, reinterpret_cast<const T*>(recvbuf_.p), meta, nr);
  }
}

template <class T>
void Halo::update(T* base, int nfields, size_t stride, bool wrap, int loc, int kind, T fill) {
  CICE_REQUIRE(nfields >= 1, "halo: no field");
  if (remote_) reserve(nfields);    // any number of levels in one message per neighbour
  // The wrap list goes first: a wide-halo refresh copies whole rows INCLUDING their E/W ghost
  // columns, so the owner's ghost columns must be current before they are packed or copied.
  if (ncopy_ && wrap) {
    int t = ncopy_ * nfields;
    hipLaunchKernelGGL(k_halo_copy<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, src_.p, dst_.p, ncopy_);
  }
  if (nfill_) {
    int t = nfill_ * nfields;
    hipLaunchKernelGGL(k_halo_fill<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields, stride,
                       fill_.p, nfill_, fill);
  }
  if (nsend_ || nrecv_) {
    // sources are physical cells, destinations ghost cells: the on-rank refresh below may run between
    // pack and unpack
    const int total_s = nsend_ ? send_off_.back() + send_cnt_.back() : 0;
    const int total_r = nrecv_ ? recv_off_.back() + recv_cnt_.back() : 0;
    CICE_REQUIRE(comm_ != nullptr || link_ != nullptr, "halo: cice_comm_init has not been called on a multi-rank domain");
    T* sb = reinterpret_cast<T*>(sendbuf_.p);
    T* rb = reinterpret_cast<T*>(recvbuf_.p);
    if (total_s) {
      const int* meta = reinterpret_cast<const int*>(send_addr_.p + total_s);
      hipLaunchKernelGGL(k_pack<T>, dim3((total_s + 255) / 256), dim3(256), 0, stream_, base, nfields,
                         stride, send_addr_.p, total_s, sb, meta, nsend_);
    }
    if (link_) {
      link_exchange<T>(sb, rb, nfields, send_peer_, send_off_, send_cnt_, nsend_, recv_peer_, recv_off_, recv_cnt_, nrecv_);
    } else {
      const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : (std::is_same<T, float>::value ? ncclFloat : ncclInt32);
      CICE_NCCL(ncclGroupStart());
      for (int m = 0; m < nrecv_; ++m)
        CICE_NCCL(ncclRecv(rb + (size_t)nfields * recv_off_[m], (size_t)nfields * recv_cnt_[m], dt,
                           recv_peer_[m], (ncclComm_t)comm_, stream_));
      for (int m = 0; m < nsend_; ++m)
        CICE_NCCL(ncclSend(sb + (size_t)nfields * send_off_[m], (size_t)nfields * send_cnt_[m], dt,
                           send_peer_[m], (ncclComm_t)comm_, stream_));
      CICE_NCCL(ncclGroupEnd());
    }
  }
  if (nrefresh_) {  // sources are owned rows, destinations overlap/ghost rows: disjoint from the wrap list's sources
    int t = nrefresh_ * nfields;
    hipLaunchKernelGGL(k_halo_copy<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, rsrc_.p, rdst_.p, nrefresh_);
  }
  if (nrecv_) {
    const int total_r = recv_off_.back() + recv_cnt_.back();
    const int* meta = reinterpret_cast<const int*>(recv_addr_.p + total_r);
    hipLaunchKernelGGL(k_unpack<T>, dim3((total_r + 255) / 256), dim3(256), 0, stream_, base, nfields,
                       stride, recv_addr_.p, total_r, reinterpret_cast<const T*>(recvbuf_.p), meta,
                       nrecv_);
  }
  if (fold_ || nfsend_) {   // tripole north boundary, after all regular copies (serial/ice_boundary.F90:705)
    CICE_REQUIRE(loc >= LOC_CENTER && loc <= LOC_EFACE, "halo: field location unknown on a tripole grid");
    CICE_REQUIRE(kind >= KIND_SCALAR && kind <= KIND_ANGLE, "halo: field kind unknown on a tripole grid");
    const int sgn = kind == KIND_SCALAR ? 1 : -1;
    const int bstride = 2 * nxg_;
    T* buf = nullptr;
    if (fold_) {
      if (fold_cap_ < nfields) {
        if (fold_cap_) CICE_HIP(hipStreamSynchronize(stream_));
        foldbuf_.alloc((size_t)bstride * nfields);
        fold_cap_ = nfields;
        ++generation_;
      }
      buf = reinterpret_cast<T*>(foldbuf_.p);
      const size_t nbuf = (size_t)bstride * nfields;
      hipLaunchKernelGGL(k_fold_fill<T>, dim3((unsigned)((nbuf + 255) / 256)), dim3(256), 0, stream_, buf, nbuf, fill);
      if (nfold_src_) {
        int t = nfold_src_ * nfields;
        hipLaunchKernelGGL(k_fold_gather<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, (const T*)base, nfields,
                           stride, fold_lsrc_.p, fold_bidx_.p, nfold_src_, buf, bstride);
      }
    }
    // top rows of the other ranks' top-row blocks: straight into the buffer
    exchange<T>(base, stride, buf, (size_t)bstride, nfields, fold_send_addr_, fsend_peer_, fsend_off_, fsend_cnt_,
                nfsend_, fold_recv_addr_, frecv_peer_, frecv_off_, frecv_cnt_, nfrecv_);
    if (fold_) {
      const int l = loc - 1;
      if (nfold_pair_[l]) {
        int t = nfold_pair_[l] * nfields;
        hipLaunchKernelGGL(k_fold_sym<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, buf, nfields, bstride,
                           fold_lo_[l].p, fold_hi_[l].p, nfold_pair_[l], sgn);
      }
      if (nfold_out_[l]) {
        int t = nfold_out_[l] * nfields;
        hipLaunchKernelGGL(k_fold_out<T>, dim3((t + 255) / 256), dim3(256), 0, stream_, base, nfields, stride,
                           fold_dst_[l].p, fold_src_[l].p, nfold_out_[l], (const T*)buf, bstride, sgn);
      }
    }
  }
  CICE_HIP(hipGetLastError());
}

void Halo::all_max_u32(unsigned* dev_word) {
  if (nranks_ <= 1) return;
  if (link_) {   // in-process link: every rank posts its word, takes the maximum once all have
    LocalLink& L = *link_;
    unsigned v = 0;
    CICE_HIP(hipStreamSynchronize(stream_));
    CICE_HIP(hipMemcpy(&v, dev_word, 4, hipMemcpyDeviceToHost));
    std::unique_lock<std::mutex> lk(L.m);
    const long seq = L.red_seq[rank_] + 1;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LINK_WAIT_S);
    // nobody may still be reading the previous round's values
    if (!L.cv.wait_until(lk, deadline, [&] {
          for (int r = 0; r < L.nranks; ++r)
            if (L.red_seq[r] < seq - 1) return false;
          return true;
        }))
      throw Error{CICE_ECOMM, "in-process link: all-reduce out of step"};
    L.red_val[rank_] = v;
    L.red_seq[rank_] = seq;
    L.cv.notify_all();
    if (!L.cv.wait_until(lk, deadline, [&] {
          for (int r = 0; r < L.nranks; ++r)
            if (L.red_seq[r] < seq) return false;
          return true;
        }))
      throw Error{CICE_ECOMM, "in-process link: a rank did not reach the all-reduce"};
    unsigned mx = 0;
    for (int r = 0; r < L.nranks; ++r) mx = std::max(mx, L.red_val[r]);
    lk.unlock();
    CICE_HIP(hipMemcpy(dev_word, &mx, 4, hipMemcpyHostToDevice));
    return;
  }
  if (!comm_) return;
  CICE_NCCL(ncclAllReduce(dev_word, dev_word, 1, ncclUint32, ncclMax, (ncclComm_t)comm_, stream_));
}

void Halo::update_r8(double* base, int nfields, size_t stride, bool wrap, int loc, int kind, double fill) {
  update<double>(base, nfields, stride, wrap, loc, kind, fill);
}
void Halo::update_i4(int32_t* base, int nfields, size_t stride, int loc, int kind, int32_t fill) {
  update<int32_t>(base, nfields, stride, true, loc, kind, fill);
}
void Halo::update_r4(float* base, int nfields, size_t stride, int loc, int kind, float fill) {
  update<float>(base, nfields, stride, true, loc, kind, fill);
}

}  // namespace cice
