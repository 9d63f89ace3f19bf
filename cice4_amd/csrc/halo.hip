#include "halo.h"

#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
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

// ---- links without RCCL (halo.h) --------------------------------------------------------------------------------------
// The protocol of both: messages of a (sender, receiver) pair are numbered by the pair; a sender posts message m only
// after the receiver has taken m - 1 (one mailbox per pair), a receiver takes m once it is posted.  Every wait is bounded.
namespace {
constexpr int LINK_WAIT_S = 60;   // a rank whose partner never shows up fails instead of hanging
}

struct LocalLink {
  int nranks = 0;
  bool mirror = false;   // MirrorLink: no partner at all (see below)
  virtual ~LocalLink() = default;
  virtual void post(int src, int dst, const void* data, size_t bytes) = 0;   // blocks until the mailbox is free
  virtual void take(int src, int dst, void* data, size_t bytes) = 0;         // blocks until the message is there
  virtual unsigned all_max(int rank, unsigned v) = 0;
};

// ranks = contexts of one process (one host thread each)
struct InProcLink : LocalLink {
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::vector<char>> box;   // box[src * nranks + dst]
  std::vector<long> posted, taken, red_seq;
  std::vector<unsigned> red_val;
  explicit InProcLink(int n) {
    nranks = n;
    box.resize((size_t)n * n);
    posted.assign((size_t)n * n, 0);
    taken.assign((size_t)n * n, 0);
    red_seq.assign(n, 0);
    red_val.assign(n, 0);
  }
  void post(int src, int dst, const void* data, size_t bytes) override {
    std::unique_lock<std::mutex> lk(m);
    const size_t slot = (size_t)src * nranks + dst;
    if (!cv.wait_for(lk, std::chrono::seconds(LINK_WAIT_S), [&] { return taken[slot] == posted[slot]; }))
      throw Error{CICE_ECOMM, "in-process link: the partner rank did not take the previous message (is it running?)"};
    box[slot].assign((const char*)data, (const char*)data + bytes);
    posted[slot] += 1;
    cv.notify_all();
  }
  void take(int src, int dst, void* data, size_t bytes) override {
    std::unique_lock<std::mutex> lk(m);
    const size_t slot = (size_t)src * nranks + dst;
    if (!cv.wait_for(lk, std::chrono::seconds(LINK_WAIT_S), [&] { return posted[slot] == taken[slot] + 1; }))
      throw Error{CICE_ECOMM, "in-process link: no message from the partner rank (is it running?)"};
    if (box[slot].size() != bytes)
      throw Error{CICE_ECOMM, "in-process link: the ranks are not making the same sequence of halo updates"};
    std::memcpy(data, box[slot].data(), bytes);
    taken[slot] += 1;
    cv.notify_all();
  }
  unsigned all_max(int rank, unsigned v) override {
    std::unique_lock<std::mutex> lk(m);
    const long seq = red_seq[rank] + 1;
    auto all_at = [&](long q) {
      for (int r = 0; r < nranks; ++r)
        if (red_seq[r] < q) return false;
      return true;
    };
    // nobody may still be reading the previous round's values
    if (!cv.wait_for(lk, std::chrono::seconds(LINK_WAIT_S), [&] { return all_at(seq - 1); }))
      throw Error{CICE_ECOMM, "in-process link: all-reduce out of step"};
    red_val[rank] = v;
    red_seq[rank] = seq;
    cv.notify_all();
    if (!cv.wait_for(lk, std::chrono::seconds(LINK_WAIT_S), [&] { return all_at(seq); }))
      throw Error{CICE_ECOMM, "in-process link: a rank did not reach the all-reduce"};
    unsigned mx = 0;
    for (int r = 0; r < nranks; ++r) mx = std::max(mx, red_val[r]);
    return mx;
  }
};

// TIMING AID, not a communicator: ONE rank of an N-rank decomposition alone on its device.  Every message the rank
// would send comes straight back as the message it would receive from that neighbour (a slab's two neighbours get and
// send messages of the same shape), device to device, without a host round trip: the rank runs exactly the kernels,
// tile lists, pack / unpack and launch sequence it would run in the N-rank job, with nobody else on the chip -- the
// per-rank cost that a node with one GPU per rank would see, minus the link.  The RESULTS are those of a mirror
// boundary and mean nothing.  (bench.py --as-rank R --of N)
struct MirrorLink : LocalLink {
  explicit MirrorLink(int n) { nranks = n; mirror = true; }
  void post(int, int, const void*, size_t) override {}
  void take(int, int, void*, size_t) override {}
  unsigned all_max(int, unsigned v) override { return v; }
};
LocalLink* mirror_link_new(int nranks) { return new MirrorLink(nranks); }

LocalLink* local_link_get(int link_id, int nranks) {
  static std::mutex gm;
  static std::map<int, std::unique_ptr<LocalLink>> links;
  std::lock_guard<std::mutex> g(gm);
  auto& l = links[link_id];
  if (!l) l.reset(new InProcLink(nranks));
  if (l->nranks != nranks) throw Error{CICE_EINVAL, "cice_comm_init_local: this link exists with another number of ranks"};
  return l.get();
}

// ranks = PROCESSES of one host: a file under /dev/shm holds the counters and one mailbox per (sender, receiver) pair.
// For running the multi-process path (bench.py --gpus N, the Fortran MPI driver) on a box with ONE GPU, where RCCL
// refuses two ranks on one device.  Rank 0 creates the file, the others wait for it; polls sleep 20 us.
struct ShmLink : LocalLink {
  struct Header {
    std::atomic<unsigned> magic;
    unsigned nranks;
    unsigned long long box_bytes;
  };
  char* base = nullptr;
  size_t total = 0, box_bytes = 0;
  std::string name;
  bool owner = false;
  // layout: Header | posted[R*R] | taken[R*R] | size[R*R] | red_seq[R] | red_val[R] | (pad to 4096) | boxes
  std::atomic<long>* posted() const { return (std::atomic<long>*)(base + 64); }
  std::atomic<long>* taken() const { return posted() + (size_t)nranks * nranks; }
  std::atomic<unsigned long long>* sizes() const { return (std::atomic<unsigned long long>*)(taken() + (size_t)nranks * nranks); }
  std::atomic<long>* red_seq() const { return (std::atomic<long>*)(sizes() + (size_t)nranks * nranks); }
  std::atomic<unsigned>* red_val() const { return (std::atomic<unsigned>*)(red_seq() + nranks); }
  size_t head_bytes() const {
    const size_t h = 64 + ((size_t)nranks * nranks * 3 + nranks) * 8 + (size_t)nranks * 4;
    return (h + 4095) / 4096 * 4096;
  }
  char* box(size_t slot) const { return base + head_bytes() + slot * box_bytes; }
  ShmLink(const char* nm, int rank, int n, size_t bbytes) {
    nranks = n;
    name = nm;
    box_bytes = (bbytes + 4095) / 4096 * 4096;
    total = head_bytes() + (size_t)n * n * box_bytes;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LINK_WAIT_S);
    int fd = -1;
    if (rank == 0) {
      (void)shm_unlink(nm);
      fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)total) != 0) throw Error{CICE_ECOMM, std::string("shared-memory link: cannot create ") + nm};
      owner = true;
    } else {
      while ((fd = shm_open(nm, O_RDWR, 0600)) < 0) {
        if (std::chrono::steady_clock::now() > deadline) throw Error{CICE_ECOMM, std::string("shared-memory link: rank 0 never created ") + nm};
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      struct stat st;
      while (fstat(fd, &st) == 0 && (size_t)st.st_size < total) {
        if (std::chrono::steady_clock::now() > deadline) throw Error{CICE_ECOMM, "shared-memory link: the file never reached its size (do the ranks agree on it?)"};
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
    base = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) throw Error{CICE_ECOMM, "shared-memory link: mmap failed"};
    Header* h = (Header*)base;
    if (rank == 0) {   // a fresh file is zero-filled: counters start at 0
      h->nranks = (unsigned)n;
      h->box_bytes = box_bytes;
      h->magic.store(0x43494345u, std::memory_order_release);
    } else {
      while (h->magic.load(std::memory_order_acquire) != 0x43494345u) {
        if (std::chrono::steady_clock::now() > deadline) throw Error{CICE_ECOMM, "shared-memory link: rank 0 never initialised the file"};
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (h->nranks != (unsigned)n || h->box_bytes != box_bytes) throw Error{CICE_ECOMM, "shared-memory link: the ranks disagree on its shape"};
    }
  }
  ~ShmLink() override {
    if (base && base != MAP_FAILED) munmap(base, total);
    if (owner) (void)shm_unlink(name.c_str());
  }
  template <class F>
  static void wait(F&& ok, const char* what) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(LINK_WAIT_S);
    int spins = 0;
    while (!ok()) {
      if (++spins < 200) continue;
      if (std::chrono::steady_clock::now() > deadline) throw Error{CICE_ECOMM, what};
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }
  void post(int src, int dst, const void* data, size_t bytes) override {
    if (bytes > box_bytes) throw Error{CICE_ECOMM, "shared-memory link: message larger than the mailbox (cice_comm_init_shm: box_bytes)"};
    const size_t slot = (size_t)src * nranks + dst;
    wait([&] { return taken()[slot].load(std::memory_order_acquire) == posted()[slot].load(std::memory_order_relaxed); },
         "shared-memory link: the partner rank did not take the previous message (is it running?)");
    std::memcpy(box(slot), data, bytes);
    sizes()[slot].store(bytes, std::memory_order_relaxed);
    posted()[slot].fetch_add(1, std::memory_order_release);
  }
  void take(int src, int dst, void* data, size_t bytes) override {
    const size_t slot = (size_t)src * nranks + dst;
    wait([&] { return posted()[slot].load(std::memory_order_acquire) == taken()[slot].load(std::memory_order_relaxed) + 1; },
         "shared-memory link: no message from the partner rank (is it running?)");
    if (sizes()[slot].load(std::memory_order_relaxed) != bytes)
      throw Error{CICE_ECOMM, "shared-memory link: the ranks are not making the same sequence of halo updates"};
    std::memcpy(data, box(slot), bytes);
    taken()[slot].fetch_add(1, std::memory_order_release);
  }
  unsigned all_max(int rank, unsigned v) override {
    const long seq = red_seq()[rank].load(std::memory_order_relaxed) + 1;
    auto all_at = [&](long q) {
      for (int r = 0; r < nranks; ++r)
        if (red_seq()[r].load(std::memory_order_acquire) < q) return false;
      return true;
    };
    wait([&] { return all_at(seq - 1); }, "shared-memory link: all-reduce out of step");
    red_val()[rank].store(v, std::memory_order_relaxed);
    red_seq()[rank].store(seq, std::memory_order_release);
    wait([&] { return all_at(seq); }, "shared-memory link: a rank did not reach the all-reduce");
    unsigned mx = 0;
    for (int r = 0; r < nranks; ++r) mx = std::max(mx, red_val()[r].load(std::memory_order_relaxed));
    return mx;
  }
};

LocalLink* shm_link_open(const char* name, int rank, int nranks, size_t box_bytes) {
  return new ShmLink(name, rank, nranks, box_bytes);   // owned by the context (cice_destroy)
}
void link_close(LocalLink* l) { delete l; }

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

// The whole fold of a rank that holds every top-row block in ONE workgroup: buffer in LDS, fill -> gather -> symmetry ->
// copy out with barriers in between -- one launch instead of four inside every subcycle of a tripole grid (each of the
// four is a few hundred elements).  Same lists, same arithmetic, same order per element.
template <class T>
__global__ __launch_bounds__(1024) void k_fold_one(T* __restrict__ base, int nfields, size_t stride,
                                                   const int32_t* __restrict__ lsrc, const int32_t* __restrict__ bidx,
                                                   int nsrc, const int32_t* __restrict__ lo, const int32_t* __restrict__ hi,
                                                   int npair, const int32_t* __restrict__ dst,
                                                   const int32_t* __restrict__ src, int nout, int bstride, int sgn, T fill) {
  extern __shared__ unsigned char fold_lds[];
  T* buf = reinterpret_cast<T*>(fold_lds);
  const int nt = blockDim.x, t0 = threadIdx.x;
  for (int t = t0; t < bstride * nfields; t += nt) buf[t] = fill;
  __syncthreads();
  for (int t = t0; t < nsrc * nfields; t += nt) {
    const int k = t / nsrc, e = t - k * nsrc;
    buf[(size_t)k * bstride + bidx[e]] = base[(size_t)k * stride + lsrc[e]];
  }
  __syncthreads();
  for (int t = t0; t < npair * nfields; t += nt) {
    const int k = t / npair, e = t - k * npair;
    T* b = buf + (size_t)k * bstride;
    const T xavg = fold_avg(b[lo[e]], b[hi[e]], sgn);
    b[lo[e]] = xavg;
    b[hi[e]] = (T)sgn * xavg;
  }
  __syncthreads();
  for (int t = t0; t < nout * nfields; t += nt) {
    const int k = t / nout, e = t - k * nout;
    base[(size_t)k * stride + dst[e]] = (T)sgn * buf[(size_t)k * bstride + src[e]];
  }
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
  fold_rows_ = d.fold_rows();
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
  if (L.mirror) {   // what would be sent comes back as what would be received (message m of either list: the same neighbour)
    for (int m = 0; m < std::min(ns, nr); ++m)
      CICE_HIP(hipMemcpyAsync(rb + (size_t)nfields * roff[m], sb + (size_t)nfields * soff[m],
                              (size_t)nfields * std::min(scnt[m], rcnt[m]) * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    return;
  }
  CICE_HIP(hipStreamSynchronize(stream_));   // the pack kernel has run
  std::vector<char> tmp;
  for (int m = 0; m < ns; ++m) {
    const size_t bytes = (size_t)nfields * scnt[m] * sizeof(T);
    tmp.resize(bytes);
    CICE_HIP(hipMemcpy(tmp.data(), sb + (size_t)nfields * soff[m], bytes, hipMemcpyDeviceToHost));
    L.post(rank_, speer[m], tmp.data(), bytes);
  }
  for (int m = 0; m < nr; ++m) {
    const size_t bytes = (size_t)nfields * rcnt[m] * sizeof(T);
    tmp.resize(bytes);
    L.take(rpeer[m], rank_, tmp.data(), bytes);
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
void Halo::update(T* base, int nfields, size_t stride, bool wrap, int loc, int kind, T fill, int parts) {
  CICE_REQUIRE(nfields >= 1, "halo: no field");
  if (parts & HALO_COPIES) update_copies<T>(base, nfields, stride, wrap, fill);
  if (parts & HALO_FOLD) update_fold<T>(base, nfields, stride, loc, kind, fill);
  CICE_HIP(hipGetLastError());
}

template <class T>
void Halo::update_copies(T* base, int nfields, size_t stride, bool wrap, T fill) {
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
}

template <class T>
void Halo::update_fold(T* base, int nfields, size_t stride, int loc, int kind, T fill) {
  if (fold_ || nfsend_) {   // tripole north boundary, after all regular copies (serial/ice_boundary.F90:705)
    CICE_REQUIRE(loc >= LOC_CENTER && loc <= LOC_EFACE, "halo: field location unknown on a tripole grid");
    CICE_REQUIRE(kind >= KIND_SCALAR && kind <= KIND_ANGLE, "halo: field kind unknown on a tripole grid");
    const int sgn = kind == KIND_SCALAR ? 1 : -1;
    const int bstride = fold_rows_ * nxg_;
    T* buf = nullptr;
    if (fold_ && !nfsend_ && !nfrecv_ && (size_t)bstride * nfields * sizeof(T) <= 64 * 1024) {
      const int l = loc - 1;
      hipLaunchKernelGGL(k_fold_one<T>, dim3(1), dim3(1024), (size_t)bstride * nfields * sizeof(T), stream_, base, nfields,
                         stride, fold_lsrc_.p, fold_bidx_.p, nfold_src_, fold_lo_[l].p, fold_hi_[l].p, nfold_pair_[l],
                         fold_dst_[l].p, fold_src_[l].p, nfold_out_[l], bstride, sgn, fill);
      CICE_HIP(hipGetLastError());
      return;
    }
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
}

void Halo::all_max_u32(unsigned* dev_word) {
  if (nranks_ <= 1) return;
  if (link_) {   // every rank posts its word and takes the maximum once all have
    unsigned v = 0;
    CICE_HIP(hipStreamSynchronize(stream_));
    CICE_HIP(hipMemcpy(&v, dev_word, 4, hipMemcpyDeviceToHost));
    const unsigned mx = link_->all_max(rank_, v);
    CICE_HIP(hipMemcpy(dev_word, &mx, 4, hipMemcpyHostToDevice));
    return;
  }
  if (!comm_) return;
  CICE_NCCL(ncclAllReduce(dev_word, dev_word, 1, ncclUint32, ncclMax, (ncclComm_t)comm_, stream_));
}

void Halo::update_r8(double* base, int nfields, size_t stride, bool wrap, int loc, int kind, double fill, int parts) {
  update<double>(base, nfields, stride, wrap, loc, kind, fill, parts);
}
void Halo::update_i4(int32_t* base, int nfields, size_t stride, int loc, int kind, int32_t fill) {
  update<int32_t>(base, nfields, stride, true, loc, kind, fill, HALO_ALL);
}
void Halo::update_r4(float* base, int nfields, size_t stride, int loc, int kind, float fill) {
  update<float>(base, nfields, stride, true, loc, kind, fill, HALO_ALL);
}

}  // namespace cice
