// C-ABI of libcice4_amd.so (include/cice4_amd.h).  Exceptions never cross the
// boundary: every entry returns a status code and records the message.
#include <rccl/rccl.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "domain.h"
#include "evp.h"
#include "halo.h"
#include "atmo.h"
#include "therm.h"
#include "transport.h"

using namespace cice;

struct cice_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  Domain dom;
  bool have_domain = false;
  std::unique_ptr<Halo> halo;
  std::unique_ptr<Evp> evp;
  std::unique_ptr<Transport> transport;
  std::unique_ptr<Upwind> upwind;
  // RCCL communicator of this rank (cice_comm_init): created once, handed to every Halo built afterwards --
  // the block decomposition may change (cice_domain_create*), the set of ranks does not
  ncclComm_t comm = nullptr;
  int comm_rank = -1, comm_nranks = 0;
  CopyFan fan;                 // side streams for the entries that move many separate host arrays: ONE set per context,
                               // shared by the dynamics, the thermodynamic half-step and the transport
  hipStream_t cs() { return fan.forked ? fan.next() : stream; }   // the stream for the next host <-> device copy
  // evp -> transport chain (cice_transport_chain): the host arrays the transport calls will be given; chain_ready: a
  // cice_evp call has prefetched them and no transport call has consumed that yet
  cice_transport_fields chain{};
  bool chain_on = false, chain_ready = false;
  const double *chain_aicen = nullptr, *chain_vicen = nullptr, *chain_u = nullptr, *chain_v = nullptr;   // what cice_evp was given
  double chio = 0.006;         // coupled flavour: the namelist's chio (cice_thermo_set_chio)
  double nml[4] = {1.0, 0.0, 0.00536, 0.0};   // coupled flavour: cosw, sinw, dragio, use_ocnslope last sent to the device
  bool nml_set = false;
  LocalLink* link = nullptr;   // stand-in for the communicator without RCCL (cice_comm_init_local / _shm; tests)
  bool link_owned = false;     // the shared-memory form belongs to this context
  // Page-locked host ranges of this context: [start, end) in bytes, disjoint.  One manager for the explicit
  // registrations (cice_host_register, cice_evp_pin_fields): a new range that touches registered ones is registered
  // as their union (a whole array after some of its slices), because a copy whose host range is partly registered
  // is refused by the runtime.  CICE4_AMD_PIN=0 in the environment leaves everything pageable (diagnostic).
  // Exact byte ranges, not page-rounded, for the same reason (a neighbouring variable sharing the last page).
  std::vector<std::pair<uintptr_t, uintptr_t>> pin_ranges;
  std::vector<std::pair<uintptr_t, uintptr_t>> pin_refused;  // ranges hipHostRegister turned down (not retried)
  void pin_range(const void* p, size_t bytes) {
    if (!p || !bytes) return;
    static const bool off = [] { const char* e = std::getenv("CICE4_AMD_PIN"); return e && e[0] == '0'; }();
    if (off) return;
    uintptr_t lo = (uintptr_t)p, hi = (uintptr_t)p + bytes;
    for (const auto& r : pin_ranges)
      if (lo >= r.first && hi <= r.second) return;            // already inside a registered range
    for (const auto& r : pin_refused)
      if (lo >= r.first && hi <= r.second) return;
    const uintptr_t lo0 = lo, hi0 = hi;
    std::vector<std::pair<uintptr_t, uintptr_t>> released;   // registered ranges the new one touches
    for (size_t k = 0; k < pin_ranges.size();) {
      if (pin_ranges[k].first <= hi0 && lo0 <= pin_ranges[k].second) {
        lo = std::min(lo, pin_ranges[k].first);
        hi = std::max(hi, pin_ranges[k].second);
        if (released.empty()) (void)hipDeviceSynchronize();    // no copy may be in flight on a range being released
        if (hipHostUnregister((void*)pin_ranges[k].first) != hipSuccess) (void)hipGetLastError();
        released.push_back(pin_ranges[k]);
        pin_ranges.erase(pin_ranges.begin() + k);
      } else {
        ++k;
      }
    }
    if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
      pin_ranges.push_back({lo, hi});
      return;
    }
    // registered by somebody else, or not registrable: what was page-locked before stays page-locked (the union is
    // all or nothing for the runtime, not for us), only the request itself stays pageable and is not asked for again
    (void)hipGetLastError();
    for (const auto& r : released) {
      if (hipHostRegister((void*)r.first, r.second - r.first, hipHostRegisterDefault) == hipSuccess) pin_ranges.push_back(r);
      else (void)hipGetLastError();
    }
    pin_refused.push_back({lo0, hi0});
  }
  void unpin_all() {
    for (const auto& r : pin_ranges)
      if (hipHostUnregister((void*)r.first) != hipSuccess) (void)hipGetLastError();
    pin_ranges.clear();
    pin_refused.clear();
  }
  // staging of the host-pointer entries (thermo_vertical is called ncat x nblocks times per step with
  // the same block size: allocated once, grown only when a larger block comes along)
  DevBuf<double> tv_stage, fz_stage, halo_stage;
  // page-locked gather buffer and cell offsets of the compact thermo_vertical path
  void* tv_host = nullptr;
  size_t tv_host_bytes = 0;
  std::vector<size_t> tv_cells;
  // frame of the rank's blocks (cells a halo update can read or write), for host-array halo updates
  std::vector<int32_t> frame;
  std::vector<size_t> frame_at;
  std::unique_ptr<Halo> frame_halo;   // the domain's lists re-addressed to positions in the gathered frame
  DevBuf<double> frame_pack;
  void* frame_host = nullptr;
  size_t frame_host_bytes = 0;
  DevBuf<int32_t> tv_list;
  // thermo
  ThermoParams tp{};
  bool have_thermo = false;
  DevBuf<unsigned long long> tkey;  // THERMO_STATUS_WORDS: [0] error key, then the update counters (therm.h)
  // batched thermo state
  struct Batch {
    int nx = 0, ny = 0, nb = 0;
    DevBuf<int32_t> blk;
    DevBuf<double> aicen, trcrn, vicen, vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, fbot, Tbot,
        lhcoef, shcoef, fswsfc, fswint, fswthrun, Sswabs, Iswabs, out15, mlt_onset, frz_onset;
    DevBuf<double> mrg_in, mrg_acc, fz_in;   // merge_fluxes inputs / accumulators, frzmlt inputs + rside
    DevBuf<int32_t> perm;                    // columns of every chunk sorted by expected work (k_thermo_sort)
    int sort_chunk = 0, sort_group = 8;      // chunk 0: no sorting (k_thermo_dense) -- the default: DESIGN.md 3.3
    DevBuf<unsigned char> niter;             // solver iterations of every (cell, category) in the last step
    DevBuf<double> atm_in;                   // uatm, vatm, wind, zlvl, strax, stray (cice_step_therm1_abl)
    std::vector<int32_t> hblk;               // ilo, ihi, jlo, jhi per block (host copy of blk)
  } tb;
  // device is required lazily: domain queries work on a CPU-only host
  // Every C-ABI entry binds the calling thread to this context's device first (CICE_TRY): the host
  // process may have changed the current device since the last call (another context on another GPU,
  // torch.cuda.set_device, another thread).
  void bind_device() {
    if (stream) CICE_HIP(hipSetDevice(device));
  }
  void need_device() {
    if (stream) {
      CICE_HIP(hipSetDevice(device));
      return;
    }
    int cnt = 0;
    CICE_HIP(hipGetDeviceCount(&cnt));
    if (cnt < 1) throw Error{CICE_EDEVICE, "no HIP device visible"};
    if (device >= 0) CICE_HIP(hipSetDevice(device));
    else CICE_HIP(hipGetDevice(&device));
    CICE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    // one operation at once: the runtime binds a stream to a hardware queue when it first has work, and contexts that share a
    // device (ranks of a rehearsal on one GPU, whose one-launch loops wait for each other) need their MAIN streams on queues
    // of their own -- created, and bound, one after the other (tests/ranks_case.py)
    void* p = nullptr;
    CICE_HIP(hipMalloc(&p, 64));
    CICE_HIP(hipMemsetAsync(p, 0, 64, stream));
    CICE_HIP(hipStreamSynchronize(stream));
    CICE_HIP(hipFree(p));
  }
  void need_halo() {
    need_device();
    if (!halo) {
      CICE_REQUIRE(have_domain, "cice_domain_create has not been called");
      halo.reset(new Halo());
      frame_halo.reset();
      halo->init(dom, stream);
      if (comm) halo->set_comm((ncclComm*)comm, comm_rank, comm_nranks);
      if (link) halo->set_link(link, comm_rank, comm_nranks);
    }
  }
};

static std::string g_create_err;

#define CICE_TRY(ctx_) \
  cice_ctx* c_ = (ctx_); \
  if (!c_) return CICE_EINVAL; \
  try {                        \
    c_->bind_device();         \
    c_->fan.forked = false;   /* an entry that failed between fork and join leaves nothing behind for the next */
#define CICE_CATCH                                            \
  }                                                           \
  catch (const Error& e) {                                    \
    c_->err = e.msg;                                          \
    return e.code;                                            \
  }                                                           \
  catch (const std::exception& e) {                           \
    c_->err = e.what();                                       \
    return CICE_EINVAL;                                       \
  }                                                           \
  return CICE_OK;

// Calibration stream for the HBM counters: one 8-byte load and one 8-byte store per lane,
// the access width of the hot kernels (MI355X_MICROARCH.md: FETCH_SIZE is calibrated for
// 16-B lanes only, other widths must be calibrated on a known byte count).
__global__ __launch_bounds__(256) void k_diag_copy8(const double* __restrict__ src,
                                                    double* __restrict__ dst, size_t n) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = src[t] + 1.0;
}

__global__ __launch_bounds__(256) void k_diag_copy16(const double2* __restrict__ src,
                                                     double2* __restrict__ dst, size_t n2) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n2) {
    double2 v = src[t];
    v.x += 1.0;
    dst[t] = v;
  }
}

// Host-pointer form of ice_HaloUpdate (what rccl/ice_boundary.F90 calls with a module array): the field is
// staged through a persistent device buffer (grown only when a larger field comes along) and ALL its levels
// travel in one update = one message per neighbour (bound_state's 65 levels included, ice_state.F90:162-217).
template <class T>
static void halo_apply(cice_ctx* c, T* d, int nlev, size_t n, int loc, int kind, double fill) {
  if (std::is_same<T, double>::value)
    c->halo->update_r8(reinterpret_cast<double*>(d), nlev, n, true, loc, kind, fill);
  else if (std::is_same<T, float>::value)
    c->halo->update_r4(reinterpret_cast<float*>(d), nlev, n, loc, kind, (float)fill);
  else
    c->halo->update_i4(reinterpret_cast<int32_t*>(d), nlev, n, loc, kind, (int32_t)fill);
}

template <class T>
static void halo_host(cice_ctx* c, T* field, int nlev, int loc = LOC_CENTER, int kind = KIND_SCALAR, double fill = 0.0) {
  c->need_halo();
  CICE_REQUIRE(field && nlev >= 1, "bad argument");
  const size_t n = (size_t)c->dom.nblocks() * c->dom.nx_block * c->dom.ny_block;
  const size_t words = (n * nlev * sizeof(T) + 7) / 8;
  if (c->halo_stage.n < words) c->halo_stage.alloc(words);
  T* d = reinterpret_cast<T*>(c->halo_stage.p);
  CICE_HIP(hipMemcpyAsync(d, field, n * nlev * sizeof(T), hipMemcpyHostToDevice, c->stream));
  halo_apply<T>(c, d, nlev, n, loc, kind, fill);
  CICE_HIP(hipMemcpyAsync(field, d, n * nlev * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  CICE_HIP(hipStreamSynchronize(c->stream));
}

// The same for a field in the reference's own array layout (nx_block, ny_block, nz, nblocks) -- block outermost,
// what ice_HaloUpdate3D/4D receive: strided copies to and from the level-major device layout replace the
// repacking on the host.
//
// Only the cells a halo update can read or write travel: the FRAME of the rank's blocks (physical edge cells and ghost
// cells: every address that occurs in a copy, fill, message or fold list of the domain; ~4 (nx + ny) of the nx * ny
// cells of a block).  The host gathers the frame into a page-locked buffer (a few thousand elements per level), one
// copy takes it to the device, the update runs ON THE GATHERED BUFFER (a second Halo whose lists address positions
// in the frame instead of cells of the field), one copy brings it back and the host scatters it.  At gx1 a 2-D update moves 22 KB each way instead of 1 MB, a 25-level
// one 0.6 MB instead of 25 -- the reference's own timer of ice_HaloUpdate (timer_bound) in the whole model fell
// accordingly.
static void frame_build(cice_ctx* c) {
  const Domain& dm = c->dom;
  const size_t n = (size_t)dm.nblocks() * dm.nx_block * dm.ny_block;
  std::vector<char> mark(n, 0);
  auto add = [&](const std::vector<int32_t>& v) {
    for (int32_t a : v)
      if (a >= 0 && (size_t)a < n) mark[a] = 1;
  };
  add(dm.hsrc); add(dm.hdst); add(dm.hfill); add(dm.rsrc); add(dm.rdst); add(dm.fold_lsrc);
  for (const HaloMsg& m : dm.send) add(m.addr);
  for (const HaloMsg& m : dm.recv) add(m.addr);
  for (const HaloMsg& m : dm.fold_send) add(m.addr);
  for (int l = 0; l < 4; ++l) add(dm.fold_out[l].dst);
  c->frame.clear();
  std::vector<int32_t> pos(n, -1);
  for (size_t a = 0; a < n; ++a)
    if (mark[a]) {
      pos[a] = (int32_t)c->frame.size();
      c->frame.push_back((int32_t)a);
    }
  // the same lists with every field address replaced by its position in the gathered frame: the update then runs on
  // the gathered buffer itself (level stride = frame size), copies, fills, messages and folds alike
  Domain fd = dm;
  auto remap = [&](std::vector<int32_t>& v) {
    for (int32_t& a : v)
      if (a >= 0 && (size_t)a < n) a = pos[a];
  };
  remap(fd.hsrc); remap(fd.hdst); remap(fd.hfill); remap(fd.rsrc); remap(fd.rdst); remap(fd.fold_lsrc);
  for (HaloMsg& m : fd.send) remap(m.addr);
  for (HaloMsg& m : fd.recv) remap(m.addr);
  for (HaloMsg& m : fd.fold_send) remap(m.addr);
  for (int l = 0; l < 4; ++l) remap(fd.fold_out[l].dst);
  c->frame_halo.reset(new Halo());
  c->frame_halo->init(fd, c->stream);
  if (c->comm) c->frame_halo->set_comm((ncclComm*)c->comm, c->comm_rank, c->comm_nranks);
  if (c->link) c->frame_halo->set_link(c->link, c->comm_rank, c->comm_nranks);
  CICE_HIP(hipStreamSynchronize(c->stream));
}

template <class T>
static void halo_apply_on(Halo& h, T* d, int nlev, size_t n, int loc, int kind, double fill) {
  if (std::is_same<T, double>::value) h.update_r8(reinterpret_cast<double*>(d), nlev, n, true, loc, kind, fill);
  else if (std::is_same<T, float>::value) h.update_r4(reinterpret_cast<float*>(d), nlev, n, loc, kind, (float)fill);
  else h.update_i4(reinterpret_cast<int32_t*>(d), nlev, n, loc, kind, (int32_t)fill);
}

// A HOST array on a domain without messages (one rank: every ghost cell mirrors a cell of the same array, takes the
// fill value or comes out of the tripole fold): the update is a few thousand element copies inside the caller's own
// array, done right here on the host from the domain's lists -- what serial/ice_boundary.F90:591-873 does, in the
// order Halo::update works (copy list, fill list, refresh list, fold).  No device round trip: the whole model's Bound
// timer is back at the reference's.  Device-resident fields (cice_halo_update_dev_*) and
// domains with off-rank neighbours keep the device path.
template <class T>
static T fold_avg_host(T x1, T x2, int sgn) {
  if (std::is_same<T, int32_t>::value) return (T)std::round(0.5 * (double)(x1 + sgn * x2));   // nint()
  return (T)0.5 * (x1 + (T)sgn * x2);
}

// strides (in elements) of the caller's array: level (z1, z2) of block b starts at b * sb + z2 * s2 + z1 * s1; the
// contiguous (nx, ny, nz, nblocks) array is nz1 = nz, s1 = np, nz2 = 1, sb = nz * np
struct LevelStrides { int nz1, nz2; size_t s1, s2, sb; };

template <class T>
static void halo_host_lists(const Domain& dm, T* field, const LevelStrides& ls, int loc, int kind, T fill) {
  const size_t np = (size_t)dm.nx_block * dm.ny_block;
  // list address (level-major numbering: block * np + cell) -> element of level 0 in the caller's layout
  auto at = [&](int32_t a) { const size_t b = (size_t)a / np; return b * ls.sb + ((size_t)a - b * np); };
  const int nz = ls.nz1 * ls.nz2;
  const bool fold = dm.fold;
  if (fold) {
    CICE_REQUIRE(loc >= LOC_CENTER && loc <= LOC_EFACE, "halo: field location unknown on a tripole grid");
    CICE_REQUIRE(kind >= KIND_SCALAR && kind <= KIND_ANGLE, "halo: field kind unknown on a tripole grid");
  }
  const int sgn = kind == KIND_SCALAR ? 1 : -1;
  std::vector<T> buf(fold ? (size_t)dm.fold_rows() * dm.nxg : 0);
  for (int z = 0; z < nz; ++z) {
    T* f = field + (size_t)(z % ls.nz1) * ls.s1 + (size_t)(z / ls.nz1) * ls.s2;
    for (size_t e = 0; e < dm.hsrc.size(); ++e) f[at(dm.hdst[e])] = f[at(dm.hsrc[e])];
    for (int32_t a : dm.hfill) f[at(a)] = fill;
    for (size_t e = 0; e < dm.rsrc.size(); ++e) f[at(dm.rdst[e])] = f[at(dm.rsrc[e])];
    if (fold) {
      const int l = loc - 1;
      std::fill(buf.begin(), buf.end(), fill);
      for (size_t e = 0; e < dm.fold_lsrc.size(); ++e) buf[dm.fold_bidx[e]] = f[at(dm.fold_lsrc[e])];
      for (size_t e = 0; e < dm.fold_lo[l].size(); ++e) {
        const int32_t lo = dm.fold_lo[l][e], hi = dm.fold_hi[l][e];
        const T x = fold_avg_host<T>(buf[lo], buf[hi], sgn);
        buf[lo] = x;
        buf[hi] = (T)sgn * x;
      }
      const Domain::FoldOut& fo = dm.fold_out[l];
      for (size_t e = 0; e < fo.dst.size(); ++e) f[at(fo.dst[e])] = (T)sgn * buf[fo.src[e]];
    }
  }
}

static bool domain_has_messages(const Domain& dm) {
  return !dm.send.empty() || !dm.recv.empty() || !dm.fold_send.empty() || !dm.fold_recv.empty();
}

template <class T>
static void halo_host_blocked(cice_ctx* c, T* field, int nz, int loc, int kind, double fill) {
  CICE_REQUIRE(field && nz >= 1, "bad argument");
  CICE_REQUIRE(c->have_domain, "cice_domain_create has not been called");
  static const bool force_dev = std::getenv("CICE4_AMD_HALO_HOST_ON_DEVICE") != nullptr;   // test aid: the frame path
  if (!domain_has_messages(c->dom) && !force_dev) {
    const size_t np_ = (size_t)c->dom.nx_block * c->dom.ny_block;
    halo_host_lists<T>(c->dom, field, LevelStrides{nz, 1, np_, 0, (size_t)nz * np_}, loc, kind, (T)fill);
    return;
  }
  c->need_halo();
  const int nb = c->dom.nblocks();
  const size_t np = (size_t)c->dom.nx_block * c->dom.ny_block, n = np * nb;
  if (!c->frame_halo) frame_build(c);   // the frame belongs to the domain: dropped by cice_domain_create*
  const size_t nc = c->frame.size();
  if (nc > 0 && nc * 2 <= n) {   // the frame is the smaller part of the field: move only the frame
    const size_t cnt = nc * nz, bytes = cnt * sizeof(T);
    if (c->frame_host_bytes < bytes) {
      if (c->frame_host) (void)hipHostFree(c->frame_host);
      c->frame_host = nullptr;
      c->frame_host_bytes = 0;
      CICE_HIP(hipHostMalloc(&c->frame_host, bytes + bytes / 2, hipHostMallocDefault));
      c->frame_host_bytes = bytes + bytes / 2;
    }
    if (c->frame_pack.n < (bytes + 7) / 8) c->frame_pack.alloc((bytes + 7) / 8);
    T* hp = static_cast<T*>(c->frame_host);
    T* dp = reinterpret_cast<T*>(c->frame_pack.p);
    const int32_t* cell = c->frame.data();
    std::vector<size_t>& at = c->frame_at;     // element (level 0) of every frame cell in the caller's layout
    at.resize(nc);
    for (size_t k = 0; k < nc; ++k) {
      const size_t b = (size_t)cell[k] / np, q = (size_t)cell[k] - b * np;
      at[k] = b * nz * np + q;
    }
    for (int z = 0; z < nz; ++z) {
      T* out = hp + (size_t)z * nc;
      const T* src = field + (size_t)z * np;
      for (size_t k = 0; k < nc; ++k) out[k] = src[at[k]];
    }
    CICE_HIP(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, c->stream));
    halo_apply_on<T>(*c->frame_halo, dp, nz, nc, loc, kind, fill);
    CICE_HIP(hipGetLastError());
    CICE_HIP(hipMemcpyAsync(hp, dp, bytes, hipMemcpyDeviceToHost, c->stream));
    CICE_HIP(hipStreamSynchronize(c->stream));
    for (int z = 0; z < nz; ++z) {
      const T* in = hp + (size_t)z * nc;
      T* dst = field + (size_t)z * np;
      for (size_t k = 0; k < nc; ++k) dst[at[k]] = in[k];
    }
    return;
  }
  const size_t words = (n * nz * sizeof(T) + 7) / 8;
  if (c->halo_stage.n < words) c->halo_stage.alloc(words);
  T* d = reinterpret_cast<T*>(c->halo_stage.p);
  if (nz == 1 || nb == 1) {
    CICE_HIP(hipMemcpyAsync(d, field, n * nz * sizeof(T), hipMemcpyHostToDevice, c->stream));
  } else {
    for (int b = 0; b < nb; ++b)
      CICE_HIP(hipMemcpy2DAsync(d + (size_t)b * np, n * sizeof(T), field + (size_t)b * nz * np, np * sizeof(T),
                                np * sizeof(T), nz, hipMemcpyHostToDevice, c->stream));
  }
  halo_apply<T>(c, d, nz, n, loc, kind, fill);
  if (nz == 1 || nb == 1) {
    CICE_HIP(hipMemcpyAsync(field, d, n * nz * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  } else {
    for (int b = 0; b < nb; ++b)
      CICE_HIP(hipMemcpy2DAsync(field + (size_t)b * nz * np, np * sizeof(T), d + (size_t)b * np, n * sizeof(T),
                                np * sizeof(T), nz, hipMemcpyDeviceToHost, c->stream));
  }
  CICE_HIP(hipStreamSynchronize(c->stream));
}

// The same for a SECTION of a 4-d module array, e.g. trcrn(:,:,1:ntrcr,:,:) in bound_state (source/ice_state.F90:206): the
// horizontal planes are whole, the levels (z1, z2) and the blocks are strided.  On a one-rank domain the lists are
// applied in place (no copy of the section: that copy was most of the model's Bound timer); otherwise the section is
// gathered into a contiguous array, updated by the general path and scattered back.
template <class T>
static void halo_host_strided(cice_ctx* c, T* field, const LevelStrides& ls, int loc, int kind, double fill) {
  CICE_REQUIRE(field && ls.nz1 >= 1 && ls.nz2 >= 1, "bad argument");
  CICE_REQUIRE(c->have_domain, "cice_domain_create has not been called");
  const size_t np = (size_t)c->dom.nx_block * c->dom.ny_block;
  const int nb = c->dom.nblocks(), nz = ls.nz1 * ls.nz2;
  static const bool force_dev = std::getenv("CICE4_AMD_HALO_HOST_ON_DEVICE") != nullptr;
  if (!domain_has_messages(c->dom) && !force_dev) {
    halo_host_lists<T>(c->dom, field, ls, loc, kind, (T)fill);
    return;
  }
  std::vector<T> tmp((size_t)nb * nz * np);
  for (int b = 0; b < nb; ++b)
    for (int z = 0; z < nz; ++z)
      std::memcpy(tmp.data() + ((size_t)b * nz + z) * np,
                  field + (size_t)b * ls.sb + (size_t)(z / ls.nz1) * ls.s2 + (size_t)(z % ls.nz1) * ls.s1, np * sizeof(T));
  halo_host_blocked<T>(c, tmp.data(), nz, loc, kind, fill);
  for (int b = 0; b < nb; ++b)
    for (int z = 0; z < nz; ++z)
      std::memcpy(field + (size_t)b * ls.sb + (size_t)(z / ls.nz1) * ls.s2 + (size_t)(z % ls.nz1) * ls.s1,
                  tmp.data() + ((size_t)b * nz + z) * np, np * sizeof(T));
}

// Device-resident form: the field already lives in device memory (nlev levels of nblocks*ny_block*nx_block
// elements, level stride = one such plane set); nothing crosses PCIe, no allocation, asynchronous on the
// library's stream.
template <class T>
static void halo_dev(cice_ctx* c, T* dev_field, int nlev, int loc = LOC_CENTER, int kind = KIND_SCALAR, double fill = 0.0) {
  c->need_halo();
  CICE_REQUIRE(dev_field && nlev >= 1, "bad argument");
  hipPointerAttribute_t at{};
  CICE_REQUIRE(hipPointerGetAttributes(&at, dev_field) == hipSuccess && at.type == hipMemoryTypeDevice,
               "cice_halo_update_dev: not a device pointer");
  const size_t n = (size_t)c->dom.nblocks() * c->dom.nx_block * c->dom.ny_block;
  halo_apply<T>(c, dev_field, nlev, n, loc, kind, fill);
}

extern "C" {

int cice_create(cice_ctx** ctx, int device) {
  if (!ctx) return CICE_EINVAL;
  try {
    *ctx = new cice_ctx();
    (*ctx)->device = device;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return CICE_EINVAL;
  }
  return CICE_OK;
}

int cice_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int cice_host_register(cice_ctx* ctx, void* host, size_t bytes) {
  CICE_TRY(ctx)
  CICE_REQUIRE(host && bytes, "NULL array");
  c_->need_device();
  c_->pin_range(host, bytes);
  CICE_CATCH
}

// Undo every cice_host_register / cice_evp_pin_fields of this context.  Page-locked host ranges MUST be
// released before the host frees that memory: the runtime keeps treating the range as DMA-able, and a later
// allocation that lands there is read through a stale mapping (GPU memory access fault).
int cice_host_unregister_all(cice_ctx* ctx) {
  CICE_TRY(ctx)
  if (c_->stream) CICE_HIP(hipStreamSynchronize(c_->stream));
  c_->unpin_all();
  CICE_CATCH
}

int cice_destroy(cice_ctx* ctx) {
  if (!ctx) return CICE_EINVAL;
  if (ctx->frame_host) (void)hipHostFree(ctx->frame_host);
  if (ctx->tv_host) (void)hipHostFree(ctx->tv_host);
  ctx->unpin_all();
  ctx->evp.reset();
  ctx->transport.reset();
  ctx->chain_on = ctx->chain_ready = false;
  ctx->upwind.reset();
  ctx->frame_halo.reset();
  ctx->halo.reset();
  if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
  if (ctx->link && ctx->link_owned) link_close(ctx->link);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return CICE_OK;
}

// The library is compiled for one set of ice_domain_size parameters (CICE_NCAT, ...): every stride of the
// category / layer / tracer dimensions of the caller's module arrays is derived from them.  A host model built
// with other sizes must not get past its init calls.
int cice_check_sizes(cice_ctx* ctx, int ncat, int nilyr, int nslyr, int max_ntrcr) {
  CICE_TRY(ctx)
  if (ncat != NCAT || nilyr != NILYR || nslyr != NSLYR || max_ntrcr != NTRCR)
    throw Error{CICE_EINVAL, "libcice4_amd is built for ncat=" + std::to_string(NCAT) + " nilyr=" + std::to_string(NILYR) +
                                 " nslyr=" + std::to_string(NSLYR) + " max_ntrcr=" + std::to_string(NTRCR) +
                                 "; the host model has ncat=" + std::to_string(ncat) + " nilyr=" + std::to_string(nilyr) +
                                 " nslyr=" + std::to_string(nslyr) + " max_ntrcr=" + std::to_string(max_ntrcr) +
                                 " (rebuild the library with matching CICE_NCAT / CICE_NILYR / CICE_NSLYR / CICE_MAX_NTRCR)"};
  CICE_CATCH
}

const char* cice_last_error(const cice_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int cice_diag_stream_copy(cice_ctx* ctx, long long n_doubles, float* elapsed_ms) {
  CICE_TRY(ctx)
  CICE_REQUIRE(n_doubles > 0, "bad size");
  c_->need_device();
  DevBuf<double> a, b;
  a.alloc((size_t)n_doubles);
  b.alloc((size_t)n_doubles);
  a.zero(c_->stream);
  hipEvent_t e0, e1;
  CICE_HIP(hipEventCreate(&e0));
  CICE_HIP(hipEventCreate(&e1));
  const dim3 g((unsigned)(((size_t)n_doubles + 255) / 256));
  hipLaunchKernelGGL(k_diag_copy8, g, dim3(256), 0, c_->stream, (const double*)a.p, b.p, (size_t)n_doubles);
  CICE_HIP(hipEventRecord(e0, c_->stream));
  hipLaunchKernelGGL(k_diag_copy8, g, dim3(256), 0, c_->stream, (const double*)a.p, b.p, (size_t)n_doubles);
  CICE_HIP(hipEventRecord(e1, c_->stream));
  {  // same bytes with 16-byte lanes, for comparison in the kernel trace only
    const size_t n2 = (size_t)n_doubles / 2;
    const dim3 g2((unsigned)((n2 + 255) / 256));
    for (int r = 0; r < 2; ++r)
      hipLaunchKernelGGL(k_diag_copy16, g2, dim3(256), 0, c_->stream, (const double2*)a.p, (double2*)b.p, n2);
  }
  CICE_HIP(hipEventSynchronize(e1));
  if (elapsed_ms) CICE_HIP(hipEventElapsedTime(elapsed_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  CICE_CATCH
}

int cice_device_sync(cice_ctx* ctx) {
  CICE_TRY(ctx)
  c_->need_device();
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

// ---- domain ---------------------------------------------------------------------------------
int cice_domain_create(cice_ctx* ctx, int nxg, int nyg, int bsx, int bsy, int ew, int ns, int rank,
                       int npx, int npy) {
  CICE_TRY(ctx)
  CICE_REQUIRE(ew >= 0 && ew <= 2 && ns >= 0 && ns <= 4,
               "boundary type must be 0 (open), 1 (cyclic), 2 (closed) or, north-south only, 3 (tripole) or 4 (tripoleT)");
  c_->dom.self_comm = std::getenv("CICE4_AMD_SELF_COMM") != nullptr;  // test aid, see domain.h
  const char* msg = c_->dom.create(nxg, nyg, bsx, bsy, ew, ns, rank, npx, npy);
  if (msg[0]) throw Error{CICE_EINVAL, std::string("cice_domain_create: ") + msg};
  c_->have_domain = true;
  c_->evp.reset();
  c_->transport.reset();
  c_->chain_on = c_->chain_ready = false;
  c_->upwind.reset();
  c_->halo.reset();
  c_->frame_halo.reset();
  CICE_CATCH
}

int cice_domain_create_map(cice_ctx* ctx, int nxg, int nyg, int bsx, int bsy, int ew, int ns, int rank,
                           int nranks, const int* owner, const int* local_id) {
  CICE_TRY(ctx)
  CICE_REQUIRE(ew >= 0 && ew <= 2 && ns >= 0 && ns <= 4,
               "boundary type must be 0 (open), 1 (cyclic), 2 (closed) or, north-south only, 3 (tripole) or 4 (tripoleT)");
  c_->dom.self_comm = std::getenv("CICE4_AMD_SELF_COMM") != nullptr;
  const char* msg = c_->dom.create_map(nxg, nyg, bsx, bsy, ew, ns, rank, nranks, owner, local_id);
  if (msg[0]) throw Error{CICE_EINVAL, std::string("cice_domain_create_map: ") + msg};
  c_->have_domain = true;
  c_->evp.reset();
  c_->transport.reset();
  c_->chain_on = c_->chain_ready = false;
  c_->upwind.reset();
  c_->halo.reset();
  c_->frame_halo.reset();
  CICE_CATCH
}

// host copies of the index lists of the current domain (tests, external tools): *n entries; out may be NULL
int cice_domain_list(const cice_ctx* ctx, const char* name, int loc, int* n, int32_t* out) {
  if (!ctx || !ctx->have_domain || !name || !n) return CICE_EINVAL;
  const Domain& d = ctx->dom;
  const std::vector<int32_t>* v = nullptr;
  const std::string k(name);
  const int l = loc - 1;
  const bool lok = l >= 0 && l < 4;
  if (k == "hfill") v = &d.hfill;
  else if (k == "fold_lsrc") v = &d.fold_lsrc;
  else if (k == "fold_bidx") v = &d.fold_bidx;
  else if (k == "fold_dst" && lok) v = &d.fold_out[l].dst;
  else if (k == "fold_src" && lok) v = &d.fold_out[l].src;
  else if (k == "fold_lo" && lok) v = &d.fold_lo[l];
  else if (k == "fold_hi" && lok) v = &d.fold_hi[l];
  if (!v) return CICE_EINVAL;
  *n = (int)v->size();
  if (out && !v->empty()) std::memcpy(out, v->data(), v->size() * 4);
  return CICE_OK;
}

int cice_domain_create_slabs(cice_ctx* ctx, int nxg, int nyg, int nblocks_y, int ew, int ns, int rank,
                             int nranks, int overlap) {
  CICE_TRY(ctx)
  CICE_REQUIRE(ew >= 0 && ew <= 2 && ns >= 0 && ns <= 4,
               "boundary type must be 0 (open), 1 (cyclic), 2 (closed) or, north-south only, 3 (tripole) / 4 (tripoleT)");
  c_->dom.self_comm = std::getenv("CICE4_AMD_SELF_COMM") != nullptr;
  const char* msg = c_->dom.create_slabs(nxg, nyg, nblocks_y, ew, ns, rank, nranks, overlap);
  if (msg[0]) throw Error{CICE_EINVAL, std::string("cice_domain_create_slabs: ") + msg};
  c_->have_domain = true;
  c_->evp.reset();
  c_->transport.reset();
  c_->chain_on = c_->chain_ready = false;
  c_->upwind.reset();
  c_->halo.reset();
  c_->frame_halo.reset();
  CICE_CATCH
}

int cice_domain_halo_refresh(const cice_ctx* ctx, int* n, int32_t* src, int32_t* dst) {
  if (!ctx || !ctx->have_domain || !n) return CICE_EINVAL;
  const Domain& d = ctx->dom;
  *n = (int)d.rsrc.size();
  if (src && dst && !d.rsrc.empty()) {
    std::memcpy(src, d.rsrc.data(), d.rsrc.size() * 4);
    std::memcpy(dst, d.rdst.data(), d.rdst.size() * 4);
  }
  return CICE_OK;
}

int cice_domain_info(const cice_ctx* ctx, int info[9]) {
  if (!ctx || !info || !ctx->have_domain) return CICE_EINVAL;
  const Domain& d = ctx->dom;
  int ns = 0, nr = 0;
  for (const HaloMsg& m : d.send) ns += (int)m.addr.size();
  for (const HaloMsg& m : d.recv) nr += (int)m.addr.size();
  const int v[9] = {d.nx_block, d.ny_block, d.nblocks(), (int)d.all.size(), (int)d.hsrc.size(),
                    (int)d.send.size(), (int)d.recv.size(), ns, nr};
  std::memcpy(info, v, sizeof(v));
  return CICE_OK;
}

int cice_domain_block(const cice_ctx* ctx, int lb, int info[10]) {
  if (!ctx || !info || !ctx->have_domain || lb < 0 || lb >= ctx->dom.nblocks()) return CICE_EINVAL;
  const Block& b = ctx->dom.all[ctx->dom.local[lb]];
  const int v[10] = {b.ilo, b.ihi, b.jlo, b.jhi, b.i0, b.j0, b.gid, b.owner, b.own_jlo, b.own_jhi};
  std::memcpy(info, v, sizeof(v));
  return CICE_OK;
}

int cice_domain_halo_local(const cice_ctx* ctx, int32_t* src, int32_t* dst) {
  if (!ctx || !ctx->have_domain || !src || !dst) return CICE_EINVAL;
  const Domain& d = ctx->dom;
  if (!d.hsrc.empty()) {
    std::memcpy(src, d.hsrc.data(), d.hsrc.size() * 4);
    std::memcpy(dst, d.hdst.data(), d.hdst.size() * 4);
  }
  return CICE_OK;
}

int cice_domain_halo_msg(const cice_ctx* ctx, int dir, int msg, int* peer, int* count, int32_t* addr) {
  if (!ctx || !ctx->have_domain) return CICE_EINVAL;
  // dir 0 / 1: ghost-cell messages (send / receive); 2 / 3: tripole top rows into the global buffer
  // (send: local addresses, receive: buffer indices)
  if (dir < 0 || dir > 3) return CICE_EINVAL;
  const std::vector<HaloMsg>& v = dir == 0 ? ctx->dom.send : dir == 1 ? ctx->dom.recv
                                  : dir == 2 ? ctx->dom.fold_send : ctx->dom.fold_recv;
  if (msg < 0 || msg >= (int)v.size()) return CICE_EINVAL;
  if (peer) *peer = v[msg].peer;
  if (count) *count = (int)v[msg].addr.size();
  if (addr) std::memcpy(addr, v[msg].addr.data(), v[msg].addr.size() * 4);
  return CICE_OK;
}

// ---- communication ---------------------------------------------------------------------------
// CICE4_AMD_SKIP_COMM (test aid): no RCCL communicator is created -- for checks of the block topology
// of a multi-rank run on hosts without one GPU per rank; any later exchange then fails loudly.
static bool skip_comm() { return std::getenv("CICE4_AMD_SKIP_COMM") != nullptr; }

int cice_comm_unique_id(char uid[128]) {
  if (!uid) return CICE_EINVAL;
  if (skip_comm()) {
    std::memset(uid, 0, 128);
    return CICE_OK;
  }
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return CICE_ECOMM;
  std::memcpy(uid, &id, 128);
  return CICE_OK;
}

int cice_comm_init(cice_ctx* ctx, const char uid[128], int rank, int nranks) {
  CICE_TRY(ctx)
  if (skip_comm()) return CICE_OK;
  CICE_REQUIRE(uid != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "cice_comm_init: bad arguments");
  c_->need_halo();
  if (!c_->comm) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    std::memcpy(&id, uid, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c_->comm, nranks, id, rank);
    if (r != ncclSuccess) {
      c_->comm = nullptr;
      throw Error{CICE_ECOMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)};
    }
    c_->comm_rank = rank;
    c_->comm_nranks = nranks;
  } else {
    CICE_REQUIRE(rank == c_->comm_rank && nranks == c_->comm_nranks,
                 "cice_comm_init: this context already has a communicator with another rank / size");
  }
  c_->halo->set_comm((ncclComm*)c_->comm, c_->comm_rank, c_->comm_nranks);
  if (c_->frame_halo) c_->frame_halo->set_comm((ncclComm*)c_->comm, c_->comm_rank, c_->comm_nranks);
  CICE_CATCH
}

// In-process link instead of an RCCL communicator (halo.h): the ranks are contexts of this process, one host thread each.
int cice_comm_init_local(cice_ctx* ctx, int link_id, int rank, int nranks) {
  CICE_TRY(ctx)
  CICE_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "cice_comm_init_local: bad arguments");
  CICE_REQUIRE(!c_->comm, "cice_comm_init_local: this context already has an RCCL communicator");
  c_->need_halo();
  c_->link = local_link_get(link_id, nranks);
  c_->comm_rank = rank;
  c_->comm_nranks = nranks;
  c_->halo->set_link(c_->link, rank, nranks);
  if (c_->frame_halo) c_->frame_halo->set_link(c_->link, rank, nranks);
  CICE_CATCH
}

// The same between processes of one host (a file under /dev/shm; box_bytes = the largest message).
int cice_comm_init_shm(cice_ctx* ctx, const char* name, int rank, int nranks, long long box_bytes) {
  CICE_TRY(ctx)
  CICE_REQUIRE(name && name[0] == '/' && nranks >= 1 && rank >= 0 && rank < nranks && box_bytes > 0,
               "cice_comm_init_shm: bad arguments (the name must start with '/')");
  CICE_REQUIRE(!c_->comm && !c_->link, "cice_comm_init_shm: this context already has a communicator");
  c_->need_halo();
  c_->link = shm_link_open(name, rank, nranks, (size_t)box_bytes);
  c_->link_owned = true;
  c_->comm_rank = rank;
  c_->comm_nranks = nranks;
  c_->halo->set_link(c_->link, rank, nranks);
  if (c_->frame_halo) c_->frame_halo->set_link(c_->link, rank, nranks);
  CICE_CATCH
}

// TIMING AID (halo.hip: MirrorLink): this context is rank `rank` of `nranks`, alone; its messages come back to it.
int cice_comm_init_mirror(cice_ctx* ctx, int rank, int nranks) {
  CICE_TRY(ctx)
  CICE_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "cice_comm_init_mirror: bad arguments");
  CICE_REQUIRE(!c_->comm && !c_->link, "cice_comm_init_mirror: this context already has a communicator");
  c_->need_halo();
  c_->link = mirror_link_new(nranks);
  c_->link_owned = true;
  c_->comm_rank = rank;
  c_->comm_nranks = nranks;
  c_->halo->set_link(c_->link, rank, nranks);
  if (c_->frame_halo) c_->frame_halo->set_link(c_->link, rank, nranks);
  CICE_CATCH
}

// Ranks of this context's communicator as RCCL itself counts them (ncclCommCount); 0 before cice_comm_init.
int cice_comm_count(cice_ctx* ctx, int* nranks) {
  CICE_TRY(ctx)
  CICE_REQUIRE(nranks != nullptr, "NULL argument");
  *nranks = 0;
  if (c_->link) *nranks = c_->comm_nranks;
  if (c_->comm) {
    const ncclResult_t r = ncclCommCount(c_->comm, nranks);
    if (r != ncclSuccess) throw Error{CICE_ECOMM, std::string("ncclCommCount: ") + ncclGetErrorString(r)};
  }
  CICE_CATCH
}

// ---- EVP -------------------------------------------------------------------------------------
int cice_evp_init(cice_ctx* ctx, const cice_evp_config* cfg, const cice_evp_grid* grid) {
  CICE_TRY(ctx)
  CICE_REQUIRE(cfg && grid, "NULL argument");
  c_->need_halo();
  c_->evp.reset(new Evp(c_->dom, *c_->halo, c_->stream, c_->fan));
  c_->evp->init(*cfg, *grid);
  CICE_CATCH
}

#define NEED_EVP CICE_REQUIRE(c_->evp != nullptr, "cice_evp_init has not been called")

int cice_evp_upload(cice_ctx* ctx, const cice_evp_fields* f) {
  CICE_TRY(ctx) c_->chain_ready = false; NEED_EVP; CICE_REQUIRE(f, "NULL argument"); c_->evp->upload(*f); CICE_CATCH
}
int cice_evp_download(cice_ctx* ctx, cice_evp_fields* f) {
  CICE_TRY(ctx) NEED_EVP; CICE_REQUIRE(f, "NULL argument"); c_->evp->download(*f); CICE_CATCH
}
int cice_evp_step(cice_ctx* ctx, double dt) { CICE_TRY(ctx) c_->chain_ready = false; NEED_EVP; c_->evp->forget_host_state(); c_->evp->step(dt); CICE_CATCH }
int cice_evp(cice_ctx* ctx, double dt, cice_evp_fields* f) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(f, "NULL argument");
  c_->chain_ready = false;
  const bool chain = c_->chain_on && c_->transport;
  c_->evp->run(dt, *f, [&]() {
    if (!chain) return;
    // the rest of the transport's state travels while the subcycle loop runs (the link idles then); see cice_transport_chain
    c_->transport->prefetch(c_->chain);
    c_->chain_aicen = f->aicen; c_->chain_vicen = f->vicen; c_->chain_u = f->uvel; c_->chain_v = f->vvel;
  });
  c_->chain_ready = chain;   // only a call that got this far leaves device copies the transport may take over
  CICE_CATCH
}
// f1 hand-off: the state the batched thermodynamic step left on the device becomes the dynamics' input without crossing
// PCIe (valid when nothing on the host has changed aicen / vicen / vsnon since: the caller's statement).
int cice_evp_adopt_thermo_state(cice_ctx* ctx) {
  CICE_TRY(ctx) c_->chain_ready = false;
  NEED_EVP;
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0, "cice_thermo_batch_alloc has not been called");
  CICE_REQUIRE(t.nx == c_->dom.nx_block && t.ny == c_->dom.ny_block && t.nb == c_->dom.nblocks(),
               "cice_evp_adopt_thermo_state: the thermodynamic batch has another block layout than the dynamics");
  c_->evp->adopt_state(t.aicen.p, t.vicen.p, t.vsnon.p);
  CICE_CATCH
}
int cice_evp_pin_fields(cice_ctx* ctx, const cice_evp_fields* f) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(f, "NULL argument");
  const size_t n = (size_t)c_->dom.nblocks() * c_->dom.nx_block * c_->dom.ny_block;
  auto pin = [&](const void* h, size_t bytes) { c_->pin_range(h, bytes); };
  const double* r8[] = {f->aice, f->vice, f->vsno, f->aice0, f->strairxT, f->strairyT, f->uocn, f->vocn,
                        f->ss_tltx, f->ss_tlty, f->uvel, f->vvel, f->stressp_1, f->stressp_2, f->stressp_3,
                        f->stressp_4, f->stressm_1, f->stressm_2, f->stressm_3, f->stressm_4, f->stress12_1,
                        f->stress12_2, f->stress12_3, f->stress12_4, f->fm, f->strtltx, f->strtlty, f->strocnx,
                        f->strocny, f->strintx, f->strinty, f->strairx, f->strairy, f->strength, f->divu,
                        f->shear, f->rdg_conv, f->rdg_shear, f->prs_sig, f->strocnxT, f->strocnyT};
  for (const double* h : r8) pin(h, n * 8);
  pin(f->aicen, n * NCAT * 8);
  pin(f->vicen, n * NCAT * 8);
  pin(f->iceumask, n * 4);
  CICE_CATCH
}
int cice_evp_prepare(cice_ctx* ctx, double dt) { CICE_TRY(ctx) c_->chain_ready = false; NEED_EVP; c_->evp->forget_host_state(); c_->evp->prepare(dt); CICE_CATCH }
int cice_evp_subcycles(cice_ctx* ctx, int ksub0, int nsub, float* ms) {
  CICE_TRY(ctx) c_->chain_ready = false; NEED_EVP; c_->evp->forget_host_state(); c_->evp->subcycles(ksub0, nsub, ms); CICE_CATCH
}
int cice_evp_finish(cice_ctx* ctx) { CICE_TRY(ctx) c_->chain_ready = false; NEED_EVP; c_->evp->forget_host_state(); c_->evp->finish(); CICE_CATCH }
int cice_evp_download_stresses(cice_ctx* ctx, cice_evp_fields* f) {
  CICE_TRY(ctx) NEED_EVP; CICE_REQUIRE(f, "NULL argument"); c_->evp->download_stresses(*f); CICE_CATCH
}
int cice_evp_set_option(cice_ctx* ctx, const char* key, int value) {
  CICE_TRY(ctx) NEED_EVP; CICE_REQUIRE(key, "NULL key"); c_->evp->set_option(key, value); CICE_CATCH
}
int cice_evp_get_info(cice_ctx* ctx, const char* key, int* value) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(key && value, "NULL argument");
  if (!std::strcmp(key, "derive_metrics")) *value = c_->evp->derives_metrics() ? 1 : 0;
  else if (!std::strcmp(key, "waves")) *value = c_->evp->tile_waves();
  else if (!std::strcmp(key, "rows_per_wave")) *value = c_->evp->tile_rows();
  else if (!std::strcmp(key, "fused")) *value = c_->evp->can_fuse() ? 1 : 0;
  else if (!std::strcmp(key, "fused_waves")) *value = c_->evp->fused_waves();
  else if (!std::strcmp(key, "skew")) *value = c_->evp->can_skew() || c_->evp->can_skew_fold() ? 1 : 0;
  else if (!std::strcmp(key, "skew_fold")) *value = !c_->evp->can_skew() && c_->evp->can_skew_fold() ? 1 : 0;
  else if (!std::strcmp(key, "skew_levels")) *value = c_->evp->skew_levels();
  else if (!std::strcmp(key, "skew_subs")) *value = c_->evp->skew_subs(c_->evp->skew_levels());
  else if (!std::strcmp(key, "skew_pairs")) *value = c_->evp->pairs_ok() ? 1 : 0;
  else if (!std::strcmp(key, "skew_fill")) *value = c_->evp->skew_rows_on() ? c_->evp->skew_fill_pct() : 0;
  else if (!std::strcmp(key, "resident_map")) *value = c_->evp->resident_map();
  else if (!std::strcmp(key, "skew_rowact")) *value = c_->evp->rowact_on() ? 1 : 0;
  else if (!std::strcmp(key, "skew_balance")) *value = c_->evp->skew_rows_on() && c_->evp->balance_on() ? 1 : 0;
  else if (!std::strcmp(key, "skew_balanced")) *value = (int)std::min<long long>(c_->evp->balanced_sweeps(), 2000000000LL);
  else if (!std::strcmp(key, "skew_trim_ext")) *value = (c_->evp->can_skew() && c_->evp->can_trim()) ? 1 : 0;
  else if (!std::strcmp(key, "skew_split")) *value = (c_->evp->can_skew() && c_->evp->can_split()) ? 1 : 0;
  else if (!std::strcmp(key, "skew_strips")) *value = c_->evp->skew_strips(c_->evp->skew_levels(), nullptr);
  else if (!std::strcmp(key, "skew_seg_rows")) *value = c_->evp->skew_seg_rows(c_->evp->skew_levels());
  else if (!std::strcmp(key, "resident")) *value = (c_->evp->can_reside() || c_->evp->can_reside_peer()) ? 1 : 0;
  else if (!std::strcmp(key, "resident_peer")) *value = c_->evp->can_reside_peer() ? 1 : 0;
  else if (!std::strcmp(key, "last_launches")) *value = c_->evp->last_launches;
  else if (!std::strcmp(key, "resident_peer_fine")) *value = c_->evp->peer_buffers_fine() ? 1 : 0;
#ifdef CICE4_AMD_EXPERIMENTS
  else if (!std::strcmp(key, "experiments")) *value = 1;
#else
  else if (!std::strcmp(key, "experiments")) *value = 0;     // (the variants measured slower are not in this build: evp.hip)
#endif
  else if (!std::strcmp(key, "resident_granules")) *value = c_->evp->granules_in_use() ? 1 : 0;
  else if (!std::strcmp(key, "resident_waves")) *value = c_->evp->resident_waves();
  else if (!std::strcmp(key, "resident_dense")) *value = c_->evp->can_reside() && c_->evp->resident_dense() ? 1 : 0;
  else throw Error{CICE_EINVAL, std::string("unknown info key ") + key};
  CICE_CATCH
}
int cice_evp_peer_export(cice_ctx* ctx, void* bufs[3], long long* plane) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(bufs && plane, "NULL argument");
  c_->evp->peer_export(bufs);
  *plane = (long long)c_->dom.nblocks() * c_->dom.nx_block * c_->dom.ny_block;
  CICE_CATCH
}
int cice_evp_peer_connect(cice_ctx* ctx, int side, void* xu0, void* xu1, void* rprog, long long plane) {
  CICE_TRY(ctx) NEED_EVP; c_->evp->peer_connect(side, xu0, xu1, rprog, plane); CICE_CATCH
}
// Any cartesian layout with one block per rank (round 5): the ranks this rank's block exchanges ghost cells with, and the
// connection of one of them by its rank.
int cice_evp_peer_ranks(cice_ctx* ctx, int* n, int32_t ranks[8]) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(n && ranks, "NULL argument");
  const std::vector<int> v = c_->evp->peer_ranks();
  CICE_REQUIRE(v.size() <= 8, "cice_evp_peer_ranks: more than eight neighbouring ranks");
  *n = (int)v.size();
  for (size_t k = 0; k < v.size(); ++k) ranks[k] = v[k];
  CICE_CATCH
}
int cice_evp_peer_connect_rank(cice_ctx* ctx, int rank, void* xu0, void* xu1, void* rprog, long long plane) {
  CICE_TRY(ctx) NEED_EVP; c_->evp->peer_connect_rank(rank, xu0, xu1, rprog, plane); CICE_CATCH
}
int cice_evp_peer_connect_rank_ipc(cice_ctx* ctx, int rank, const char handles[3][64], long long plane) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(handles, "NULL argument");
  void* p[3];
  for (int k = 0; k < 3; ++k) {
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles[k], 64);
    CICE_HIP(hipIpcOpenMemHandle(&p[k], h, hipIpcMemLazyEnablePeerAccess));
  }
  c_->evp->peer_connect_rank(rank, p[0], p[1], p[2], plane);
  CICE_CATCH
}
// The same buffers as IPC handles (3 x 64 bytes) for a neighbour in ANOTHER process, and their opening on the other
// side.  (Across processes / GPUs; not exercised on the one-GPU test boxes, where two contexts of one process exchange
// plain pointers.)
int cice_evp_peer_export_ipc(cice_ctx* ctx, char handles[3][64], long long* plane) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(handles && plane, "NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
  void* bufs[3];
  c_->evp->peer_export(bufs);
  for (int k = 0; k < 3; ++k) CICE_HIP(hipIpcGetMemHandle((hipIpcMemHandle_t*)handles[k], bufs[k]));
  *plane = (long long)c_->dom.nblocks() * c_->dom.nx_block * c_->dom.ny_block;
  CICE_CATCH
}
int cice_evp_peer_connect_ipc(cice_ctx* ctx, int side, const char handles[3][64], long long plane) {
  CICE_TRY(ctx)
  NEED_EVP;
  CICE_REQUIRE(handles, "NULL argument");
  void* p[3];
  for (int k = 0; k < 3; ++k) {
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles[k], 64);
    CICE_HIP(hipIpcOpenMemHandle(&p[k], h, hipIpcMemLazyEnablePeerAccess));
  }
  c_->evp->peer_connect(side, p[0], p[1], p[2], plane);
  CICE_CATCH
}
int cice_evp_debug(cice_ctx* ctx, const char* what, long long* out, long long* count) {
  CICE_TRY(ctx)
  CICE_REQUIRE(what && count, "NULL argument");
  if (!std::strcmp(what, "thermo_niter")) {   // one byte per (cell, category) of the batched thermo state, packed in the words
    const long long nbytes = (long long)c_->tb.niter.n, nw = (nbytes + 7) / 8;
    if (out && nbytes) {
      CICE_REQUIRE(*count >= nw, "cice_evp_debug: buffer too small");
      CICE_HIP(hipStreamSynchronize(c_->stream));
      CICE_HIP(hipMemcpy(out, c_->tb.niter.p, (size_t)nbytes, hipMemcpyDeviceToHost));
    }
    *count = nw;
    return CICE_OK;
  }
  if (!std::strcmp(what, "thermo_perm")) {    // the permutation of the last sorted thermo step, two int32 per word
    const long long nbytes = (long long)c_->tb.perm.n * 4, nw = (nbytes + 7) / 8;
    if (out && nbytes) {
      CICE_REQUIRE(*count >= nw, "cice_evp_debug: buffer too small");
      CICE_HIP(hipStreamSynchronize(c_->stream));
      CICE_HIP(hipMemcpy(out, c_->tb.perm.p, (size_t)nbytes, hipMemcpyDeviceToHost));
    }
    *count = nw;
    return CICE_OK;
  }
  NEED_EVP;
  *count = c_->evp->debug_read(what, out, *count);
  CICE_CATCH
}
int cice_evp_active_cells(cice_ctx* ctx, long long* nt, long long* nu) {
  CICE_TRY(ctx) NEED_EVP; c_->evp->active_cells(nt, nu); CICE_CATCH
}

int cice_evp_stress(cice_ctx* ctx, double dt, int ndte, int damping, int nx, int ny, int ksub,
                    int icellt, const int32_t* ti, const int32_t* tj, const double* uvel,
                    const double* vvel, const double* dxt, const double* dyt, const double* dxhy,
                    const double* dyhx, const double* cxp, const double* cyp, const double* cxm,
                    const double* cym, const double* tarear, const double* tinyarea,
                    const double* strength, double* sp1, double* sp2, double* sp3, double* sp4,
                    double* sm1, double* sm2, double* sm3, double* sm4, double* s121, double* s122,
                    double* s123, double* s124, double* shear, double* divu, double* prs_sig,
                    double* rdg_conv, double* rdg_shear, double* str) {
  CICE_TRY(ctx)
  c_->need_device();
  const double* g10[10] = {dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarear, tinyarea};
  double* sg[12] = {sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124};
  double* dg[5] = {shear, divu, prs_sig, rdg_conv, rdg_shear};
  CICE_REQUIRE(nx >= 3 && ny >= 3 && ndte >= 1, "bad dimensions");
  Evp::stress_host(c_->stream, dt, ndte, damping, nx, ny, ksub, icellt, ti, tj, uvel, vvel, g10,
                   strength, sg, dg, str);
  CICE_CATCH
}

int cice_evp_stepu(cice_ctx* ctx, int nx, int ny, int icellu, const int32_t* ui, const int32_t* uj,
                   const double* aiu, const double* str, const double* uocn, const double* vocn,
                   const double* waterx, const double* watery, const double* forcex,
                   const double* forcey, const double* umassdtei, const double* fm,
                   const double* uarear, double* strocnx, double* strocny, double* strintx,
                   double* strinty, double* uvel, double* vvel) {
  CICE_TRY(ctx)
  c_->need_device();
  const double* in10[10] = {aiu, uocn, vocn, waterx, watery, forcex, forcey, umassdtei, fm, uarear};
  double* io6[6] = {strocnx, strocny, strintx, strinty, uvel, vvel};
  CICE_REQUIRE(nx >= 3 && ny >= 3, "bad dimensions");
  Evp::stepu_host(c_->stream, nx, ny, icellu, ui, uj, in10, str, io6);
  CICE_CATCH
}

int cice_halo_update_r8(cice_ctx* ctx, double* field, int nlev) {
  CICE_TRY(ctx) halo_host<double>(c_, field, nlev); CICE_CATCH
}
int cice_halo_update_i4(cice_ctx* ctx, int32_t* field, int nlev) {
  CICE_TRY(ctx) halo_host<int32_t>(c_, field, nlev); CICE_CATCH
}
// the same with the field location / kind (FieldLoc, FieldKind codes of ice_constants.F90:185-205: they decide
// offsets and sign at a tripole fold) and the fill value for ghost cells facing eliminated land blocks
int cice_halo_update_ex_r8(cice_ctx* ctx, double* field, int nlev, int loc, int kind, double fill) {
  CICE_TRY(ctx) halo_host<double>(c_, field, nlev, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_ex_r4(cice_ctx* ctx, float* field, int nlev, int loc, int kind, float fill) {
  CICE_TRY(ctx) halo_host<float>(c_, field, nlev, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_ex_i4(cice_ctx* ctx, int32_t* field, int nlev, int loc, int kind, int32_t fill) {
  CICE_TRY(ctx) halo_host<int32_t>(c_, field, nlev, loc, kind, fill); CICE_CATCH
}
// host field in the reference's (nx_block, ny_block, nz, nblocks) layout (nz = product of the level dimensions)
int cice_halo_update_blocked_r8(cice_ctx* ctx, double* field, int nz, int loc, int kind, double fill) {
  CICE_TRY(ctx) halo_host_blocked<double>(c_, field, nz, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_blocked_r4(cice_ctx* ctx, float* field, int nz, int loc, int kind, float fill) {
  CICE_TRY(ctx) halo_host_blocked<float>(c_, field, nz, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_blocked_i4(cice_ctx* ctx, int32_t* field, int nz, int loc, int kind, int32_t fill) {
  CICE_TRY(ctx) halo_host_blocked<int32_t>(c_, field, nz, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_strided_r8(cice_ctx* ctx, double* field, int nz1, long long stride1, int nz2, long long stride2,
                                long long stride_block, int loc, int kind, double fill) {
  CICE_TRY(ctx)
  CICE_REQUIRE(stride1 >= 0 && stride2 >= 0 && stride_block >= 0, "negative stride");
  halo_host_strided<double>(c_, field, LevelStrides{nz1, nz2, (size_t)stride1, (size_t)stride2, (size_t)stride_block}, loc,
                            kind, fill);
  CICE_CATCH
}
int cice_halo_update_dev_ex_r8(cice_ctx* ctx, double* dev_field, int nlev, int loc, int kind, double fill) {
  CICE_TRY(ctx) halo_dev<double>(c_, dev_field, nlev, loc, kind, fill); CICE_CATCH
}
int cice_halo_update_dev_r8(cice_ctx* ctx, double* dev_field, int nlev) {
  CICE_TRY(ctx) halo_dev<double>(c_, dev_field, nlev); CICE_CATCH
}
int cice_halo_update_dev_i4(cice_ctx* ctx, int32_t* dev_field, int nlev) {
  CICE_TRY(ctx) halo_dev<int32_t>(c_, dev_field, nlev); CICE_CATCH
}
// device memory for callers that keep fields resident (cice_halo_update_dev_*): plain hipMalloc/hipFree on
// the context's device plus explicit copies ordered on the library's stream
int cice_device_alloc(cice_ctx* ctx, size_t bytes, void** dev) {
  CICE_TRY(ctx)
  CICE_REQUIRE(dev != nullptr, "NULL argument");
  c_->need_device();
  CICE_HIP(hipMalloc(dev, bytes));
  CICE_CATCH
}
int cice_device_free(cice_ctx* ctx, void* dev) {
  CICE_TRY(ctx)
  c_->need_device();
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_HIP(hipFree(dev));
  CICE_CATCH
}
int cice_device_copy(cice_ctx* ctx, void* dst, const void* src, size_t bytes, int to_device) {
  CICE_TRY(ctx)
  CICE_REQUIRE(dst && src, "NULL argument");
  c_->need_device();
  CICE_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, c_->stream));
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

// ---- thermodynamics --------------------------------------------------------------------------
int cice_thermo_init(cice_ctx* ctx, const cice_thermo_config* cfg, double* salin, double* Tmlt) {
  CICE_TRY(ctx)
  CICE_REQUIRE(cfg, "NULL argument");
  CICE_REQUIRE(cfg->conduct == 0 || cfg->conduct == 1, "conduct must be 0 (MU71) or 1 (bubbly)");
  CICE_REQUIRE(cfg->nt_Tsfc >= 1 && cfg->nt_Tsfc <= NTRCR, "nt_Tsfc out of range");
  if (!cfg->heat_capacity)
    throw Error{CICE_EUNSUPPORTED, "zero-layer thermodynamics (heat_capacity = F) is not implemented on the device"};
  c_->tp.init(*cfg);
  c_->have_thermo = true;
  if (salin) std::memcpy(salin, c_->tp.salin, sizeof(c_->tp.salin));
  if (Tmlt) std::memcpy(Tmlt, c_->tp.Tmlt, sizeof(c_->tp.Tmlt));
  CICE_CATCH
}

static void decode_err(unsigned long long key, int nx, int ncat, const int32_t* indxi,
                       const int32_t* indxj, int32_t* l_stop, int32_t* istop, int32_t* jstop,
                       int32_t* nstop, int32_t* bstop) {
  *l_stop = 0; *istop = 0; *jstop = 0;
  if (nstop) *nstop = 0;
  if (bstop) *bstop = 0;
  if (key == ~0ull) return;
  *l_stop = 1;
  const unsigned long long order = key & ((1ull << 40) - 1);
  const unsigned long long cb = key >> 44;
  if (indxi) {
    *istop = indxi[order];
    *jstop = indxj[order];
  } else {
    *jstop = (int32_t)(order / nx) + 1;
    *istop = (int32_t)(order % nx) + 1;
  }
  if (nstop) *nstop = (int32_t)(cb % ncat) + 1;
  if (bstop) *bstop = (int32_t)(cb / ncat) + 1;
}

int cice_thermo_vertical(cice_ctx* ctx, int nx, int ny, double dt, int icells, const int32_t* indxi,
                         const int32_t* indxj, double* aicen, double* trcrn, double* vicen,
                         double* vsnon, double* eicen, double* esnon, const double* flw,
                         const double* potT, const double* Qa, const double* rhoa,
                         const double* fsnow, const double* fbot, const double* Tbot,
                         const double* lhcoef, const double* shcoef, double* fswsfc, double* fswint,
                         double* fswthrun, double* Sswabs, double* Iswabs, double* fsurfn,
                         double* fcondtopn, double* fsensn, double* flatn, double* fswabsn,
                         double* flwoutn, double* evapn, double* freshn, double* fsaltn,
                         double* fhocnn, double* meltt, double* melts, double* meltb, double* congel,
                         double* snoice, double* mlt_onset, double* frz_onset, double yday,
                         int32_t* l_stop, int32_t* istop, int32_t* jstop) {
  CICE_TRY(ctx) c_->chain_ready = false;
  CICE_REQUIRE(c_->have_thermo, "cice_thermo_init has not been called");
  CICE_REQUIRE(l_stop && istop && jstop, "NULL status pointer");
  CICE_REQUIRE(nx >= 1 && ny >= 1, "bad dimensions");
  const size_t np = (size_t)nx * ny;
  CICE_REQUIRE(icells >= 0 && (size_t)icells <= np, "icells out of range");
  CICE_REQUIRE(icells == 0 || (indxi && indxj), "thermo_vertical: NULL index list");
  {  // every array is checked before anything is queued on the stream
    const void* all[] = {aicen, trcrn, vicen, vsnon, eicen, esnon, flw, potT, Qa, rhoa, fsnow, fbot, Tbot,
                         lhcoef, shcoef, fswsfc, fswint, fswthrun, Sswabs, Iswabs, fsurfn, fcondtopn, fsensn,
                         flatn, fswabsn, flwoutn, evapn, freshn, fsaltn, fhocnn, meltt, melts, meltb, congel,
                         snoice, mlt_onset, frz_onset};
    for (const void* q : all) CICE_REQUIRE(q != nullptr, "thermo_vertical: NULL array");
  }
  for (int e = 0; e < icells; ++e)
    CICE_REQUIRE(indxi[e] >= 1 && indxi[e] <= nx && indxj[e] >= 1 && indxj[e] <= ny,
                 "thermo_vertical: index outside block");
  c_->need_device();
  hipStream_t s = c_->stream;
  // plane map of the single staging buffer
  enum { A_AICEN = 0, A_TRCRN = 1, A_VICEN = A_TRCRN + NTRCR, A_VSNON, A_EICEN, A_ESNON = A_EICEN + NILYR,
         A_FLW = A_ESNON + NSLYR, A_POTT, A_QA, A_RHOA, A_FSNOW, A_FBOT, A_TBOT, A_LH, A_SH, A_FSWSFC,
         A_FSWINT, A_FSWTHRU, A_SSW, A_ISW = A_SSW + NSLYR, A_OUT = A_ISW + NILYR, A_MLT = A_OUT + 15,
         A_FRZ, A_END };
  DevBuf<double>& d = c_->tv_stage;
  DevBuf<int32_t>& li = c_->tv_list;
  if ((size_t)icells * 2 <= np) {
    // Few of the block's cells carry ice of this category (the rule on a real grid: the reference compresses to a
    // list for that reason): only the listed cells travel.  The host gathers them plane by plane into a page-locked
    // buffer, ONE copy takes all planes to the device, the list kernel runs on that compact "1 x icells block", ONE
    // copy brings everything back, the host zeroes the output planes (:299-329) and scatters the listed cells.
    // 54 copies of whole planes become 2 of icells elements per plane.
    const size_t m = (size_t)icells;
    double* houts[15] = {fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn, evapn, freshn, fsaltn,
                         fhocnn, meltt, melts, meltb, congel, snoice};
    auto zero_out = [&]() {
      for (int k = 0; k < 15; ++k)
        if (c_->tp.calc_Tsfc || !(k == 0 || k == 1 || k == 3)) std::memset(houts[k], 0, np * 8);
    };
    if (m == 0) {
      zero_out();
      *l_stop = 0; *istop = 0; *jstop = 0;
      return CICE_OK;
    }
    const size_t bytes = ((size_t)A_END * m) * 8 + 2 * m * 4;
    if (c_->tv_host_bytes < bytes) {
      if (c_->tv_host) (void)hipHostFree(c_->tv_host);
      c_->tv_host = nullptr;
      c_->tv_host_bytes = 0;
      CICE_HIP(hipHostMalloc(&c_->tv_host, bytes + bytes / 2, hipHostMallocDefault));
      c_->tv_host_bytes = bytes + bytes / 2;
    }
    if (d.n < (size_t)A_END * m) d.alloc((size_t)A_END * std::max(m, np / 8));
    if (li.n < 2 * m) li.alloc(2 * std::max(m, np / 8));
    double* hp = static_cast<double*>(c_->tv_host);
    int32_t* hl = reinterpret_cast<int32_t*>(hp + (size_t)A_END * m);
    std::vector<size_t>& cq = c_->tv_cells;
    cq.resize(m);
    for (size_t e = 0; e < m; ++e) {
      cq[e] = (size_t)(indxj[e] - 1) * nx + (indxi[e] - 1);
      hl[e] = (int32_t)e + 1;      // the compact block is one row of m cells
      hl[m + e] = 1;
    }
    auto gather = [&](int plane, const double* h, int planes = 1) {
      for (int k = 0; k < planes; ++k) {
        double* o = hp + (size_t)(plane + k) * m;
        const double* src = h + (size_t)k * np;
        for (size_t e = 0; e < m; ++e) o[e] = src[cq[e]];
      }
    };
    const int it_T = c_->tp.nt_Tsfc - 1;
    gather(A_AICEN, aicen); gather(A_TRCRN + it_T, trcrn + (size_t)it_T * np); gather(A_VICEN, vicen);
    gather(A_VSNON, vsnon); gather(A_EICEN, eicen, NILYR); gather(A_ESNON, esnon, NSLYR);
    gather(A_FLW, flw); gather(A_POTT, potT); gather(A_QA, Qa); gather(A_RHOA, rhoa); gather(A_FSNOW, fsnow);
    gather(A_FBOT, fbot); gather(A_TBOT, Tbot); gather(A_LH, lhcoef); gather(A_SH, shcoef);
    gather(A_FSWSFC, fswsfc); gather(A_FSWINT, fswint); gather(A_FSWTHRU, fswthrun);
    gather(A_SSW, Sswabs, NSLYR); gather(A_ISW, Iswabs, NILYR);
    gather(A_MLT, mlt_onset); gather(A_FRZ, frz_onset);
    if (!c_->tp.calc_Tsfc) { gather(A_OUT + 0, fsurfn); gather(A_OUT + 1, fcondtopn); gather(A_OUT + 3, flatn); }
    CICE_HIP(hipMemcpyAsync(d.p, hp, (size_t)A_END * m * 8, hipMemcpyHostToDevice, s));
    CICE_HIP(hipMemcpyAsync(li.p, hl, 2 * m * 4, hipMemcpyHostToDevice, s));
    c_->tkey.alloc(THERMO_STATUS_WORDS);
    CICE_HIP(hipMemsetAsync(c_->tkey.p, 0xff, 8, s));
    CICE_HIP(hipMemsetAsync(c_->tkey.p + 1, 0, (THERMO_STATUS_WORDS - 1) * 8, s));
    ThermoArgs a{};
    a.p = c_->tp; a.nx = (int)m; a.ny = 1; a.ncat = 1; a.nblocks = 1; a.dt = dt; a.yday = yday;
    a.icells = icells; a.indxi = li.p; a.indxj = li.p + m; a.blk = nullptr;
    auto P = [&](int plane) { return d.p + (size_t)plane * m; };
    a.aicen = P(A_AICEN); a.trcrn = P(A_TRCRN); a.vicen = P(A_VICEN); a.vsnon = P(A_VSNON);
    a.eicen = P(A_EICEN); a.esnon = P(A_ESNON); a.flw = P(A_FLW); a.potT = P(A_POTT); a.Qa = P(A_QA);
    a.rhoa = P(A_RHOA); a.fsnow = P(A_FSNOW); a.fbot = P(A_FBOT); a.Tbot = P(A_TBOT);
    a.lhcoef = P(A_LH); a.shcoef = P(A_SH); a.fswsfc = P(A_FSWSFC); a.fswint = P(A_FSWINT);
    a.fswthrun = P(A_FSWTHRU); a.Sswabs = P(A_SSW); a.Iswabs = P(A_ISW);
    double** outs[15] = {&a.fsurfn, &a.fcondtopn, &a.fsensn, &a.flatn, &a.fswabsn, &a.flwoutn, &a.evapn,
                         &a.freshn, &a.fsaltn, &a.fhocnn, &a.meltt, &a.melts, &a.meltb, &a.congel,
                         &a.snoice};
    for (int k = 0; k < 15; ++k) *outs[k] = P(A_OUT + k);
    a.mlt_onset = P(A_MLT); a.frz_onset = P(A_FRZ);
    a.errkey = c_->tkey.p; a.nupdates = c_->tkey.p + THERMO_COUNT_STRIDE;
    thermo_launch_list(a, s);
    CICE_HIP(hipMemcpyAsync(hp, d.p, (size_t)A_END * m * 8, hipMemcpyDeviceToHost, s));
    unsigned long long key = 0;
    CICE_HIP(hipMemcpyAsync(&key, c_->tkey.p, 8, hipMemcpyDeviceToHost, s));
    CICE_HIP(hipStreamSynchronize(s));
    zero_out();
    auto scatter = [&](int plane, double* h, int planes = 1) {
      for (int k = 0; k < planes; ++k) {
        const double* in = hp + (size_t)(plane + k) * m;
        double* dst = h + (size_t)k * np;
        for (size_t e = 0; e < m; ++e) dst[cq[e]] = in[e];
      }
    };
    scatter(A_AICEN, aicen); scatter(A_TRCRN + it_T, trcrn + (size_t)it_T * np); scatter(A_VICEN, vicen);
    scatter(A_VSNON, vsnon); scatter(A_EICEN, eicen, NILYR); scatter(A_ESNON, esnon, NSLYR);
    scatter(A_FSWSFC, fswsfc); scatter(A_FSWINT, fswint); scatter(A_SSW, Sswabs, NSLYR); scatter(A_ISW, Iswabs, NILYR);
    for (int k = 0; k < 15; ++k) scatter(A_OUT + k, houts[k]);
    scatter(A_MLT, mlt_onset); scatter(A_FRZ, frz_onset);
    decode_err(key, nx, 1, indxi, indxj, l_stop, istop, jstop, nullptr, nullptr);
    return CICE_OK;
  }
  if (d.n < (size_t)A_END * np) d.alloc((size_t)A_END * np);
  if (li.n < 2 * np) li.alloc(2 * np);
  auto up = [&](int plane, const double* h, int planes = 1) {
    CICE_REQUIRE(h != nullptr, "thermo_vertical: NULL array");
    CICE_HIP(hipMemcpyAsync(d.p + (size_t)plane * np, h, (size_t)planes * np * 8, hipMemcpyHostToDevice, c_->cs()));
  };
  // of the tracers only Tsfc is read and written by the column physics (:137-142, :508-513)
  const int it_T = c_->tp.nt_Tsfc - 1;
  c_->fan.fork(s);   // 22 separate host arrays in, 27 out: spread over the side streams
  up(A_AICEN, aicen); up(A_TRCRN + it_T, trcrn + (size_t)it_T * np); up(A_VICEN, vicen); up(A_VSNON, vsnon);
  up(A_EICEN, eicen, NILYR); up(A_ESNON, esnon, NSLYR);
  up(A_FLW, flw); up(A_POTT, potT); up(A_QA, Qa); up(A_RHOA, rhoa); up(A_FSNOW, fsnow);
  up(A_FBOT, fbot); up(A_TBOT, Tbot); up(A_LH, lhcoef); up(A_SH, shcoef);
  up(A_FSWSFC, fswsfc); up(A_FSWINT, fswint); up(A_FSWTHRU, fswthrun);
  up(A_SSW, Sswabs, NSLYR); up(A_ISW, Iswabs, NILYR);
  up(A_MLT, mlt_onset); up(A_FRZ, frz_onset);
  if (!c_->tp.calc_Tsfc) {  // intent(in) then (ice_therm_vertical.F90:213-217): planes 0, 1, 3 of the outputs
    up(A_OUT + 0, fsurfn); up(A_OUT + 1, fcondtopn); up(A_OUT + 3, flatn);
  }
  if (icells) {
    CICE_HIP(hipMemcpyAsync(li.p, indxi, (size_t)icells * 4, hipMemcpyHostToDevice, c_->cs()));
    CICE_HIP(hipMemcpyAsync(li.p + np, indxj, (size_t)icells * 4, hipMemcpyHostToDevice, c_->cs()));
  }
  c_->fan.join();
  c_->tkey.alloc(THERMO_STATUS_WORDS);
  CICE_HIP(hipMemsetAsync(c_->tkey.p, 0xff, 8, s));
  CICE_HIP(hipMemsetAsync(c_->tkey.p + 1, 0, (THERMO_STATUS_WORDS - 1) * 8, s));
  ThermoArgs a{};
  a.p = c_->tp; a.nx = nx; a.ny = ny; a.ncat = 1; a.nblocks = 1; a.dt = dt; a.yday = yday;
  a.icells = icells; a.indxi = li.p; a.indxj = li.p + np; a.blk = nullptr;
  auto P = [&](int plane) { return d.p + (size_t)plane * np; };
  a.aicen = P(A_AICEN); a.trcrn = P(A_TRCRN); a.vicen = P(A_VICEN); a.vsnon = P(A_VSNON);
  a.eicen = P(A_EICEN); a.esnon = P(A_ESNON); a.flw = P(A_FLW); a.potT = P(A_POTT); a.Qa = P(A_QA);
  a.rhoa = P(A_RHOA); a.fsnow = P(A_FSNOW); a.fbot = P(A_FBOT); a.Tbot = P(A_TBOT);
  a.lhcoef = P(A_LH); a.shcoef = P(A_SH); a.fswsfc = P(A_FSWSFC); a.fswint = P(A_FSWINT);
  a.fswthrun = P(A_FSWTHRU); a.Sswabs = P(A_SSW); a.Iswabs = P(A_ISW);
  double** outs[15] = {&a.fsurfn, &a.fcondtopn, &a.fsensn, &a.flatn, &a.fswabsn, &a.flwoutn, &a.evapn,
                       &a.freshn, &a.fsaltn, &a.fhocnn, &a.meltt, &a.melts, &a.meltb, &a.congel,
                       &a.snoice};
  for (int k = 0; k < 15; ++k) *outs[k] = P(A_OUT + k);
  a.mlt_onset = P(A_MLT); a.frz_onset = P(A_FRZ);
  a.errkey = c_->tkey.p; a.nupdates = c_->tkey.p + THERMO_COUNT_STRIDE;
  thermo_launch_list(a, s);
  c_->fan.fork(s);
  auto down = [&](int plane, double* h, int planes = 1) {
    CICE_REQUIRE(h != nullptr, "thermo_vertical: NULL array");
    CICE_HIP(hipMemcpyAsync(h, d.p + (size_t)plane * np, (size_t)planes * np * 8, hipMemcpyDeviceToHost, c_->cs()));
  };
  down(A_AICEN, aicen); down(A_TRCRN + it_T, trcrn + (size_t)it_T * np); down(A_VICEN, vicen); down(A_VSNON, vsnon);
  down(A_EICEN, eicen, NILYR); down(A_ESNON, esnon, NSLYR);
  down(A_FSWSFC, fswsfc); down(A_FSWINT, fswint); down(A_SSW, Sswabs, NSLYR); down(A_ISW, Iswabs, NILYR);
  double* houts[15] = {fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn, evapn, freshn, fsaltn,
                       fhocnn, meltt, melts, meltb, congel, snoice};
  for (int k = 0; k < 15; ++k) down(A_OUT + k, houts[k]);
  down(A_MLT, mlt_onset); down(A_FRZ, frz_onset);
  c_->fan.join();
  unsigned long long key = 0;
  CICE_HIP(hipMemcpyAsync(&key, c_->tkey.p, 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipStreamSynchronize(s));
  decode_err(key, nx, 1, indxi, indxj, l_stop, istop, jstop, nullptr, nullptr);
  CICE_CATCH
}

int cice_thermo_batch_alloc(cice_ctx* ctx, int nx, int ny, int nb) {
  CICE_TRY(ctx)
  CICE_REQUIRE(nx >= 3 && ny >= 3 && nb >= 1, "bad dimensions");
  c_->need_device();
  auto& t = c_->tb;
  t.nx = nx; t.ny = ny; t.nb = nb;
  const size_t np = (size_t)nx * ny, n2 = np * nb, nc = n2 * NCAT;
  std::vector<int32_t> hb;
  if (c_->have_domain && c_->dom.nblocks() == nb && c_->dom.nx_block == nx && c_->dom.ny_block == ny) {
    for (int gid : c_->dom.local) {
      const Block& b = c_->dom.all[gid];
      hb.insert(hb.end(), {b.ilo, b.ihi, b.own_jlo, b.own_jhi});  // owned rows only
    }
  } else {
    for (int b = 0; b < nb; ++b) hb.insert(hb.end(), {2, nx - 1, 2, ny - 1});
  }
  t.blk.alloc(hb.size());
  t.blk.upload(hb.data(), c_->stream);
  t.hblk = hb;
  t.mrg_in.alloc(5 * nc); t.mrg_acc.alloc(20 * n2); t.fz_in.alloc(7 * n2);
  t.aicen.alloc(nc); t.trcrn.alloc(nc * NTRCR); t.vicen.alloc(nc); t.vsnon.alloc(nc);
  t.eicen.alloc(nc * NILYR); t.esnon.alloc(nc * NSLYR);
  for (DevBuf<double>* d : {&t.flw, &t.potT, &t.Qa, &t.rhoa, &t.fsnow, &t.fbot, &t.Tbot, &t.mlt_onset,
                            &t.frz_onset})
    d->alloc(n2);
  for (DevBuf<double>* d : {&t.lhcoef, &t.shcoef, &t.fswsfc, &t.fswint, &t.fswthrun}) d->alloc(nc);
  t.Sswabs.alloc(nc * NSLYR); t.Iswabs.alloc(nc * NILYR);
  t.out15.alloc(nc * 15);
  t.out15.zero(c_->stream);
  c_->tkey.alloc(THERMO_STATUS_WORDS);
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

static void batch_upload(cice_ctx* c_, const cice_thermo_fields* h, bool with_fbot_tbot, bool with_coef = true) {
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0 && h, "cice_thermo_batch_alloc has not been called");
  struct U { DevBuf<double>* d; const double* h; };
  U us[] = {{&t.aicen, h->aicen}, {&t.trcrn, h->trcrn}, {&t.vicen, h->vicen}, {&t.vsnon, h->vsnon},
            {&t.eicen, h->eicen}, {&t.esnon, h->esnon}, {&t.flw, h->flw}, {&t.potT, h->potT},
            {&t.Qa, h->Qa}, {&t.rhoa, h->rhoa}, {&t.fsnow, h->fsnow}, {&t.fbot, h->fbot},
            {&t.Tbot, h->Tbot}, {&t.lhcoef, h->lhcoef}, {&t.shcoef, h->shcoef}, {&t.fswsfc, h->fswsfc},
            {&t.fswint, h->fswint}, {&t.fswthrun, h->fswthrun}, {&t.Sswabs, h->Sswabs},
            {&t.Iswabs, h->Iswabs}, {&t.mlt_onset, h->mlt_onset}, {&t.frz_onset, h->frz_onset}};
  for (U& x : us) {
    if (!with_fbot_tbot && (x.d == &t.fbot || x.d == &t.Tbot)) continue;   // produced on the device
    if (!with_coef && (x.d == &t.lhcoef || x.d == &t.shcoef)) continue;    // likewise (atmo_boundary_layer)
    CICE_REQUIRE(x.h != nullptr, "cice_thermo_batch_upload: NULL field");
    if (x.d == &t.trcrn && c_->have_thermo) {
      // of trcrn(nx, ny, max_ntrcr, ncat, nblocks) the column physics reads and writes the surface temperature only:
      // that plane of every (category, block), one strided copy (5 planes at ncat = 5 instead of 25)
      const size_t np = (size_t)t.nx * t.ny, o = (size_t)(c_->tp.nt_Tsfc - 1) * np;
      CICE_HIP(hipMemcpy2DAsync(t.trcrn.p + o, (size_t)NTRCR * np * 8, x.h + o, (size_t)NTRCR * np * 8, np * 8,
                                (size_t)NCAT * t.nb, hipMemcpyHostToDevice, c_->cs()));
      continue;
    }
    x.d->upload(x.h, c_->cs());
  }
  if (c_->have_thermo && !c_->tp.calc_Tsfc) {  // surface fluxes are inputs (ice_therm_vertical.F90:213-217)
    const size_t nc = (size_t)t.nx * t.ny * t.nb * NCAT;
    const double* in3[3] = {h->fsurfn, h->fcondtopn, h->flatn};
    const int plane[3] = {0, 1, 3};
    for (int k = 0; k < 3; ++k) {
      CICE_REQUIRE(in3[k] != nullptr, "cice_thermo_batch_upload: calc_Tsfc = F needs fsurfn, fcondtopn, flatn");
      CICE_HIP(hipMemcpyAsync(t.out15.p + (size_t)plane[k] * nc, in3[k], nc * 8, hipMemcpyHostToDevice, c_->cs()));
    }
  }
}

int cice_thermo_batch_upload(cice_ctx* ctx, const cice_thermo_fields* h) {
  CICE_TRY(ctx) c_->chain_ready = false;
  batch_upload(c_, h, true);
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

// number of columns updated: the sum of the kernel's counters (therm.h)
static long long status_count(const unsigned long long* h) {
  long long n = 0;
  for (int k = 0; k < THERMO_COUNT_SLOTS; ++k) n += (long long)h[THERMO_COUNT_STRIDE * (1 + k)];
  return n;
}

// launches the dense kernel; the status words (error key, update counters) land in `status` once the
// stream has been synchronised
static void batch_step(cice_ctx* c_, double dt, double yday, unsigned long long status[THERMO_STATUS_WORDS], float* elapsed_ms,
                       hipEvent_t* ev) {
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0, "cice_thermo_batch_alloc has not been called");
  CICE_REQUIRE(c_->have_thermo, "cice_thermo_init has not been called");
  hipStream_t s = c_->stream;
  const size_t nc = (size_t)t.nx * t.ny * t.nb * NCAT;
  CICE_HIP(hipMemsetAsync(c_->tkey.p, 0xff, 8, s));
  CICE_HIP(hipMemsetAsync(c_->tkey.p + 1, 0, (THERMO_STATUS_WORDS - 1) * 8, s));
  ThermoArgs a{};
  a.p = c_->tp; a.nx = t.nx; a.ny = t.ny; a.ncat = NCAT; a.nblocks = t.nb; a.dt = dt; a.yday = yday;
  a.icells = 0; a.indxi = nullptr; a.indxj = nullptr; a.blk = t.blk.p;
  a.aicen = t.aicen.p; a.trcrn = t.trcrn.p; a.vicen = t.vicen.p; a.vsnon = t.vsnon.p;
  a.eicen = t.eicen.p; a.esnon = t.esnon.p; a.flw = t.flw.p; a.potT = t.potT.p; a.Qa = t.Qa.p;
  a.rhoa = t.rhoa.p; a.fsnow = t.fsnow.p; a.fbot = t.fbot.p; a.Tbot = t.Tbot.p;
  a.lhcoef = t.lhcoef.p; a.shcoef = t.shcoef.p; a.fswsfc = t.fswsfc.p; a.fswint = t.fswint.p;
  a.fswthrun = t.fswthrun.p; a.Sswabs = t.Sswabs.p; a.Iswabs = t.Iswabs.p;
  double** outs[15] = {&a.fsurfn, &a.fcondtopn, &a.fsensn, &a.flatn, &a.fswabsn, &a.flwoutn, &a.evapn,
                       &a.freshn, &a.fsaltn, &a.fhocnn, &a.meltt, &a.melts, &a.meltb, &a.congel,
                       &a.snoice};
  for (int k = 0; k < 15; ++k) *outs[k] = t.out15.p + (size_t)k * nc;
  a.mlt_onset = t.mlt_onset.p; a.frz_onset = t.frz_onset.p;
  a.errkey = c_->tkey.p; a.nupdates = c_->tkey.p + THERMO_COUNT_STRIDE;
  if (t.niter.n != nc) {
    t.niter.alloc(nc);
    t.niter.zero(s);
  }
  a.niter = t.niter.p;
  if (elapsed_ms) {
    CICE_HIP(hipEventCreate(&ev[0]));
    CICE_HIP(hipEventCreate(&ev[1]));
    CICE_HIP(hipEventRecord(ev[0], s));
  }
  static const int env_chunk = [] { const char* e = std::getenv("CICE4_AMD_THERMO_SORT"); return e ? std::atoi(e) : -1; }();
  const int chunk = env_chunk >= 0 ? env_chunk : t.sort_chunk;
  static const int env_group = [] { const char* e = std::getenv("CICE4_AMD_THERMO_GROUP"); return e ? std::atoi(e) : -1; }();
  const int group = env_group > 0 ? env_group : t.sort_group;
  if (chunk >= 256 && chunk <= 2048 && chunk % 256 == 0 && (group == 1 || group == 2 || group == 4 || group == 8 ||
                                                            group == 16 || group == 32)) {
    const size_t np = (size_t)t.nx * t.ny;
    const size_t want = thermo_sorted_plane(np, chunk) * t.nb * NCAT;
    if (t.perm.n != want) t.perm.alloc(want);
    // the Tsfc tracer plane of (category, block) cb: trcrn is (nx, ny, max_ntrcr, ncat, nb)
    thermo_launch_sorted(a, chunk, group, t.perm.p, t.trcrn.p + (size_t)(c_->tp.nt_Tsfc - 1) * np, (size_t)NTRCR * np, s);
  } else {
    thermo_launch_dense(a, s);
  }
  if (elapsed_ms) CICE_HIP(hipEventRecord(ev[1], s));
  CICE_HIP(hipMemcpyAsync(status, c_->tkey.p, THERMO_STATUS_WORDS * 8, hipMemcpyDeviceToHost, s));
}

const char* cice_build_flavour(void) {
#ifdef CICE4_AMD_AUSCOM
  return "auscom";
#else
  return "standalone";
#endif
}

int cice_set_auscom(cice_ctx* ctx, double cosw, double sinw, double dragio, int use_ocnslope) {
  CICE_TRY(ctx)
#ifdef CICE4_AMD_AUSCOM
  const double want[4] = {cosw, sinw, dragio, use_ocnslope ? 1.0 : 0.0};
  if (c_->nml_set && !std::memcmp(want, c_->nml, sizeof(want))) return CICE_OK;   // called before every evp(dt)
  CICE_HIP(hipStreamSynchronize(c_->stream));   // nothing in flight reads the old values
  evp_set_namelist(cosw, sinw, dragio, use_ocnslope);
  std::memcpy(c_->nml, want, sizeof(want));
  c_->nml_set = true;
#else
  (void)cosw; (void)sinw; (void)dragio; (void)use_ocnslope;
  throw Error{CICE_EINVAL, "cice_set_auscom: this is the stand-alone build of the library (libcice4_amd.so); the coupled "
                           "one, with the access-om constants and the hemisphere-dependent turning angle, is "
                           "libcice4_amd_auscom.so"};
#endif
  CICE_CATCH
}

int cice_thermo_set_chio(cice_ctx* ctx, double chio) {
  CICE_TRY(ctx)
#ifdef CICE4_AMD_AUSCOM
  c_->chio = chio;   // a kernel argument of frzmlt_bottom_lateral: later launches see it
#else
  (void)chio;
  throw Error{CICE_EINVAL, "cice_thermo_set_chio: this is the stand-alone build of the library (chio is the constant "
                           "0.006 there, ice_therm_vertical.F90:680); the coupled one is libcice4_amd_auscom.so"};
#endif
  CICE_CATCH
}

int cice_thermo_set_option(cice_ctx* ctx, const char* key, int value) {
  CICE_TRY(ctx)
  CICE_REQUIRE(key, "NULL key");
  if (!std::strcmp(key, "sort_chunk")) {
    CICE_REQUIRE(value == 0 || (value >= 256 && value <= 2048 && value % 256 == 0), "sort_chunk must be 0 or 256 .. 2048 in steps of 256");
    c_->tb.sort_chunk = value;
  } else if (!std::strcmp(key, "sort_group")) {
    CICE_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32, "sort_group must be 1, 2, 4, 8, 16 or 32");
    c_->tb.sort_group = value;
  } else {
    throw Error{CICE_EINVAL, std::string("unknown option ") + key};
  }
  CICE_CATCH
}

int cice_thermo_batch_step(cice_ctx* ctx, double dt, double yday, long long* n_updates,
                           int32_t* l_stop, int32_t* istop, int32_t* jstop, int32_t* nstop,
                           int32_t* bstop, float* elapsed_ms) {
  CICE_TRY(ctx) c_->chain_ready = false;
  CICE_REQUIRE(l_stop && istop && jstop, "NULL status pointer");
  unsigned long long h[THERMO_STATUS_WORDS];
  hipEvent_t ev[2] = {nullptr, nullptr};
  batch_step(c_, dt, yday, h, elapsed_ms, ev);
  CICE_HIP(hipStreamSynchronize(c_->stream));
  if (elapsed_ms) {
    CICE_HIP(hipEventElapsedTime(elapsed_ms, ev[0], ev[1]));
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
  }
  if (n_updates) *n_updates = status_count(h);
  decode_err(h[0], c_->tb.nx, NCAT, nullptr, nullptr, l_stop, istop, jstop, nstop, bstop);
  CICE_CATCH
}

static void batch_download(cice_ctx* c_, cice_thermo_fields* h) {
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0 && h, "cice_thermo_batch_alloc has not been called");
  const size_t nc = (size_t)t.nx * t.ny * t.nb * NCAT;
  struct D { const DevBuf<double>* d; double* h; };
  D ds[] = {{&t.aicen, h->aicen}, {&t.trcrn, h->trcrn}, {&t.vicen, h->vicen}, {&t.vsnon, h->vsnon},
            {&t.eicen, h->eicen}, {&t.esnon, h->esnon}, {&t.fswsfc, h->fswsfc}, {&t.fswint, h->fswint},
            {&t.Sswabs, h->Sswabs}, {&t.Iswabs, h->Iswabs}, {&t.mlt_onset, h->mlt_onset},
            {&t.frz_onset, h->frz_onset}};
  for (D& x : ds) {
    if (!x.h) continue;
    if (x.d == &t.trcrn && c_->have_thermo) {   // the surface-temperature plane, as it was uploaded
      const size_t np = (size_t)t.nx * t.ny, o = (size_t)(c_->tp.nt_Tsfc - 1) * np;
      CICE_HIP(hipMemcpy2DAsync(x.h + o, (size_t)NTRCR * np * 8, t.trcrn.p + o, (size_t)NTRCR * np * 8, np * 8,
                                (size_t)NCAT * t.nb, hipMemcpyDeviceToHost, c_->cs()));
      continue;
    }
    x.d->download(x.h, c_->cs());
  }
  double* houts[15] = {h->fsurfn, h->fcondtopn, h->fsensn, h->flatn, h->fswabsn, h->flwoutn, h->evapn,
                       h->freshn, h->fsaltn, h->fhocnn, h->meltt, h->melts, h->meltb, h->congel,
                       h->snoice};
  for (int k = 0; k < 15; ++k)
    if (houts[k])
      CICE_HIP(hipMemcpyAsync(houts[k], t.out15.p + (size_t)k * nc, nc * 8, hipMemcpyDeviceToHost, c_->cs()));
}

int cice_thermo_batch_download(cice_ctx* ctx, cice_thermo_fields* h) {
  CICE_TRY(ctx) c_->chain_ready = false;
  batch_download(c_, h);
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

// aicen_init_dev: device copy of the initial concentrations (cice_step_therm1 keeps one); otherwise
// f->aicen_init is uploaded
// phases: the uploads, the kernel and the downloads can be asked for separately (cice_step_therm1 puts its copies on
// side streams and the uploads in front of every kernel)
enum { MRG_UP = 1, MRG_RUN = 2, MRG_DOWN = 4, MRG_ALL = 7 };
static void batch_merge(cice_ctx* c_, const cice_merge_fields* f, const double* aicen_init_dev,
                        bool atmo_on_device = false, int phases = MRG_ALL) {
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0 && f, "cice_thermo_batch_alloc has not been called");
  hipStream_t s = c_->stream;
  const size_t n2 = (size_t)t.nx * t.ny * t.nb, nc = n2 * NCAT;
  DevBuf<double>&up = t.mrg_in, &acc = t.mrg_acc;
  const double* hin[5] = {f->aicen_init, f->strairxn, f->strairyn, f->Trefn, f->Qrefn};
  for (int k = 0; k < 5 && (phases & MRG_UP); ++k) {
    if (k == 0 && aicen_init_dev) continue;
    if (k > 0 && atmo_on_device) continue;    // strairxn, strairyn, Trefn, Qrefn were produced in place
    CICE_REQUIRE(hin[k] != nullptr, "cice_thermo_batch_merge: NULL input");
    CICE_HIP(hipMemcpyAsync(up.p + (size_t)k * nc, hin[k], nc * 8, hipMemcpyHostToDevice, c_->cs()));
  }
  for (int k = 0; k < 20 && (phases & MRG_UP); ++k) {
    CICE_REQUIRE(f->acc[k] != nullptr, "cice_thermo_batch_merge: NULL accumulator");
    CICE_HIP(hipMemcpyAsync(acc.p + (size_t)k * n2, f->acc[k], n2 * 8, hipMemcpyHostToDevice, c_->cs()));
  }
  MergeArgs a{};
  a.nx = t.nx; a.ny = t.ny; a.ncat = NCAT; a.nblocks = t.nb; a.blk = t.blk.p;
  a.aicen_init = aicen_init_dev ? aicen_init_dev : up.p; a.flw = t.flw.p;
  auto out = [&](int k) { return (const double*)(t.out15.p + (size_t)k * nc); };
  // out15 order: fsurfn fcondtopn fsensn flatn fswabsn flwoutn evapn freshn fsaltn fhocnn meltt melts
  //              meltb congel snoice
  const double* src[20] = {up.p + nc, up.p + 2 * nc, out(0), out(1), out(2), out(3), out(4), out(5),
                           out(6), up.p + 3 * nc, up.p + 4 * nc, out(7), out(8), out(9), t.fswthrun.p,
                           out(10), out(12), out(11), out(13), out(14)};
  for (int k = 0; k < 20; ++k) {
    a.src[k] = src[k];
    a.acc[k] = acc.p + (size_t)k * n2;
  }
  if (phases & MRG_RUN) merge_launch(a, s);
  for (int k = 0; k < 20 && (phases & MRG_DOWN); ++k)
    CICE_HIP(hipMemcpyAsync(f->acc[k], acc.p + (size_t)k * n2, n2 * 8, hipMemcpyDeviceToHost, c_->cs()));
}

int cice_thermo_batch_merge(cice_ctx* ctx, const cice_merge_fields* f) {
  CICE_TRY(ctx) c_->chain_ready = false;
  batch_merge(c_, f, nullptr);
  CICE_HIP(hipStreamSynchronize(c_->stream));
  CICE_CATCH
}

// One call for the thermodynamic half of a time step on all local blocks (the work of step_therm1,
// drivers/cice4/CICE_RunMod.F90:260-598, minus atmo_boundary_layer, whose per-category outputs are inputs
// here): ONE upload, frzmlt_bottom_lateral (:363) -> thermo_vertical for every category (:502) ->
// merge_fluxes (:565) on the device, ONE download, one synchronisation.
static void step_therm1(cice_ctx* c_, double dt, double yday, cice_thermo_fields* st, const cice_frzmlt_fields* fz,
                        const cice_merge_fields* mg, const cice_atmo_fields* atm, long long* n_updates,
                        int32_t* l_stop, int32_t* istop, int32_t* jstop, int32_t* nstop, int32_t* bstop) {
  auto& t = c_->tb;
  CICE_REQUIRE(t.nb > 0, "cice_thermo_batch_alloc has not been called");
  CICE_REQUIRE(c_->have_thermo, "cice_thermo_init has not been called");
  CICE_REQUIRE(st && fz && mg && l_stop && istop && jstop, "NULL argument");
  CICE_REQUIRE(fz->aice && fz->frzmlt && fz->sst && fz->Tf && fz->strocnxT && fz->strocnyT, "NULL frzmlt input");
  hipStream_t s = c_->stream;
  const size_t np = (size_t)t.nx * t.ny, n2 = np * t.nb, nc = n2 * NCAT;
  // every upload first, spread over the side streams (about 150 separate host arrays), then the kernels
  c_->fan.fork(s);
  batch_upload(c_, st, false, atm == nullptr);
  if (atm) {
    CICE_REQUIRE(atm->uatm && atm->vatm && atm->wind && atm->zlvl, "NULL atmosphere input");
    CICE_REQUIRE(atm->calc_strair || (atm->strax && atm->stray), "calc_strair = F needs strax, stray");
    if (t.atm_in.n < 6 * n2) t.atm_in.alloc(6 * n2);
    const double* ain[6] = {atm->uatm, atm->vatm, atm->wind, atm->zlvl, atm->strax, atm->stray};
    for (int k = 0; k < (atm->calc_strair ? 4 : 6); ++k)
      CICE_HIP(hipMemcpyAsync(t.atm_in.p + (size_t)k * n2, ain[k], n2 * 8, hipMemcpyHostToDevice, c_->cs()));
  }
  {
    const double* fin[6] = {fz->aice, fz->frzmlt, fz->sst, fz->Tf, fz->strocnxT, fz->strocnyT};
    for (int k = 0; k < 6; ++k)
      CICE_HIP(hipMemcpyAsync(t.fz_in.p + (size_t)k * n2, fin[k], n2 * 8, hipMemcpyHostToDevice, c_->cs()));
  }
  batch_merge(c_, mg, t.mrg_in.p, atm != nullptr, MRG_UP);
  c_->fan.join();
  if (atm) {   // atmo_boundary_layer for every category (CICE_RunMod.F90:402-425), on the state before the update
    AtmoArgs a{};
    a.p.init();
    a.nx = t.nx; a.ny = t.ny; a.ncat = NCAT; a.nblocks = t.nb; a.ocn = 0; a.calc_strair = atm->calc_strair != 0;
    a.blk = t.blk.p; a.aicen = t.aicen.p; a.Tsf = t.trcrn.p; a.it_Tsfc = c_->tp.nt_Tsfc - 1;
    a.potT = t.potT.p; a.Qa = t.Qa.p; a.rhoa = t.rhoa.p;
    a.uatm = t.atm_in.p; a.vatm = t.atm_in.p + n2; a.wind = t.atm_in.p + 2 * n2; a.zlvl = t.atm_in.p + 3 * n2;
    a.strax = t.atm_in.p + 4 * n2; a.stray = t.atm_in.p + 5 * n2;
    a.strx = t.mrg_in.p + nc; a.stry = t.mrg_in.p + 2 * nc; a.Tref = t.mrg_in.p + 3 * nc; a.Qref = t.mrg_in.p + 4 * nc;
    a.lhcoef = t.lhcoef.p; a.shcoef = t.shcoef.p;
    atmo_launch_dense(a, s);
  }
  for (int b = 0; b < t.nb; ++b) {   // frzmlt_bottom_lateral per block, on the uploaded enthalpies
    FrzmltArgs a{};
    a.nx = t.nx; a.ny = t.ny; a.dt = dt; a.ustar_min = c_->tp.ustar_min; a.chio = c_->chio;
    a.ilo = t.hblk[4 * b]; a.ihi = t.hblk[4 * b + 1]; a.jlo = t.hblk[4 * b + 2]; a.jhi = t.hblk[4 * b + 3];
    const size_t o = (size_t)b * np;
    a.aice = t.fz_in.p + o; a.frzmlt = t.fz_in.p + n2 + o; a.sst = t.fz_in.p + 2 * n2 + o;
    a.Tf = t.fz_in.p + 3 * n2 + o; a.strocnxT = t.fz_in.p + 4 * n2 + o; a.strocnyT = t.fz_in.p + 5 * n2 + o;
    a.Tbot = t.Tbot.p + o; a.fbot = t.fbot.p + o; a.rside = t.fz_in.p + 6 * n2 + o;
    a.eicen = t.eicen.p + (size_t)b * NCAT * NILYR * np; a.esnon = t.esnon.p + (size_t)b * NCAT * NSLYR * np;
    frzmlt_launch(a, s);
  }
  // aicen_init of merge_fluxes = the concentrations before the column update (CICE_RunMod.F90:342-355)
  CICE_HIP(hipMemcpyAsync(t.mrg_in.p, t.aicen.p, nc * 8, hipMemcpyDeviceToDevice, s));
  unsigned long long h[THERMO_STATUS_WORDS];
  batch_step(c_, dt, yday, h, nullptr, nullptr);
  batch_merge(c_, mg, t.mrg_in.p, atm != nullptr, MRG_RUN);
  c_->fan.fork(s);   // ... and every download after the last kernel
  batch_merge(c_, mg, t.mrg_in.p, atm != nullptr, MRG_DOWN);
  batch_download(c_, st);
  if (atm) {
    double* aout[6] = {atm->strairxn, atm->strairyn, atm->Trefn, atm->Qrefn, atm->lhcoef, atm->shcoef};
    const double* asrc[6] = {t.mrg_in.p + nc, t.mrg_in.p + 2 * nc, t.mrg_in.p + 3 * nc, t.mrg_in.p + 4 * nc,
                             t.lhcoef.p, t.shcoef.p};
    for (int k = 0; k < 6; ++k)
      if (aout[k]) CICE_HIP(hipMemcpyAsync(aout[k], asrc[k], nc * 8, hipMemcpyDeviceToHost, c_->cs()));
  }
  if (fz->Tbot) t.Tbot.download(fz->Tbot, c_->cs());
  if (fz->fbot) t.fbot.download(fz->fbot, c_->cs());
  if (fz->rside) CICE_HIP(hipMemcpyAsync(fz->rside, t.fz_in.p + 6 * n2, n2 * 8, hipMemcpyDeviceToHost, c_->cs()));
  c_->fan.join();
  CICE_HIP(hipStreamSynchronize(s));
  if (n_updates) *n_updates = status_count(h);
  decode_err(h[0], t.nx, NCAT, nullptr, nullptr, l_stop, istop, jstop, nstop, bstop);
}

int cice_step_therm1(cice_ctx* ctx, double dt, double yday, cice_thermo_fields* st,
                     const cice_frzmlt_fields* fz, const cice_merge_fields* mg, long long* n_updates,
                     int32_t* l_stop, int32_t* istop, int32_t* jstop, int32_t* nstop, int32_t* bstop) {
  CICE_TRY(ctx) c_->chain_ready = false;
  step_therm1(c_, dt, yday, st, fz, mg, nullptr, n_updates, l_stop, istop, jstop, nstop, bstop);
  CICE_CATCH
}

// ... with atmo_boundary_layer on the device as well: lhcoef / shcoef of `st` and the four atmosphere fields of `mg`
// are not read; what the routine produced comes back through `atm` where asked for.
int cice_step_therm1_abl(cice_ctx* ctx, double dt, double yday, cice_thermo_fields* st,
                         const cice_frzmlt_fields* fz, const cice_merge_fields* mg, const cice_atmo_fields* atm,
                         long long* n_updates, int32_t* l_stop, int32_t* istop, int32_t* jstop, int32_t* nstop,
                         int32_t* bstop) {
  CICE_TRY(ctx) c_->chain_ready = false;
  CICE_REQUIRE(atm != nullptr, "NULL argument");
  step_therm1(c_, dt, yday, st, fz, mg, atm, n_updates, l_stop, istop, jstop, nstop, bstop);
  CICE_CATCH
}

// atmo_boundary_layer (source/ice_atmo.F90:56-384), one block, host pointers, the reference's argument list
// (sfctype: 0 'ice', 1 'ocn'; calc_strair is the module variable of ice_atmo).
int cice_atmo_boundary_layer(cice_ctx* ctx, int nx, int ny, int sfctype, int icells, const int32_t* indxi,
                             const int32_t* indxj, const double* Tsf, const double* potT, const double* uatm,
                             const double* vatm, const double* wind, const double* zlvl, const double* Qa,
                             const double* rhoa, int calc_strair, double* strx, double* stry, double* Tref,
                             double* Qref, double* delt, double* delq, double* lhcoef, double* shcoef) {
  CICE_TRY(ctx)
  CICE_REQUIRE(nx >= 1 && ny >= 1 && icells >= 0 && (size_t)icells <= (size_t)nx * ny, "bad dimensions");
  CICE_REQUIRE(sfctype == 0 || sfctype == 1, "sfctype: 0 'ice' or 1 'ocn'");
  CICE_REQUIRE(Tsf && potT && uatm && vatm && wind && zlvl && Qa && rhoa && strx && stry && Tref && Qref && delt &&
                   delq && lhcoef && shcoef && (icells == 0 || (indxi && indxj)), "atmo_boundary_layer: NULL array");
  c_->need_device();
  hipStream_t s = c_->stream;
  const size_t np = (size_t)nx * ny;
  DevBuf<double>& d = c_->fz_stage;
  if (d.n < 16 * np) d.alloc(16 * np);
  DevBuf<int32_t>& li = c_->tv_list;
  if (li.n < 2 * np) li.alloc(2 * np);
  const double* in[8] = {Tsf, potT, uatm, vatm, wind, zlvl, Qa, rhoa};
  for (int k = 0; k < 8; ++k) CICE_HIP(hipMemcpyAsync(d.p + (size_t)k * np, in[k], np * 8, hipMemcpyHostToDevice, s));
  if (!calc_strair) {   // strx, stry are left as they are (:309)
    CICE_HIP(hipMemcpyAsync(d.p + 8 * np, strx, np * 8, hipMemcpyHostToDevice, s));
    CICE_HIP(hipMemcpyAsync(d.p + 9 * np, stry, np * 8, hipMemcpyHostToDevice, s));
  }
  if (icells > 0) {
    CICE_HIP(hipMemcpyAsync(li.p, indxi, (size_t)icells * 4, hipMemcpyHostToDevice, s));
    CICE_HIP(hipMemcpyAsync(li.p + np, indxj, (size_t)icells * 4, hipMemcpyHostToDevice, s));
  }
  AtmoArgs a{};
  a.p.init();
  a.nx = nx; a.ny = ny; a.ncat = 1; a.nblocks = 1; a.ocn = sfctype; a.calc_strair = calc_strair != 0;
  a.icells = icells; a.indxi = li.p; a.indxj = li.p + np;
  a.Tsf = d.p; a.potT = d.p + np; a.uatm = d.p + 2 * np; a.vatm = d.p + 3 * np; a.wind = d.p + 4 * np;
  a.zlvl = d.p + 5 * np; a.Qa = d.p + 6 * np; a.rhoa = d.p + 7 * np;
  double* out[8] = {strx, stry, Tref, Qref, delt, delq, lhcoef, shcoef};
  double** dev[8] = {&a.strx, &a.stry, &a.Tref, &a.Qref, &a.delt, &a.delq, &a.lhcoef, &a.shcoef};
  for (int k = 0; k < 8; ++k) *dev[k] = d.p + (size_t)(8 + k) * np;
  atmo_launch_list(a, s);
  for (int k = 0; k < 8; ++k) CICE_HIP(hipMemcpyAsync(out[k], *dev[k], np * 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipStreamSynchronize(s));
  CICE_CATCH
}

int cice_frzmlt_bottom_lateral(cice_ctx* ctx, int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                               double dt, const double* aice, const double* frzmlt,
                               const double* eicen, const double* esnon, const double* sst,
                               const double* Tf, const double* strocnxT, const double* strocnyT,
                               double* Tbot, double* fbot, double* rside) {
  CICE_TRY(ctx)
  CICE_REQUIRE(c_->have_thermo, "cice_thermo_init has not been called");
  CICE_REQUIRE(nx >= 1 && ny >= 1 && ilo >= 1 && ihi <= nx && jlo >= 1 && jhi <= ny, "bad dimensions");
  CICE_REQUIRE(aice && frzmlt && eicen && esnon && sst && Tf && strocnxT && strocnyT && Tbot && fbot && rside,
               "frzmlt_bottom_lateral: NULL array");     // before anything is queued on the stream
  c_->need_device();
  hipStream_t s = c_->stream;
  const size_t np = (size_t)nx * ny;
  const int NE = NCAT * NILYR, NSN = NCAT * NSLYR;
  DevBuf<double>& d = c_->fz_stage;
  if (d.n < (size_t)(9 + NE + NSN) * np) d.alloc((size_t)(9 + NE + NSN) * np);
  auto up = [&](size_t plane, const double* h, size_t planes = 1) {
    CICE_REQUIRE(h != nullptr, "frzmlt_bottom_lateral: NULL array");
    CICE_HIP(hipMemcpyAsync(d.p + plane * np, h, planes * np * 8, hipMemcpyHostToDevice, s));
  };
  up(0, aice); up(1, frzmlt); up(2, sst); up(3, Tf); up(4, strocnxT); up(5, strocnyT);
  up(9, eicen, NE); up(9 + NE, esnon, NSN);
  FrzmltArgs a{};
  a.nx = nx; a.ny = ny; a.ilo = ilo; a.ihi = ihi; a.jlo = jlo; a.jhi = jhi; a.dt = dt;
  a.ustar_min = c_->tp.ustar_min; a.chio = c_->chio;
  a.aice = d.p; a.frzmlt = d.p + np; a.sst = d.p + 2 * np; a.Tf = d.p + 3 * np;
  a.strocnxT = d.p + 4 * np; a.strocnyT = d.p + 5 * np;
  a.Tbot = d.p + 6 * np; a.fbot = d.p + 7 * np; a.rside = d.p + 8 * np;
  a.eicen = d.p + 9 * np; a.esnon = d.p + (size_t)(9 + NE) * np;
  frzmlt_launch(a, s);
  CICE_HIP(hipMemcpyAsync(Tbot, a.Tbot, np * 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipMemcpyAsync(fbot, a.fbot, np * 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipMemcpyAsync(rside, a.rside, np * 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipStreamSynchronize(s));
  CICE_CATCH
}

// ---- horizontal transport ---------------------------------------------------------------------
int cice_transport_init(cice_ctx* ctx, const cice_transport_config* cfg, const cice_transport_grid* grid) {
  CICE_TRY(ctx)
  CICE_REQUIRE(cfg && grid, "NULL argument");
  c_->need_halo();
  c_->transport.reset(new Transport(c_->dom, *c_->halo, c_->stream, c_->fan));
  c_->transport->init(*cfg, *grid);
  CICE_CATCH
}

int cice_transport_remap(cice_ctx* ctx, double dt, const cice_transport_fields* f, int32_t* l_stop,
                         int32_t* istop, int32_t* jstop) {
  CICE_TRY(ctx)
  CICE_REQUIRE(c_->transport != nullptr, "cice_transport_init has not been called");
  CICE_REQUIRE(f, "NULL argument");
  if (c_->chain_on && c_->chain_ready && c_->evp && f->aice0 == c_->chain.aice0 && f->trcrn == c_->chain.trcrn &&
      f->vsnon == c_->chain.vsnon && f->eicen == c_->chain.eicen && f->esnon == c_->chain.esnon &&
      f->aicen == c_->chain_aicen && f->vicen == c_->chain_vicen && f->uvel == c_->chain_u && f->vvel == c_->chain_v)
    c_->transport->adopt(c_->evp->d_uv(), c_->evp->d_aicen(), c_->evp->d_vicen());
  c_->chain_ready = false;
  c_->transport->remap(dt, *f, l_stop, istop, jstop);
  CICE_CATCH
}

// test aid, no device needed: smallest shift of the sweep kernel's strip layout that is right for a block of ncol columns
// (K levels, S wavefronts per level), -1 if none; *strips = column strips of the block with it
int cice_debug_skew_layout(int K, int S, int ncol, int cyclic, int* strips) {
  if (K < 2 || K > 8 || (S != 1 && S != 3) || ncol < 1) return -2;
  for (int shift = 0; shift < 2 * K + 4; ++shift)
    if (evp_skew_layout_ok(K, S, ncol, shift, cyclic != 0)) {
      const int ownw = 62 * S + 2 - 2 * K, f = ownw - 1 - shift, npos = ncol + 1;
      if (strips) *strips = npos <= f ? 1 : 1 + (npos - f + ownw - 1) / ownw;
      return shift;
    }
  return -1;
}

// test aid, no device needed: one strip's step of the measured balancing of the sweep's segments (cice::balance_strip)
int cice_debug_balance_strip(int rows, int n, const int32_t* ends, const double* durations, const double* weights,
                             const unsigned char* rows_with_ice, int32_t* new_ends, double* total) {
  if (rows < 1 || n < 1 || !ends || !durations || !weights || !new_ends) return -2;
  for (int i = 0; i < n; ++i)
    if (ends[i] < (i ? ends[i - 1] : 0) || ends[i] > rows) return -2;
  std::vector<double> cost((size_t)rows);
  std::vector<int> e(ends, ends + n), ne((size_t)n);
  const double t = cice::balance_strip(rows, n, e.data(), durations, weights, rows_with_ice, cost.data(), ne.data());
  if (total) *total = t;
  for (int i = 0; i < n; ++i) new_ends[i] = t > 0 ? ne[(size_t)i] : ends[i];
  return 0;
}

int cice_transport_chain(cice_ctx* ctx, const cice_transport_fields* f) {
  CICE_TRY(ctx)
  c_->chain_ready = false;
  c_->chain_on = false;
  if (f) {
    CICE_REQUIRE(c_->transport != nullptr, "cice_transport_chain: cice_transport_init has not been called");
    CICE_REQUIRE(f->aice0 && f->aicen && f->trcrn && f->vicen && f->vsnon && f->eicen && f->esnon && f->uvel && f->vvel,
                 "cice_transport_chain: NULL field");
    c_->chain = *f;
    c_->chain_on = true;
  }
  CICE_CATCH
}

int cice_transport_upwind_init(cice_ctx* ctx, const cice_transport_config* cfg, int nt_Tsfc, const double* HTE,
                               const double* HTN, const double* tarea) {
  CICE_TRY(ctx)
  CICE_REQUIRE(cfg, "NULL argument");
  c_->need_halo();
  c_->upwind.reset(new Upwind(c_->dom, *c_->halo, c_->stream, c_->fan));
  c_->upwind->init(*cfg, nt_Tsfc, HTE, HTN, tarea);
  CICE_CATCH
}

int cice_transport_upwind(cice_ctx* ctx, double dt, const cice_transport_fields* f) {
  CICE_TRY(ctx) c_->chain_ready = false;
  CICE_REQUIRE(c_->upwind != nullptr, "cice_transport_upwind_init has not been called");
  CICE_REQUIRE(f, "NULL argument");
  c_->upwind->step(dt, *f);
  CICE_CATCH
}

// test aid (not part of the drop-in surface): see Transport::debug_stop / debug_fetch
int cice_transport_debug(cice_ctx* ctx, int stop_stage, int which, double* out, long long* count) {
  CICE_TRY(ctx)
  CICE_REQUIRE(c_->transport != nullptr, "cice_transport_init has not been called");
  c_->transport->debug_stop(stop_stage);
  const size_t n = which >= 0 ? c_->transport->debug_fetch(which, out) : 0;
  if (count) *count = (long long)n;
  CICE_CATCH
}

}  // extern "C"
