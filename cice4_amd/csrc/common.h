// Shared definitions for the HIP hot path (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/cice4_amd.h"
#include "libm_exact.h"

namespace cice {

// drivers/cice4/ice_constants.F90:49-121,132-179 (CICE default parameter set)
namespace K {
constexpr double rhos = 330.0, rhoi = 917.0, rhow = 1026.0;
constexpr double cp_ice = 2106.0, depressT = 0.054, emissivity = 0.95;
#ifdef CICE4_AMD_AUSCOM   // the coupled flavour of the library: drivers/access-om/ice_constants.F90:21,48 (MOM's values)
constexpr double cp_ocn = 3989.24495292815, ice_ref_salinity = 5.0;
#else
constexpr double cp_ocn = 4218.0, ice_ref_salinity = 4.0;
#endif
constexpr double dragio = 0.00536, gravit = 9.80616;
constexpr double pi = 3.14159265358979323846;
constexpr double stefan_boltzmann = 567.0e-10, Tffresh = 273.15, Lsub = 2.835e6, Lvap = 2.501e6;
constexpr double Lfresh = Lsub - Lvap;
constexpr double kice = 2.03, ksno = 0.30;
constexpr double Tocnfrz = -1.8;   // :72 (a namelist variable with this default in the coupled build)
constexpr double qqqice = 11637800.0, TTTice = 5897.8;
constexpr double puny = 1.0e-11;
constexpr double c0 = 0.0, c1 = 1.0, c2 = 2.0, c4 = 4.0, p5 = 0.5, p25 = 0.25, p1 = 0.1,
                 p001 = 0.001;
constexpr double p166 = 1.0 / 6.0, p333 = 1.0 / 3.0, p111 = 1.0 / 9.0, p222 = 2.0 / 9.0;
constexpr double p055 = p111 * 0.5, p027 = p055 * 0.5;  // halved, not 1/18, 1/36 (:170-171)
// source/ice_dyn_evp.F90:76-88
constexpr double eyc = 0.36, a_min = 0.001, m_min = 0.01;
#ifndef CICE4_AMD_AUSCOM   // namelist variables in the coupled flavour (:91-97): device constants in evp.hip
constexpr double dragw = dragio * rhow, cosw = 1.0, sinw = 0.0;
#endif
// source/ice_therm_vertical.F90:45-49,64-65
constexpr double saltmax = 3.2, hs_min = 1.0e-4, betak = 0.13, kimin = 0.10, ferrmax = 1.0e-3;
}  // namespace K

constexpr int NCAT = CICE_NCAT, NILYR = CICE_NILYR, NSLYR = CICE_NSLYR, NTRCR = CICE_MAX_NTRCR;

struct Error {
  int code;
  std::string msg;
};

#define CICE_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      throw ::cice::Error{CICE_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; \
  } while (0)

#define CICE_REQUIRE(cond, text) \
  do {                           \
    if (!(cond)) throw ::cice::Error{CICE_EINVAL, text}; \
  } while (0)

// RAII device buffer
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;   // false: a view into another buffer's allocation
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owned = true;
  }
  void view(T* ptr, size_t count) {   // alloc(count) afterwards is a no-op
    release();
    p = ptr;
    n = count;
    owned = false;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    if (count == 0) return;
    CICE_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
    fine = false;
  }
  // FINE-GRAINED device memory: what ANOTHER device writes or polls while a kernel of this one runs (exchange copies and
  // progress words of the cross-rank one-launch loop).  Ordinary (coarse-grained) hipMalloc memory carries no cross-device
  // coherence guarantee before the end of a kernel; fine-grained memory does, for system-scope accesses (RCCL allocates
  // its own flags this way for the same reason).
  bool fine = false;
  void alloc_fine(size_t count) {
    if (count == n && p && fine) return;
    release();
    if (count == 0) return;
    CICE_HIP(hipExtMallocWithFlags((void**)&p, count * sizeof(T), hipDeviceMallocFinegrained));
    n = count;
    fine = true;
  }
  void zero(hipStream_t s) {
    if (p) CICE_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  void upload(const T* h, hipStream_t s) {
    CICE_HIP(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* h, hipStream_t s) const {
    CICE_HIP(hipMemcpyAsync(h, p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  }
};

// Host <-> device copies of MANY separate arrays.  Consecutive copies on one stream leave about 10 us between them
// (each waits for the completion signal of the one before): 1 MB planes move at 30 GB/s where the link does 55.
// Copies on several streams overlap.  fork(main): the side streams wait for everything enqueued on `main` so far;
// next(): the stream for the next copy, round robin; join(): `main` waits for every side stream.
// CICE4_AMD_COPY_STREAMS in the environment sets their number (1: next() is `main` itself).  Measured at gx1 size
// (scripts/gpu_r3_fan.sh): evp(dt) over PCIe 2.88 / 2.52 ms with 1 / 2, cice_step_therm1 7.21 / 6.57 ms; 3 and 4 are
// no faster and, in a process that holds other streams (torch), many times slower once the runtime's hardware queues
// are oversubscribed -- hence 2.
struct CopyFan {
  static constexpr int NMAX = 4;
  hipStream_t side[NMAX] = {};
  hipEvent_t ev[NMAX + 1] = {};
  int n = 0, k = 0;
  bool forked = false;
  hipStream_t main = nullptr;
  CopyFan() = default;
  CopyFan(const CopyFan&) = delete;
  CopyFan& operator=(const CopyFan&) = delete;
  ~CopyFan() {
    for (int i = 0; i < NMAX; ++i)
      if (side[i]) (void)hipStreamDestroy(side[i]);
    for (int i = 0; i <= NMAX; ++i)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
  }
  static int wanted() {
    static const int w = [] {
      const char* e = std::getenv("CICE4_AMD_COPY_STREAMS");
      const int v = e ? std::atoi(e) : 2;
      return v < 1 ? 1 : (v > NMAX ? NMAX : v);
    }();
    return w;
  }
  void fork(hipStream_t m) {
    main = m;
    k = 0;
    n = wanted();
    forked = n > 1;
    if (!forked) return;
    for (int i = 0; i < n; ++i)
      if (!side[i]) CICE_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    for (int i = 0; i <= n; ++i)
      if (!ev[i]) CICE_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    CICE_HIP(hipEventRecord(ev[n], main));
    for (int i = 0; i < n; ++i) CICE_HIP(hipStreamWaitEvent(side[i], ev[n], 0));
  }
  hipStream_t next() { return forked ? side[k++ % n] : main; }
  // leave the copies enqueued since fork() in flight: `main` does NOT wait for them (a later fork() / join() pair does --
  // the side streams run in order)
  void detach() { forked = false; }
  void join() {
    if (!forked) return;
    for (int i = 0; i < n; ++i) {
      CICE_HIP(hipEventRecord(ev[i], side[i]));
      CICE_HIP(hipStreamWaitEvent(main, ev[i], 0));
    }
    forked = false;
  }
};

}  // namespace cice
