// Horizontal transport by incremental remapping on the device (source/ice_transport_driver.F90:179-663
// transport_remap, source/ice_transport_remap.F90:328-881 horizontal_remap) -- SURVEY.md section 8(f3).
#pragma once
#include "common.h"
#include "domain.h"
#include "halo.h"

namespace cice {

constexpr int TR_MAXTRACE = 2 + NTRCR + NILYR + NSLYR;  // hice, hsno, trcr, qice, qsno (max_ntrace, :40)

struct TransportKernelArgs {
  int nx, ny, nb, ntrace, ntrcr;
  int type[TR_MAXTRACE], dep[TR_MAXTRACE], hasdep[TR_MAXTRACE];  // tracer_type, depend (0-based, -1 none), has_dependents
  double dt;
  const int32_t* blk;  // ilo, ihi, jlo, jhi per block (1-based)
  const double *HTN, *HTE, *dxt, *dyt, *dxu, *dyu, *tarear, *hm;
  // state, reference layout
  double *aice0, *aicen, *trcrn, *vicen, *vsnon, *eicen, *esnon;
  const double *uvel, *vvel;
  // work: mm (nx,ny,0:ncat,nb); tm (nx,ny,ntrace,ncat,nb); masks alike
  double *mm, *tm, *mmask, *tmask;
  double *mc, *mx, *my;     // (nx,ny,0:ncat,nb) each; mc directly followed by tc (one scalar halo update)
  double *tc, *tx, *ty;     // (nx,ny,ntrace,ncat,nb)
  double *dpx, *dpy;        // (nx,ny,nb)
  double *mflx, *mtflx;     // [dir 0 east / 1 north]: (nx,ny,0:ncat,nb), (nx,ny,ntrace,ncat,nb)
  unsigned long long* errkey;
};

class Transport {
 public:
  Transport(const Domain& d, Halo& h, hipStream_t s, CopyFan& f) : dom(d), halo(h), stream(s), fan(f) {}
  void init(const cice_transport_config& c, const cice_transport_grid& g);
  // one transport_remap(dt) on host arrays (upload, remap, bound_state on the device, download)
  void remap(double dt, const cice_transport_fields& f, int32_t* l_stop, int32_t* istop, int32_t* jstop);
  // evp -> transport chain (cice_transport_chain): prefetch() starts the upload of aice0, trcrn, vsnon, eicen, esnon on the
  // copy streams and returns at once; adopt() copies uvel | vvel (one buffer) and aicen, vicen (host layout) from the
  // dynamics' device buffers; the next remap() then uploads nothing
  void prefetch(const cice_transport_fields& f);
  void adopt(const double* d_uv, const double* d_aicen, const double* d_vicen);
  bool chained = false;   // the state of the next remap() is on the device already
  // test aid: stop the next remap() after kernel stage `s` (1 tracers, 2 fields + departure points + their halos,
  // 3 fluxes, 4 update; 0 = run through) and copy a work array to the host (0 mm, 1 tm, 2 mc|tc, 3 mx|tx|my|ty,
  // 4 dpx|dpy, 5 mflx, 6 mtflx, 7 mmask, 8 tmask); returns the number of doubles
  void debug_stop(int s) { stop_stage = s; }
  size_t debug_fetch(int which, double* out);
  int stop_stage = 0;

 private:
  const Domain& dom;
  Halo& halo;
  hipStream_t stream;
  CopyFan& fan;  // the context's side streams for the state upload / download
  size_t n = 0;  // nblocks * nx_block * ny_block
  int ntrace = 0, ntrcr = 0;
  void up(double* d, const double* h, int levels);   // host (nx,ny,levels,nblocks) -> device (nx,ny,nblocks) per level
  TransportKernelArgs a{};
  DevBuf<int32_t> blk;
  DevBuf<double> HTN, HTE, dxt, dyt, dxu, dyu, tarear, hm;
  DevBuf<double> aice0, aicen, trcrn, vicen, vsnon, eicen, esnon, uv;
  DevBuf<double> mm, tm, mmask, tmask, ctr, grad, dp, mflx, mtflx;
  DevBuf<unsigned long long> key;
};

// advection = 'upwind' (source/ice_transport_driver.F90:672-834 transport_upwind, :1570-1878 state_to_work,
// work_to_state, upwind_field; compute_tracers source/ice_itd.F90:1482-1590): first-order donor-cell transport of
// aice0, the state of every category (area, volumes, tracer x area / volume) and the layer enthalpies.
struct UpwindArgs {
  int nx, ny, nb, ntrcr, narr, nlev;   // narr = 1 + ncat*(3+ntrcr) work planes, nlev = narr + ncat*(nilyr+nslyr)
  int dep[NTRCR], it_Tsfc;
  double dt;
  const int32_t* blk;
  const double *HTE, *HTN, *tarea, *uvel, *vvel;
  double *uee, *vnn;
  double *aice0, *aicen, *trcrn, *vicen, *vsnon, *eicen, *esnon;
  double *phi, *phi2;   // (nx,ny,nb) per level: the fields before / after upwind_field
};

class Upwind {
 public:
  Upwind(const Domain& d, Halo& h, hipStream_t s, CopyFan& f) : dom(d), halo(h), stream(s), fan(f) {}
  void init(const cice_transport_config& c, int nt_Tsfc, const double* HTE, const double* HTN, const double* tarea);
  void step(double dt, const cice_transport_fields& f);   // one transport_upwind(dt) on host arrays

 private:
  const Domain& dom;
  Halo& halo;
  hipStream_t stream;
  CopyFan& fan;
  size_t n = 0;
  UpwindArgs a{};
  DevBuf<int32_t> blk;
  DevBuf<double> HTE, HTN, tarea, uv, edge, aice0, aicen, trcrn, vicen, vsnon, eicen, esnon, phi, phi2;
};

}  // namespace cice
