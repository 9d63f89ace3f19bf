// atmo_boundary_layer (source/ice_atmo.F90:56-384) for gfx950.  One lane = one (cell[, category]); the five
// stability iterations (:264-305) run in registers.  exp is glibc's (libm_exact.h); log and atan are the device
// library's (<= 1 ulp): the outputs agree with the reference's to ~1e-15 relative, not bit for bit -- which is why
// the bit-exact whole-driver configuration keeps the reference's host routine.
#include <cmath>

#include "atmo.h"

namespace cice {

void AtmoParams::init() {
  vonkar = 0.4; gravit = 9.80616; zvir = 0.606; cp_air = 1005.0; cpvir = 1.81e3 / 1005.0 - 1.0;
  Tffresh = 273.15; pih = 0.5 * 3.14159265358979323846; zTrf = 2.0; umin = 1.0; zref = 10.0;
  qqq[0] = 11637800.0; TTT[0] = 5897.8; Lheat[0] = 2.835e6;
  qqq[1] = 627572.4; TTT[1] = 5107.4; Lheat[1] = 2.501e6;
  rdn_ice = vonkar / std::log(zref / 0.0005);
  al2 = std::log(zref / zTrf);
}

namespace {

struct AtmoOut { double strx, stry, Tref, Qref, delt, delq, lhcoef, shcoef; };

__device__ __forceinline__ double psimhu(double xd, double pih) {   // :165-166
  return log((1.0 + xd * (2.0 + xd)) * (1.0 + xd * xd) / 8.0) - 2.0 * atan(xd) + pih;
}
__device__ __forceinline__ double psixhu(double xd) {               // :169
  return 2.0 * log((1.0 + xd * xd) / 2.0);
}

__device__ __forceinline__ AtmoOut atmo_cell(const AtmoParams& P, int ocn, double Tsf, double potT, double uatm,
                                             double vatm, double wind, double zlvl, double Qa, double rhoa) {
  AtmoOut o;
  const double vmag = fmax(P.umin, wind);                                             // :201 / :214
  const double rdn = ocn ? sqrt(0.0027 / vmag + .000142 + .0000764 * vmag) : P.rdn_ice;  // :202 / :215
  const double TsfK = Tsf + P.Tffresh;                                                // :232
  const double qsat = P.qqq[ocn] * exp_libm(-P.TTT[ocn] / TsfK);
  const double ssq = qsat / rhoa;
  const double thva = potT * (1.0 + P.zvir * Qa);
  const double delt = potT - TsfK, delq = Qa - ssq;
  const double alz = log(zlvl / P.zref);
  const double cp = P.cp_air * (1.0 + P.cpvir * ssq);
  const double rhn = rdn, ren = rdn;                                                  // :248-249
  double ustar = rdn * vmag, tstar = rhn * delt, qstar = ren * delq;
  double hol = 0.0, stable = 0.0, psixh = 0.0, rd = 0.0, rh = 0.0, re = 0.0;
  for (int k = 0; k < 5; ++k) {                                                        // :264
    hol = P.vonkar * P.gravit * zlvl * (tstar / thva + qstar / (1.0 / P.zvir + Qa)) / (ustar * ustar);
    hol = copysign(fmin(fabs(hol), 10.0), hol);
    stable = 0.5 + copysign(0.5, hol);
    double xqq = fmax(sqrt(fabs(1.0 - 16.0 * hol)), 1.0);
    xqq = sqrt(xqq);
    const double psimhs = -(0.7 * hol + 0.75 * (hol - 14.3) * exp_libm(-0.35 * hol) + 10.7);   // Jordan et al 1999
    const double psimh = psimhs * stable + (1.0 - stable) * psimhu(xqq, P.pih);
    psixh = psimhs * stable + (1.0 - stable) * psixhu(xqq);
    rd = rdn / (1.0 + rdn / P.vonkar * (alz - psimh));                                 // :291-293
    rh = rhn / (1.0 + rhn / P.vonkar * (alz - psixh));
    re = ren / (1.0 + ren / P.vonkar * (alz - psixh));
    ustar = rd * vmag; tstar = rh * delt; qstar = re * delq;
  }
  const double tau = rhoa * ustar * rd;                                               // :335
  o.strx = tau * uatm; o.stry = tau * vatm;
  o.shcoef = rhoa * ustar * cp * rh + 1.0;                                            // :359-360
  o.lhcoef = rhoa * ustar * P.Lheat[ocn] * re;
  hol = hol * P.zTrf / zlvl;                                                          // :365
  double xqq = fmax(1.0, sqrt(fabs(1.0 - 16.0 * hol)));
  xqq = sqrt(xqq);
  const double psix2 = -5.0 * hol * stable + (1.0 - stable) * psixhu(xqq);
  double fac = (rh / P.vonkar) * (alz + P.al2 - psixh + psix2);
  o.Tref = potT - delt * fac;
  o.Tref = o.Tref - 0.01 * P.zTrf;
  fac = (re / P.vonkar) * (alz + P.al2 - psixh + psix2);
  o.Qref = Qa - delq * fac;
  o.delt = delt; o.delq = delq;
  return o;
}

// one block, the reference's list: everything is zeroed first (:179-188, :313-318), then the listed cells
__global__ __launch_bounds__(256) void k_atmo_zero(AtmoArgs a) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (size_t)a.nx * a.ny) return;
  a.Tref[q] = 0.0; a.Qref[q] = 0.0; a.delt[q] = 0.0; a.delq[q] = 0.0; a.shcoef[q] = 0.0; a.lhcoef[q] = 0.0;
  if (a.calc_strair) { a.strx[q] = 0.0; a.stry[q] = 0.0; }
}
__global__ __launch_bounds__(256) void k_atmo_list(AtmoArgs a) {
  const int ij = blockIdx.x * 256 + threadIdx.x;
  if (ij >= a.icells) return;
  const size_t q = (size_t)(a.indxj[ij] - 1) * a.nx + (a.indxi[ij] - 1);
  const AtmoOut o = atmo_cell(a.p, a.ocn, a.Tsf[q], a.potT[q], a.uatm[q], a.vatm[q], a.wind[q], a.zlvl[q], a.Qa[q],
                              a.rhoa[q]);
  if (a.calc_strair) { a.strx[q] = o.strx; a.stry[q] = o.stry; }
  a.Tref[q] = o.Tref; a.Qref[q] = o.Qref; a.delt[q] = o.delt; a.delq[q] = o.delq;
  a.lhcoef[q] = o.lhcoef; a.shcoef[q] = o.shcoef;
}

// every category of every block; blockIdx.y = b * ncat + n
__global__ __launch_bounds__(256) void k_atmo_dense(AtmoArgs a) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= np) return;
  const int bn = blockIdx.y, b = bn / a.ncat;
  const int i = (int)(q % a.nx) + 1, j = (int)(q / a.nx) + 1;
  const int32_t* bl = a.blk + 4 * b;
  const size_t c = (size_t)bn * np + q, f = (size_t)b * np + q;
  const bool in = i >= bl[0] && i <= bl[1] && j >= bl[2] && j <= bl[3] && a.aicen[c] > K::puny;
  AtmoOut o{};
  if (in)
    o = atmo_cell(a.p, 0, a.Tsf[((size_t)bn * NTRCR + a.it_Tsfc) * np + q], a.potT[f], a.uatm[f], a.vatm[f],
                  a.wind[f], a.zlvl[f], a.Qa[f], a.rhoa[f]);
  if (a.calc_strair) { a.strx[c] = o.strx; a.stry[c] = o.stry; }
  else { a.strx[c] = a.strax[f]; a.stry[c] = a.stray[f]; }
  a.Tref[c] = o.Tref; a.Qref[c] = o.Qref; a.lhcoef[c] = o.lhcoef; a.shcoef[c] = o.shcoef;
  if (a.delt) a.delt[c] = o.delt;
  if (a.delq) a.delq[c] = o.delq;
}

}  // namespace

void atmo_launch_list(const AtmoArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  k_atmo_zero<<<dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s>>>(a);
  if (a.icells > 0) k_atmo_list<<<dim3((unsigned)((a.icells + 255) / 256)), dim3(256), 0, s>>>(a);
  CICE_HIP(hipGetLastError());
}

void atmo_launch_dense(const AtmoArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  k_atmo_dense<<<dim3((unsigned)((np + 255) / 256), (unsigned)(a.ncat * a.nblocks)), dim3(256), 0, s>>>(a);
  CICE_HIP(hipGetLastError());
}

}  // namespace cice
