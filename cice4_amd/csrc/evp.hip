// EVP dynamics on the device.  Behavioural source: source/ice_dyn_evp.F90 of the
// reference (file:line cited per kernel).  Layout: every field is the reference's
// (nx_block,ny_block,nblocks) array, i fastest, so a wavefront reads 64 consecutive
// i of one row = 512 contiguous bytes.
//
// The hot loop (ice_dyn_evp.F90:347-404: stress, stepu, two halo updates, ndte times) is one
// kernel per subcycle (k_subcycle) or -- where no ghost row of a local block changes between two
// subcycles -- one kernel per PAIR of subcycles (k_subcycle2).  A wavefront keeps a row of T-cells,
// a lane one column; the 8 `str` combinations of a cell live in registers and reach the momentum
// equation of the neighbouring U-cells by wavefront shuffle (i+1) and LDS (j+1), so they never
// touch HBM and the reference's `str(:,:,:) = 0` memset disappears with them.  u, v and the
// stresses are double-buffered so that tiles can recompute their rim without racing with the
// owner's update; the arithmetic per cell is exactly the reference's, in its order, compiled
// without FMA contraction.
#include "evp.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>

namespace cice {

using namespace K;

#ifdef CICE4_AMD_AUSCOM
// Namelist variables of the coupled build (ice_dyn_evp.F90:91-97; defaults of ice_init.F90:258-264); dragw is formed
// as stepu forms it (:1382).  Device constants: every kernel below names them as the stand-alone build names its
// compile-time constants.
__constant__ double cosw = 1.0, sinw = 0.0, dragw = dragio * rhow;
__constant__ int ocnslope = 0;

void evp_set_namelist(double cosw_, double sinw_, double dragio_, int use_ocnslope) {
  const double dragw_ = dragio_ * rhow;
  const int os = use_ocnslope != 0;
  CICE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cosw), &cosw_, 8));
  CICE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sinw), &sinw_, 8));
  CICE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dragw), &dragw_, 8));
  CICE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ocnslope), &os, 4));
}
#endif

void EvpScalars::set(double dt, int ndte_, int damping) {
  ndte = ndte_;
  evp_damping = damping;
  double dte = dt / (double)ndte;
  dtei = c1 / dte;
  double ecc = c4;
  ecci = p25;
  double tdamp2 = c2 * eyc * dt;
  dte2T = dte / tdamp2;
  denom1 = c1 / (c1 + dte2T);
  denom2 = c1 / (c1 + dte2T * ecc);
  rcon = 1230.0 * eyc * dt * (dtei * dtei);
}

namespace {

struct StressOut {
  double str[8];
  double divu, rdg_conv, rdg_shear, shear, prs_sig;
};

// One T-cell of `stress` (ice_dyn_evp.F90:1065-1289).  s[12] = stressp_1..4, stressm_1..4,
// stress12_1..4, updated in place.
template <bool LAST, bool DAMP>
__device__ __forceinline__ void stress_cell(const EvpScalars& sc, double u_ne, double u_nw,
                                            double u_sw, double u_se, double v_ne, double v_nw,
                                            double v_sw, double v_se, double Dxt, double Dyt,
                                            double Dxhy, double Dyhx, double Cxp, double Cyp,
                                            double Cxm, double Cym, double Tarear, double Tiny,
                                            double St, double* s, StressOut& o, const bool last_rt = true) {
  // :1065-1092 strain rates * area
  const double divune = Cyp * u_ne - Dyt * u_nw + Cxp * v_ne - Dxt * v_se;
  const double divunw = Cym * u_nw + Dyt * u_ne + Cxp * v_nw - Dxt * v_sw;
  const double divusw = Cym * u_sw + Dyt * u_se + Cxm * v_sw + Dxt * v_nw;
  const double divuse = Cyp * u_se - Dyt * u_sw + Cxm * v_se + Dxt * v_ne;
  const double tensionne = -Cym * u_ne - Dyt * u_nw + Cxm * v_ne + Dxt * v_se;
  const double tensionnw = -Cyp * u_nw + Dyt * u_ne + Cxm * v_nw + Dxt * v_sw;
  const double tensionsw = -Cyp * u_sw + Dyt * u_se + Cxp * v_sw - Dxt * v_nw;
  const double tensionse = -Cym * u_se - Dyt * u_sw + Cxp * v_se - Dxt * v_ne;
  const double shearne = -Cym * v_ne - Dyt * v_nw - Cxm * u_ne - Dxt * u_se;
  const double shearnw = -Cyp * v_nw + Dyt * v_ne - Cxm * u_nw - Dxt * u_sw;
  const double shearsw = -Cyp * v_sw + Dyt * v_se - Cxp * u_sw + Dxt * u_nw;
  const double shearse = -Cym * v_se - Dyt * v_sw - Cxp * u_se + Dxt * u_ne;
  // :1095-1098
  const double ecci = sc.ecci;
  const double Deltane = sqrt(divune * divune + ecci * (tensionne * tensionne + shearne * shearne));
  const double Deltanw = sqrt(divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw));
  const double Deltase = sqrt(divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse));
  const double Deltasw = sqrt(divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw));
  if (LAST && last_rt) {  // :1103-1115
    o.divu = p25 * (divune + divunw + divuse + divusw) * Tarear;
    const double tmp = p25 * (Deltane + Deltanw + Deltase + Deltasw) * Tarear;
    o.rdg_conv = -fmin(o.divu, c0);
    o.rdg_shear = p5 * (tmp - fabs(o.divu));
    const double ts = tensionne + tensionnw + tensionse + tensionsw;
    const double ss = shearne + shearnw + shearse + shearsw;
    o.shear = p25 * Tarear * sqrt(ts * ts + ss * ss);
  }
  double c0ne, c0nw, c0sw, c0se;
  if (DAMP) {  // :1121-1128
    c0ne = fmin(St / fmax(Deltane, c4 * Tiny), sc.rcon);
    c0nw = fmin(St / fmax(Deltanw, c4 * Tiny), sc.rcon);
    c0sw = fmin(St / fmax(Deltasw, c4 * Tiny), sc.rcon);
    c0se = fmin(St / fmax(Deltase, c4 * Tiny), sc.rcon);
    o.prs_sig = St * Deltane / fmax(Deltane, c4 * Tiny);
  } else {  // :1131-1135
    c0ne = St / fmax(Deltane, Tiny);
    c0nw = St / fmax(Deltanw, Tiny);
    c0sw = St / fmax(Deltasw, Tiny);
    c0se = St / fmax(Deltase, Tiny);
    o.prs_sig = c0ne * Deltane;
  }
  const double dte2T = sc.dte2T, denom1 = sc.denom1, denom2 = sc.denom2;
  const double c1ne = c0ne * dte2T, c1nw = c0nw * dte2T, c1sw = c0sw * dte2T, c1se = c0se * dte2T;
  // :1148-1165
  const double p1 = s[0] = (s[0] + c1ne * (divune - Deltane)) * denom1;
  const double p2 = s[1] = (s[1] + c1nw * (divunw - Deltanw)) * denom1;
  const double p3 = s[2] = (s[2] + c1sw * (divusw - Deltasw)) * denom1;
  const double p4 = s[3] = (s[3] + c1se * (divuse - Deltase)) * denom1;
  const double m1 = s[4] = (s[4] + c1ne * tensionne) * denom2;
  const double m2 = s[5] = (s[5] + c1nw * tensionnw) * denom2;
  const double m3 = s[6] = (s[6] + c1sw * tensionsw) * denom2;
  const double m4 = s[7] = (s[7] + c1se * tensionse) * denom2;
  const double t1 = s[8] = (s[8] + c1ne * shearne * p5) * denom2;
  const double t2 = s[9] = (s[9] + c1nw * shearnw * p5) * denom2;
  const double t3 = s[10] = (s[10] + c1sw * shearsw * p5) * denom2;
  const double t4 = s[11] = (s[11] + c1se * shearse * p5) * denom2;
  // :1196-1239
  const double ssigpn = p1 + p2, ssigps = p3 + p4, ssigpe = p1 + p4, ssigpw = p2 + p3;
  const double ssigp1 = (p1 + p3) * p055, ssigp2 = (p2 + p4) * p055;
  const double ssigmn = m1 + m2, ssigms = m3 + m4, ssigme = m1 + m4, ssigmw = m2 + m3;
  const double ssigm1 = (m1 + m3) * p055, ssigm2 = (m2 + m4) * p055;
  const double ssig12n = t1 + t2, ssig12s = t3 + t4, ssig12e = t1 + t4, ssig12w = t2 + t3;
  const double ssig121 = (t1 + t3) * p111, ssig122 = (t2 + t4) * p111;
  const double csigpne = p111 * p1 + ssigp2 + p027 * p3;
  const double csigpnw = p111 * p2 + ssigp1 + p027 * p4;
  const double csigpsw = p111 * p3 + ssigp2 + p027 * p1;
  const double csigpse = p111 * p4 + ssigp1 + p027 * p2;
  const double csigmne = p111 * m1 + ssigm2 + p027 * m3;
  const double csigmnw = p111 * m2 + ssigm1 + p027 * m4;
  const double csigmsw = p111 * m3 + ssigm2 + p027 * m1;
  const double csigmse = p111 * m4 + ssigm1 + p027 * m2;
  const double csig12ne = p222 * t1 + ssig122 + p055 * t3;
  const double csig12nw = p222 * t2 + ssig121 + p055 * t4;
  const double csig12sw = p222 * t3 + ssig122 + p055 * t1;
  const double csig12se = p222 * t4 + ssig121 + p055 * t2;
  const double str12ew = p5 * Dxt * (p333 * ssig12e + p166 * ssig12w);
  const double str12we = p5 * Dxt * (p333 * ssig12w + p166 * ssig12e);
  const double str12ns = p5 * Dyt * (p333 * ssig12n + p166 * ssig12s);
  const double str12sn = p5 * Dyt * (p333 * ssig12s + p166 * ssig12n);
  // :1244-1289
  double strp_tmp = p25 * Dyt * (p333 * ssigpn + p166 * ssigps);
  double strm_tmp = p25 * Dyt * (p333 * ssigmn + p166 * ssigms);
  o.str[0] = -strp_tmp - strm_tmp - str12ew + Dxhy * (-csigpne + csigmne) + Dyhx * csig12ne;
  o.str[1] = strp_tmp + strm_tmp - str12we + Dxhy * (-csigpnw + csigmnw) + Dyhx * csig12nw;
  strp_tmp = p25 * Dyt * (p333 * ssigps + p166 * ssigpn);
  strm_tmp = p25 * Dyt * (p333 * ssigms + p166 * ssigmn);
  o.str[2] = -strp_tmp - strm_tmp + str12ew + Dxhy * (-csigpse + csigmse) + Dyhx * csig12se;
  o.str[3] = strp_tmp + strm_tmp + str12we + Dxhy * (-csigpsw + csigmsw) + Dyhx * csig12sw;
  strp_tmp = p25 * Dxt * (p333 * ssigpe + p166 * ssigpw);
  strm_tmp = p25 * Dxt * (p333 * ssigme + p166 * ssigmw);
  o.str[4] = -strp_tmp + strm_tmp - str12ns - Dyhx * (csigpne + csigmne) + Dxhy * csig12ne;
  o.str[5] = strp_tmp - strm_tmp - str12sn - Dyhx * (csigpse + csigmse) + Dxhy * csig12se;
  strp_tmp = p25 * Dxt * (p333 * ssigpw + p166 * ssigpe);
  strm_tmp = p25 * Dxt * (p333 * ssigmw + p166 * ssigme);
  o.str[6] = -strp_tmp + strm_tmp + str12ns - Dyhx * (csigpnw + csigmnw) + Dxhy * csig12nw;
  o.str[7] = strp_tmp - strm_tmp + str12sn - Dyhx * (csigpsw + csigmsw) + Dxhy * csig12sw;
}

// Ocean stress direction, evp_prep2's expressions (ice_dyn_evp.F90:909-917).  The coupled flavour turns with the
// hemisphere: sign(1., real(fm)) is the sign BIT of fm (the conversion to single keeps it, zeros included).
__device__ __forceinline__ void water_of(double uo, double vo, double fmv, double& wx, double& wy) {
#ifdef CICE4_AMD_AUSCOM
  const double sg = __builtin_signbit(fmv) ? -1.0 : 1.0;
  wx = uo * cosw - vo * sinw * sg;
  wy = vo * cosw + uo * sinw * sg;
#else
  (void)fmv;
  wx = uo * cosw - vo * sinw;
  wy = vo * cosw + uo * sinw;
#endif
}

struct StepuOut {
  double u, v, strintx, strinty, taux, tauy;
};

// One U-cell of `stepu` (ice_dyn_evp.F90:1390-1435); sx/sy are the four-term sums of :1415-1418.
__device__ __forceinline__ void stepu_cell(double uold, double vold, double Aiu, double Uocn,
                                           double Vocn, double Waterx, double Watery, double Forcex,
                                           double Forcey, double Umassdtei, double Fm, double Uarear,
                                           double sx, double sy, StepuOut& o) {
  const double du = Uocn - uold, dv = Vocn - vold;
  const double vrel = Aiu * dragw * sqrt(du * du + dv * dv);
  o.taux = vrel * Waterx;
  o.tauy = vrel * Watery;
  const double cca = Umassdtei + vrel * cosw;
#ifdef CICE4_AMD_AUSCOM   // :1402-1408: the turning angle changes sign with the hemisphere
  const double ccb = Fm < 0.0 ? Fm - vrel * sinw : Fm + vrel * sinw;
#else
  const double ccb = Fm + vrel * sinw;
#endif
  const double ab2 = cca * cca + ccb * ccb;
  o.strintx = Uarear * sx;
  o.strinty = Uarear * sy;
  const double cc1 = o.strintx + Forcex + o.taux + Umassdtei * uold;
  const double cc2 = o.strinty + Forcey + o.tauy + Umassdtei * vold;
  o.u = (cca * cc1 + ccb * cc2) / ab2;
  o.v = (cca * cc2 - ccb * cc1) / ab2;
}

}  // namespace

struct SubArgs {
  EvpScalars sc;
  int nx, ny, tiles_x, tiles_y, nblocks;
  int diag_jmin;   // rows below it keep their diagnostics (a band of rows computed beside a sweep: its lower rows are scratch)
  int carry_top;   // tripole fold: the fold changes top-row cells WITHOUT ice too; the kernel carries their value into the new copy
  int ew_cyclic;  // k_subcycle2: columns of a block form a ring
  size_t n;  // nblocks*ny*nx
  // no two of these arrays overlap (inputs and outputs of the double-buffered fields are
  // different allocations), which lets the compiler issue loads ahead of stores
  const int32_t* __restrict__ ring_slot;  // on-rank ghost forwarding (Halo)
  const int32_t* __restrict__ fwd;
  const int32_t* __restrict__ blk;
  const int32_t* __restrict__ icetmask;
  const int32_t* __restrict__ iceumask;
  const double* __restrict__ u_in;
  const double* __restrict__ v_in;
  double* __restrict__ u_out;
  double* __restrict__ v_out;
  const double* __restrict__ sig_in;
  double* __restrict__ sig_out;
  const double* sig_in_p[12];   // the 12 planes of sig_in / sig_out as separate uniform pointers
  double* sig_out_p[12];        // (k_subcycle2: SGPR base + lane offset addressing)
  const double *__restrict__ dxt, *__restrict__ dyt, *__restrict__ dxhy, *__restrict__ dyhx,
      *__restrict__ cxp, *__restrict__ cyp, *__restrict__ cxm, *__restrict__ cym,
      *__restrict__ tarear, *__restrict__ tinyarea, *__restrict__ strength;
  const double *__restrict__ HTN, *__restrict__ HTE;  // DERIVE: the 9 T-cell metrics recomputed from these
  const double *__restrict__ aiu, *__restrict__ uocn, *__restrict__ vocn, *__restrict__ waterx,
      *__restrict__ watery, *__restrict__ forcex, *__restrict__ forcey, *__restrict__ umassdtei,
      *__restrict__ fm, *__restrict__ uarear;
  double *__restrict__ divu, *__restrict__ rdg_conv, *__restrict__ rdg_shear, *__restrict__ shear,
      *__restrict__ prs_sig, *__restrict__ strintx, *__restrict__ strinty, *__restrict__ strocnx,
      *__restrict__ strocny;
};

namespace {


// Neighbour-lane exchange without the LDS crossbar: DPP wave shifts (gfx9 family; checked against
// __shfl_up/__shfl_down on gfx950 by scripts/probe/dpp_wave_shift_probe.hip).  Lane 0 of up1 and
// lane 63 of down1 keep their own value, exactly like __shfl_up(x, 1) / __shfl_down(x, 1).
__device__ __forceinline__ double up1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double down1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

#ifndef SKEW_OPAQUE_STRIDE
#define SKEW_OPAQUE_STRIDE 1
#endif
#ifndef SKEW_TRIM        // every level computes only the rows its successors need (see the kernel)
#define SKEW_TRIM 1
#endif
#ifndef SKEW_LOADPRIO    // highest issue priority while a step's loads are being issued, the rotation afterwards
#define SKEW_LOADPRIO 1  // (0.1 degree: 278-283 us per subcycle against 281-285, A/B of round 4)
#endif
#ifndef SKEW_TPASS       // HTN, HTE, strength and the two masks of a row travel from level to level through LDS (K <= 4)
#define SKEW_TPASS 1     // (measured traffic 5.67 -> 4.48 GB per launch at 0.1 degree, time -0.3 %: profiles/r04_sweep_least_traffic_configuration.txt)
#endif
#ifndef SKEW_WIDE        // the read-only inputs of a step come interleaved: 16-byte loads, 7 instead of 14 per level and step
#define SKEW_WIDE 1
#endif
// Variants that were built, are bit-exact, measured SLOWER and are therefore not part of the product build: compile with
// -DCICE4_AMD_EXPERIMENTS to get them back (scripts/build_ab.sh; the numbers are in DESIGN.md section 3.2 / HISTORY.md) --
// three wavefronts per level in one 12-wavefront workgroup (S = 3), K = 5, 6, 8, two wavefronts per SIMD at K = 4, and
// SKEW_EARLY: a third slot for the hand-off of level 0, its stresses of the NEXT row fetched before the barrier.
#if !defined(CICE4_AMD_EXPERIMENTS) || !defined(SKEW_EARLY)
#undef SKEW_EARLY
#define SKEW_EARLY 0
#endif
// The same shifts with bound_ctrl: the lane without a source (lane 0 / lane 63) reads zero instead of keeping its own
// value, which frees the compiler from copying the operand first (one instruction per half instead of two).  For
// kernels in which that lane's result is never used (k_subcycle_skew: lanes 0 and 63 own nothing).
__device__ __forceinline__ double up1z(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double down1z(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

constexpr int TX = 64;  // T-cells per tile row = one wavefront

// Momentum update of one U-cell inside the fused kernel + forwarding of the new velocity to
// the ghost cells that mirror this cell on this rank (the on-rank part of the two
// ice_HaloUpdate calls at ice_dyn_evp.F90:397-402, folded into the producing kernel).
struct UIn {  // the ten read-only U-cell fields of stepu (ice_dyn_evp.F90:1339-1349)
  double aiu, uocn, vocn, waterx, watery, forcex, forcey, umassdtei, fm, uarear;
};

template <bool DERIVE>
__device__ __forceinline__ void load_uin(const SubArgs& a, size_t q, UIn& x) {
  x.aiu = a.aiu[q]; x.uocn = a.uocn[q]; x.vocn = a.vocn[q];
  if (!DERIVE) {
    x.waterx = a.waterx[q];
    x.watery = a.watery[q];
  }
  x.forcex = a.forcex[q]; x.forcey = a.forcey[q];
  x.umassdtei = a.umassdtei[q]; x.fm = a.fm[q]; x.uarear = a.uarear[q];
  if (DERIVE) water_of(x.uocn, x.vocn, x.fm, x.waterx, x.watery);   // instead of two loads
}

// Momentum update of one U-cell inside the fused kernel + forwarding of the new velocity to
// the ghost cells that mirror this cell on this rank (the on-rank part of the two
// ice_HaloUpdate calls at ice_dyn_evp.F90:397-402, folded into the producing kernel).
template <bool LAST>
__device__ __forceinline__ void stepu_store(const SubArgs& a, const UIn& x, size_t q, int i, int j,
                                            int ilo, int ihi, int jlo, int jhi, double uold,
                                            double vold, double sx, double sy) {
  StepuOut r;
  stepu_cell(uold, vold, x.aiu, x.uocn, x.vocn, x.waterx, x.watery, x.forcex, x.forcey, x.umassdtei,
             x.fm, x.uarear, sx, sy, r);
  a.u_out[q] = r.u;
  a.v_out[q] = r.v;
  if (LAST && j >= a.diag_jmin) {
    a.strintx[q] = r.strintx;
    a.strinty[q] = r.strinty;
    a.strocnx[q] = r.taux;
    a.strocny[q] = r.tauy;
  }
  if (a.ring_slot && (i == ilo || i == ihi || j == jlo || j == jhi)) {
    const int slot = a.ring_slot[q];
    if (slot >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int d = a.fwd[3 * slot + k];
        if (d >= 0) {
          a.u_out[d] = r.u;
          a.v_out[d] = r.v;
        }
      }
    }
  }
}

// One EVP subcycle, fused (ice_dyn_evp.F90:353-402).
//
// Workgroup = W wavefronts, tile = 64 x (W*R) T-cells.  Wavefront w walks R consecutive
// T-rows upwards; a lane keeps one column.  For every row it updates the 12 stresses of its
// T-cell and forms the 8 `str` combinations in registers; the momentum equation of the U-row
// below needs str of (i,j), (i+1,j), (i,j+1), (i+1,j+1): the i+1 values come from the next
// lane by a wavefront shuffle, the j values are carried in registers from the previous row,
// so `str` never exists in memory.  Only the first row of each wavefront is handed to the
// wavefront below through LDS (4 doubles per lane, one barrier per workgroup).  The u, v
// stencil is served the same way: the row below is carried, the west neighbour is a shuffle
// (lane 0 reloads it).  Tiles overlap by one T-row / T-column; the overlap is recomputed and
// only the owner stores it (u, v, sigma are double-buffered).
// Consecutive blockIdx values are dealt round-robin to the 8 XCDs, so blockIdx is remapped
// to give every XCD one contiguous run of tiles: the re-read overlap rows then hit that
// XCD's own L2.  (Pure performance: any placement gives the same results.)
//
// DERIVE: dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym and tinyarea are exact functions of the two
// primary lengths HTN, HTE (ice_grid.F90:335-361, primary_grid_lengths_* :1139-1289); when the
// host has verified that bit for bit for this grid (Evp::init), the kernel recomputes them with
// the same expressions from HTN(i,j), HTN(i,j-1) (carried), HTE(i,j), HTE(i-1,j) (shuffle):
// 2 loads per T-cell instead of 9, results unchanged.
template <int W, int R, bool LAST, bool DAMP, bool DERIVE>
__global__ __launch_bounds__(64 * W, 4) void k_subcycle(const SubArgs a) {
  constexpr int TROWS = W * R;
  __shared__ double s_edge[W][4][TX];
  const int per_blk = a.tiles_x * a.tiles_y;
  const int nt = per_blk * a.nblocks;
  const int chunk = (nt + 7) >> 3;
  const int tile_lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile_lin >= nt) return;  // whole workgroup
  const int b = tile_lin / per_blk;
  const int rem = tile_lin - b * per_blk;
  const int tyi = rem / a.tiles_x, txi = rem - tyi * a.tiles_x;
  const int ilo = a.blk[6 * b + 0], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2],
            jhi = a.blk[6 * b + 3];
  const int i0 = ilo + txi * (TX - 1), j0 = jlo + tyi * (TROWS - 1);  // 1-based
  const int lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nx = a.nx;
  const size_t base = (size_t)b * nx * a.ny;
  const int i = i0 + lx;
  const bool in_i = i <= ihi + 1;
  // owner of T-cell (i,j) is the tile holding U-cell (min(i,ihi), min(j,jhi))
  const bool own_i = (min(i, ihi) - i0) < (TX - 1);
  const bool u_lane = lx < TX - 1 && i <= ihi;
  const int jfirst = j0 + w * R;

  // A tripole fold (applied by the halo update after this kernel) changes the owned cells of the top row whether they
  // carry ice or not, and u, v are double-buffered: what the fold of the previous subcycle left in a cell this kernel does
  // not write has to be carried over, or the next fold averages a stale value (the reference has one array).
  if (a.carry_top && w == 0 && u_lane && jhi >= j0 && jhi <= j0 + TROWS - 2) {
    const size_t qt = base + (size_t)(jhi - 1) * nx + (i - 1);
    if (!a.iceumask[qt]) {
      a.u_out[qt] = a.u_in[qt];
      a.v_out[qt] = a.v_in[qt];
    }
  }
  // carried row below the first T-row of this wavefront
  double us = c0, vs = c0, usw = c0, vsw = c0, hn_s = c0;
  {
    const int j = jfirst - 1;
    const bool ok = in_i && j <= jhi + 1;
    const size_t q = base + (size_t)(j - 1) * nx + (i - 1);
    if (ok) {
      us = a.u_in[q];
      vs = a.v_in[q];
    }
    usw = __shfl_up(us, 1);
    vsw = __shfl_up(vs, 1);
    if (lx == 0 && ok) {
      usw = a.u_in[q - 1];
      vsw = a.v_in[q - 1];
    }
    if (DERIVE && ok) hn_s = a.HTN[q];
  }
  double p0 = c0, pe1 = c0, p4 = c0, pe6 = c0;  // str of the row below: (i,j,1) (i+1,j,2) (i,j,5) (i+1,j,7)
  // The wavefront's last U-row is finished after the barrier; fetch its read-only inputs now so
  // that their latency overlaps the stress arithmetic.
  const int ju_last = jfirst + R - 1;
  const size_t qu_last = base + (size_t)(ju_last - 1) * nx + (i - 1);
  const bool do_last = (w < W - 1) && u_lane && ju_last <= jhi && a.iceumask[qu_last];
  // (only in the one-row-per-wavefront shapes used for small grids, which are latency-bound;
  // the multi-row shapes are bandwidth-bound and need the registers for occupancy)
  constexpr bool PREFETCH = (R == 1);
  UIn xl{};
  if (PREFETCH && do_last) load_uin<DERIVE>(a, qu_last, xl);

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int j = jfirst + r;
    const bool ok = in_i && j <= jhi + 1;
    const size_t q = base + (size_t)(j - 1) * nx + (i - 1);
    double un = c0, vn = c0;
    if (ok) {
      un = a.u_in[q];
      vn = a.v_in[q];
    }
    double uw = __shfl_up(un, 1), vw = __shfl_up(vn, 1);
    if (lx == 0 && ok) {
      uw = a.u_in[q - 1];
      vw = a.v_in[q - 1];
    }
    double hn = c0, he = c0, hew = c0;
    if (DERIVE) {
      if (ok) {
        hn = a.HTN[q];
        he = a.HTE[q];
      }
      hew = __shfl_up(he, 1);
      if (lx == 0 && ok) hew = a.HTE[q - 1];
    }
    StressOut o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o.str[c] = c0;
    if (ok && a.icetmask[q] == 1) {
      double s[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) s[c] = a.sig_in[(size_t)c * a.n + q];
      double Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp, Cxm, Cym, Tiny;
      if (DERIVE) {
        Dxt = p5 * (hn + hn_s);                 // ice_grid.F90:1184 (dxt)
        Dyt = p5 * (he + hew);                  // :1271 (dyt)
        Dxhy = p5 * (he - hew);                 // :347
        Dyhx = p5 * (hn - hn_s);                // :348
        Cyp = 1.5 * he - p5 * hew;              // :354
        Cxp = 1.5 * hn - p5 * hn_s;             // :355
        Cym = -(1.5 * hew - p5 * he);           // :357
        Cxm = -(1.5 * hn_s - p5 * hn);          // :358
        Tiny = puny * (Dxt * Dyt);              // :334, :346
      } else {
        Dxt = a.dxt[q]; Dyt = a.dyt[q]; Dxhy = a.dxhy[q]; Dyhx = a.dyhx[q]; Cxp = a.cxp[q];
        Cyp = a.cyp[q]; Cxm = a.cxm[q]; Cym = a.cym[q]; Tiny = a.tinyarea[q];
      }
      stress_cell<LAST, DAMP>(a.sc, un, uw, usw, us, vn, vw, vsw, vs, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp,
                              Cxm, Cym, LAST ? a.tarear[q] : 0.0, Tiny, a.strength[q], s, o);
      if (own_i && (min(j, jhi) - j0) < (TROWS - 1)) {
#pragma unroll
        for (int c = 0; c < 12; ++c) a.sig_out[(size_t)c * a.n + q] = s[c];
        if (LAST && j >= a.diag_jmin) {
          a.divu[q] = o.divu;
          a.rdg_conv[q] = o.rdg_conv;
          a.rdg_shear[q] = o.rdg_shear;
          a.shear[q] = o.shear;
          a.prs_sig[q] = o.prs_sig;
        }
      }
    }
    const double e1 = __shfl_down(o.str[1], 1), e3 = __shfl_down(o.str[3], 1),
                 e6 = __shfl_down(o.str[6], 1), e7 = __shfl_down(o.str[7], 1);
    if (r == 0) {
      if (w > 0) {  // complete the last U-row of the wavefront below
        s_edge[w][0][lx] = o.str[2];
        s_edge[w][1][lx] = e3;
        s_edge[w][2][lx] = o.str[5];
        s_edge[w][3][lx] = e7;
      }
    } else {
      const int ju = j - 1;  // U-row between the carried row and this one
      const size_t qu = q - nx;
      if (u_lane && ju <= jhi && a.iceumask[qu]) {
        UIn x;
        load_uin<DERIVE>(a, qu, x);
        const double sx = p0 + pe1 + o.str[2] + e3;      // :1415-1416 order
        const double sy = p4 + o.str[5] + pe6 + e7;      // :1417-1418 order
        stepu_store<LAST>(a, x, qu, i, ju, ilo, ihi, jlo, jhi, us, vs, sx, sy);
      }
    }
    p0 = o.str[0]; pe1 = e1; p4 = o.str[4]; pe6 = e6;
    us = un; vs = vn; usw = uw; vsw = vw;
    hn_s = hn;
  }
  __syncthreads();
  if (do_last) {
    if (!PREFETCH) load_uin<DERIVE>(a, qu_last, xl);
    const double sx = p0 + pe1 + s_edge[w + 1][0][lx] + s_edge[w + 1][1][lx];
    const double sy = p4 + s_edge[w + 1][2][lx] + pe6 + s_edge[w + 1][3][lx];
    stepu_store<LAST>(a, xl, qu_last, i, ju_last, ilo, ihi, jlo, jhi, us, vs, sx, sy);
  }
}

// TWO EVP subcycles in one launch (ice_dyn_evp.F90:353-402 twice).
//
// sigma is 24 of the ~41 words a cell moves per subcycle and every other input is read-only, so
// running subcycles k and k+1 back to back on a tile halves the traffic per subcycle (sigma, u, v
// and the inputs cross HBM once per PAIR) and halves the launches.  The price is a redundant rim:
// a workgroup of W wavefronts x 64 lanes owns (W-3) rows x 59 columns.
//
// Wavefront w keeps T-row jt+w, a lane keeps one column; everything a cell needs for the second
// subcycle (its 12 stresses, metrics, strength, the U-cell inputs) is already in its lane's
// registers.  Only the intermediate velocity has to travel: west neighbour by shuffle, row below
// through LDS (one row of u, v per wavefront), exactly like `str`.
//   stage 1: stress on every T-cell of the tile -> str -> u' on rows/lanes that have their N/E
//            neighbours in the tile (w <= W-2, lane <= 62); nothing is stored;
//   stage 2: stress with u' -> sigma'' (valid for 1 <= w <= W-2) -> u'' (valid for 1 <= w <= W-3);
//            owners store sigma'', u'', v'' (+ the LAST diagnostics) and forward u'', v'' to
//            the ghost cells that mirror them.
// Ghost cells: rows beyond the block (j = jlo-1, jhi+1) hold values that do not change during the
// subcycling (open/closed domain edge, or rows beyond the overlap of a wide-halo slab), so their u'
// is their u.  Columns: the east ghost T-cell G = (ihi+1, j) has its own sigma (the reference
// computes it separately, :850-859) but its velocity mirrors column ilo, and the west ghost
// velocity mirrors column ihi.  Lanes therefore walk the block's columns as a ring
// ..., ihi-1, ihi, G, ilo, ilo+1, ...: the first tile starts two positions before ilo (at ihi, G)
// and the last one runs past G into ilo, ilo+1, so every tile computes the mirrored u' it needs
// itself.  Two exceptions to "west neighbour = lane-1" follow from the ring: the lane at ilo takes
// u' of ihi from lane-2 (G sits in between), and G takes its own-column u' from lane+1 (= ilo).
// With an open/closed E-W edge there is no ring: the ghost columns keep their values.
//
// Used when no ghost ROW of a local block changes between two subcycles (one block over the full
// width per rank, or wide-halo slabs with an even overlap); otherwise k_subcycle runs.
constexpr int OWN_LANE0 = 2, OWN_LANES = 59;  // lanes 2..60 own their column

// Addressing inside k_subcycle2: every array is indexed as (uniform pointer to the block's plane)
// + (one 32-bit byte offset per lane), which the compiler turns into SGPR-pair + VGPR-offset
// addressing: no 64-bit vector arithmetic per access.  (A block plane is < 4 GB: Evp::init checks.)
__device__ __forceinline__ double ld8(const double* p, unsigned off) {
  return *(const double*)((const char*)p + off);
}
__device__ __forceinline__ void st8(double* p, unsigned off, double v) { *(double*)((char*)p + off) = v; }
__device__ __forceinline__ int ld4(const int32_t* p, unsigned off) {
  return *(const int32_t*)((const char*)p + off);
}
typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dbl2 ld16(const double* p, unsigned off) {   // p + off is 16-byte aligned
  return *(const dbl2*)((const char*)p + off);
}
__device__ __forceinline__ void st16(double* p, unsigned off, double x, double y) {
  dbl2 v;
  v.x = x;
  v.y = y;
  *(dbl2*)((char*)p + off) = v;
}

template <bool DERIVE>
__device__ __forceinline__ void load_uin_o(const SubArgs& a, size_t base, unsigned qo, UIn& x) {
  x.aiu = ld8(a.aiu + base, qo); x.uocn = ld8(a.uocn + base, qo); x.vocn = ld8(a.vocn + base, qo);
  if (!DERIVE) {
    x.waterx = ld8(a.waterx + base, qo);
    x.watery = ld8(a.watery + base, qo);
  }
  x.forcex = ld8(a.forcex + base, qo); x.forcey = ld8(a.forcey + base, qo);
  x.umassdtei = ld8(a.umassdtei + base, qo); x.fm = ld8(a.fm + base, qo); x.uarear = ld8(a.uarear + base, qo);
  if (DERIVE) water_of(x.uocn, x.vocn, x.fm, x.waterx, x.watery);   // instead of two loads
}

// stepu_store with the same addressing
template <bool LAST>
__device__ __forceinline__ void stepu_store_o(const SubArgs& a, const UIn& x, size_t base, unsigned qo, int i,
                                              int j, int ilo, int ihi, int jlo, int jhi, double uold,
                                              double vold, double sx, double sy) {
  StepuOut r;
  stepu_cell(uold, vold, x.aiu, x.uocn, x.vocn, x.waterx, x.watery, x.forcex, x.forcey, x.umassdtei,
             x.fm, x.uarear, sx, sy, r);
  st8(a.u_out + base, qo, r.u);
  st8(a.v_out + base, qo, r.v);
  if (LAST) {
    st8(a.strintx + base, qo, r.strintx);
    st8(a.strinty + base, qo, r.strinty);
    st8(a.strocnx + base, qo, r.taux);
    st8(a.strocny + base, qo, r.tauy);
  }
  if (a.ring_slot && (i == ilo || i == ihi || j == jlo || j == jhi)) {
    const int slot = a.ring_slot[base + (qo >> 3)];
    if (slot >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int d = a.fwd[3 * slot + k];
        if (d >= 0) {
          a.u_out[d] = r.u;
          a.v_out[d] = r.v;
        }
      }
    }
  }
}

template <int W, bool LAST, bool DAMP, bool DERIVE>
__global__ __launch_bounds__(64 * W, (W == 12 ? 3 : 4)) void k_subcycle2(const SubArgs a) {
  __shared__ double s_str[W][4][TX];
  __shared__ double s_uv[W][2][TX];
  // W = 16 (the large grids, where a CU works through many workgroups one at a time): the eight U-cell
  // inputs are parked in LDS between the two momentum updates instead of being fetched again -- 0.1 degree
  // +6 %; at gx1 (W = 13) the extra LDS traffic costs 2 %, so the smaller shapes fetch twice (L2 hits)
  constexpr bool PARK = DERIVE && W >= 16;
  __shared__ double s_x[PARK ? W : 1][8][TX];
  const int per_blk = a.tiles_x * a.tiles_y;
  const int nt = per_blk * a.nblocks;
  const int chunk = (nt + 7) >> 3;
  const int tile_lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile_lin >= nt) return;  // whole workgroup
  const int b = tile_lin / per_blk;
  const int rem = tile_lin - b * per_blk;
  const int tyi = rem / a.tiles_x, txi = rem - tyi * a.tiles_x;
  const int ilo = a.blk[6 * b + 0], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2],
            jhi = a.blk[6 * b + 3];
  const int lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nx = a.nx;
  const size_t base = (size_t)b * nx * a.ny;
  const bool cyc = a.ew_cyclic != 0;

  // column: ring position k (0 = ilo, ncol = G) -> memory column
  const int ncol = ihi - ilo + 1;
  const int kraw = txi * OWN_LANES + lx - OWN_LANE0;
  int col = -1;
  if (cyc) {
    int k = kraw % (ncol + 1);
    if (k < 0) k += ncol + 1;
    col = ilo + k;  // k == ncol -> ihi+1 = G
  } else if (kraw >= -1 && kraw <= ncol) {
    col = ilo + kraw;  // -1 -> the west ghost column
  }
  const bool col_ok = col >= 1;
  const bool isG = col == ihi + 1;
  const bool at_ilo = col == ilo;
  const bool own_col = lx >= OWN_LANE0 && lx < OWN_LANE0 + OWN_LANES && kraw >= 0 && kraw <= ncol;
  // row
  const int j = jlo - 1 + tyi * (W - 3) + w;
  const bool row_ok = j <= jhi + 1;                    // j >= jlo-1 = 1 always
  const bool own_row = w >= 1 && w <= W - 3 && row_ok;
  const bool tcell = col_ok && row_ok && j >= jlo && col >= ilo;       // (col <= ihi+1, j <= jhi+1 hold)
  const bool ucell = col_ok && j >= jlo && j <= jhi && col >= ilo && col <= ihi;
  const unsigned qo = ((unsigned)(j - 1) * (unsigned)nx + (unsigned)(col_ok ? col - 1 : 0)) * 8u;   // byte offset in the block plane
  const unsigned nx8 = (unsigned)nx * 8u;
  const bool ld = col_ok && row_ok;
  const bool reload_w = lx == 0 || at_ilo;             // west neighbour from memory, not from lane-1

  // ---- loads (once for both subcycles) ----
  double un = c0, vn = c0, us = c0, vs = c0;
  if (ld) {
    un = ld8(a.u_in + base, qo);
    vn = ld8(a.v_in + base, qo);
    if (j >= 2) {
      us = ld8(a.u_in + base, qo - nx8);
      vs = ld8(a.v_in + base, qo - nx8);
    }
  }
  double uw = up1(un), vw = up1(vn), usw = up1(us), vsw = up1(vs);
  if (reload_w && ld && col >= 2) {
    uw = ld8(a.u_in + base, qo - 8u);
    vw = ld8(a.v_in + base, qo - 8u);
    if (j >= 2) {
      usw = ld8(a.u_in + base, qo - nx8 - 8u);
      vsw = ld8(a.v_in + base, qo - nx8 - 8u);
    }
  }
  const bool tact = tcell && ld4(a.icetmask + base, qo >> 1) == 1;
  const bool uact = ucell && ld4(a.iceumask + base, qo >> 1) != 0;
  double s[12];
  // T-cell metrics: DERIVE keeps the four primary lengths and forms the nine metrics in each stage
  // (registers); otherwise the nine loaded values are kept
  double hn = c0, he = c0, hn_s = c0, hew = c0;
  double Dxt = c0, Dyt = c0, Dxhy = c0, Dyhx = c0, Cxp = c0, Cyp = c0, Cxm = c0, Cym = c0, Tiny = c0,
         St = c0, Tarear = c0;
  {
    if (DERIVE) {
      if (ld) {
        hn = ld8(a.HTN + base, qo);
        he = ld8(a.HTE + base, qo);
        if (j >= 2) hn_s = ld8(a.HTN + base, qo - nx8);
      }
      hew = up1(he);
      if (reload_w && ld && col >= 2) hew = ld8(a.HTE + base, qo - 8u);
    }
    if (tact) {
#pragma unroll
      for (int c = 0; c < 12; ++c) s[c] = ld8(a.sig_in_p[c] + base, qo);
      if (!DERIVE) {
        Dxt = ld8(a.dxt + base, qo); Dyt = ld8(a.dyt + base, qo); Dxhy = ld8(a.dxhy + base, qo);
        Dyhx = ld8(a.dyhx + base, qo); Cxp = ld8(a.cxp + base, qo); Cyp = ld8(a.cyp + base, qo);
        Cxm = ld8(a.cxm + base, qo); Cym = ld8(a.cym + base, qo); Tiny = ld8(a.tinyarea + base, qo);
      }
      St = ld8(a.strength + base, qo);
      if (LAST) Tarear = ld8(a.tarear + base, qo);
    } else {
#pragma unroll
      for (int c = 0; c < 12; ++c) s[c] = c0;
    }
  }
  auto metrics = [&]() {
    if (DERIVE) {
      Dxt = p5 * (hn + hn_s);                 // ice_grid.F90:1184 (dxt)
      Dyt = p5 * (he + hew);                  // :1271 (dyt)
      Dxhy = p5 * (he - hew);                 // :347
      Dyhx = p5 * (hn - hn_s);                // :348
      Cyp = 1.5 * he - p5 * hew;              // :354
      Cxp = 1.5 * hn - p5 * hn_s;             // :355
      Cym = -(1.5 * hew - p5 * he);           // :357
      Cxm = -(1.5 * hn_s - p5 * hn);          // :358
      Tiny = puny * (Dxt * Dyt);              // :334, :346
    }
  };

  // ---- subcycle k: stress, str, u' ----
  StressOut o;
#pragma unroll
  for (int c = 0; c < 8; ++c) o.str[c] = c0;
  if (tact) {
    metrics();
    stress_cell<false, DAMP>(a.sc, un, uw, usw, us, vn, vw, vsw, vs, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp, Cxm,
                             Cym, 0.0, Tiny, St, s, o);
  }
  double e1 = down1(o.str[1]), e3 = down1(o.str[3]), e6 = down1(o.str[6]),
         e7 = down1(o.str[7]);
  s_str[w][0][lx] = o.str[2];
  s_str[w][1][lx] = e3;
  s_str[w][2][lx] = o.str[5];
  s_str[w][3][lx] = e7;
  __syncthreads();
  double u1 = un, v1 = vn;   // velocity of this lane's own column after subcycle k
  if (uact && lx < TX - 1 && w < W - 1) {
    const double sx = o.str[0] + e1 + s_str[w + 1][0][lx] + s_str[w + 1][1][lx];   // :1415-1416 order
    const double sy = o.str[4] + s_str[w + 1][2][lx] + e6 + s_str[w + 1][3][lx];   // :1417-1418 order
    UIn x;   // fetched for each of the two momentum updates (L2 hits the second time): they would
             // otherwise occupy 16-20 registers across both stress evaluations
    load_uin_o<DERIVE>(a, base, qo, x);
    if (PARK) {
      s_x[w][0][lx] = x.aiu; s_x[w][1][lx] = x.uocn; s_x[w][2][lx] = x.vocn; s_x[w][3][lx] = x.forcex;
      s_x[w][4][lx] = x.forcey; s_x[w][5][lx] = x.umassdtei; s_x[w][6][lx] = x.fm; s_x[w][7][lx] = x.uarear;
    }
    StepuOut r;
    stepu_cell(un, vn, x.aiu, x.uocn, x.vocn, x.waterx, x.watery, x.forcex, x.forcey, x.umassdtei, x.fm,
               x.uarear, sx, sy, r);
    u1 = r.u;
    v1 = r.v;
  }
  // The ring applies to rows whose velocity is updated (jlo..jhi): there the ghost columns mirror
  // the opposite edge after every subcycle (:397-402).  On the ghost rows nothing ever changes,
  // and their corner ghosts keep whatever they hold.
  const bool mirror_n = cyc && j >= jlo && j <= jhi, mirror_s = cyc && j - 1 >= jlo && j - 1 <= jhi;
  const double gu = down1(u1), gv = down1(v1);   // G mirrors column ilo = next lane
  const double un1 = (isG && mirror_n) ? gu : u1, vn1 = (isG && mirror_n) ? gv : v1;
  s_uv[w][0][lx] = un1;
  s_uv[w][1][lx] = vn1;
  __syncthreads();

  // ---- subcycle k+1 ----
  double us1 = us, vs1 = vs;   // wavefront 0 owns nothing: any value will do there
  if (w > 0) {
    us1 = s_uv[w - 1][0][lx];
    vs1 = s_uv[w - 1][1][lx];
  }
  double uw1 = up1(un1), vw1 = up1(vn1), usw1 = up1(us1), vsw1 = up1(vs1);
  {
    // at ilo the west neighbour is column ihi, two lanes to the left (G sits in between), or,
    // where nothing mirrors, the ghost column's own unchanged value
    const double uw2 = up1(up1(un1)), vw2 = up1(up1(vn1)), usw2 = up1(up1(us1)),
                 vsw2 = up1(up1(vs1));
    if (at_ilo) {
      uw1 = uw2; vw1 = vw2; usw1 = usw2; vsw1 = vsw2;
      if (!mirror_n && ld) {   // unchanged ghost value: read it again rather than keep it in registers
        uw1 = ld8(a.u_in + base, qo - 8u);
        vw1 = ld8(a.v_in + base, qo - 8u);
      }
      if (!mirror_s && ld && j >= 2) {
        usw1 = ld8(a.u_in + base, qo - nx8 - 8u);
        vsw1 = ld8(a.v_in + base, qo - nx8 - 8u);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) o.str[c] = c0;
  // the second stress evaluation of the first and the last wavefront feeds nothing: sigma'' is owned by
  // wavefronts 1..W-3 and u'' of wavefront w needs str of wavefronts w and w+1 <= W-2
  if (tact && w >= 1 && w <= W - 2) {
    metrics();
    stress_cell<LAST, DAMP>(a.sc, un1, uw1, usw1, us1, vn1, vw1, vsw1, vs1, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp,
                            Cxm, Cym, Tarear, Tiny, St, s, o);
    if (own_col && own_row) {
#pragma unroll
      for (int c = 0; c < 12; ++c) st8(a.sig_out_p[c] + base, qo, s[c]);
      if (LAST) {
        st8(a.divu + base, qo, o.divu);
        st8(a.rdg_conv + base, qo, o.rdg_conv);
        st8(a.rdg_shear + base, qo, o.rdg_shear);
        st8(a.shear + base, qo, o.shear);
        st8(a.prs_sig + base, qo, o.prs_sig);
      }
    }
  }
  e1 = down1(o.str[1]); e3 = down1(o.str[3]);
  e6 = down1(o.str[6]); e7 = down1(o.str[7]);
  s_str[w][0][lx] = o.str[2];   // stage-1 values were consumed before the previous barrier
  s_str[w][1][lx] = e3;
  s_str[w][2][lx] = o.str[5];
  s_str[w][3][lx] = e7;
  __syncthreads();
  if (uact && own_col && own_row) {
    const double sx = o.str[0] + e1 + s_str[w + 1][0][lx] + s_str[w + 1][1][lx];
    const double sy = o.str[4] + s_str[w + 1][2][lx] + e6 + s_str[w + 1][3][lx];
    UIn x;
    if (PARK) {   // this lane wrote them itself before the first momentum update
      x.aiu = s_x[w][0][lx]; x.uocn = s_x[w][1][lx]; x.vocn = s_x[w][2][lx]; x.forcex = s_x[w][3][lx];
      x.forcey = s_x[w][4][lx]; x.umassdtei = s_x[w][5][lx]; x.fm = s_x[w][6][lx]; x.uarear = s_x[w][7][lx];
      water_of(x.uocn, x.vocn, x.fm, x.waterx, x.watery);
    } else {
      load_uin_o<DERIVE>(a, base, qo, x);
    }
    stepu_store_o<LAST>(a, x, base, qo, col, j, ilo, ihi, jlo, jhi, u1, v1, sx, sy);
  }
}

// ---- K EVP subcycles in one sweep: time-skewed streaming (grids that do not fit the chip) ------------------------
//
// k_subcycle2 sends sigma, u, v and the inputs across HBM once per PAIR of subcycles and pays a redundant rim of
// 1.34 x for it ((16 x 64) / (13 x 59)); at 0.1 degree its memory phase and its arithmetic add up.
// Here a workgroup is a PIPELINE OF K TIME LEVELS, one wavefront per level, that sweeps a strip of 64 columns upward
// row by row, once: wavefront k turns state k into state k+1.  It marches like k_subcycle -- per step the stresses of
// one T-row, then the momentum equation of the U-row below it, `str` of the previous row carried in registers, the
// eastern neighbour by wave shift -- but only level 0 reads sigma, u, v from memory and only level K-1 writes them;
// in between a row travels from level k to level k+1 through LDS (12 + 2 doubles per lane, two slots each).
// Level k+1 needs u_k of rows r and r-1 for T-row r, and u_k(r) exists once level k has finished T-row r+1: the
// levels run TWO ROWS apart, one workgroup barrier per step.  HBM sees sigma, u, v once per K subcycles; there is no
// redundant row (only 2(K-1) steps to fill the pipeline and K rim rows at the ends of a row segment) and K+1 / K
// redundant columns at the west / east end of the 64 lanes (each level loses one lane per side; the ring's (G, ilo)
// pair costs one more, as in k_subcycle2): a strip owns 63 - 2K columns.  Loads run one step ahead of their use
// (level 0: from memory; the others: the stresses from LDS), so a wavefront's memory latency overlaps its own
// arithmetic, and the three workgroups of a CU drift apart by themselves.  The read-only inputs (HTN, HTE, strength,
// the eight momentum inputs: 88 B per cell) are fetched by every level, two steps after the level before it: L2 /
// Infinity-Cache hits.
// Same arithmetic, same order, same bits as k_subcycle; columns walk the block as a ring exactly as in k_subcycle2.
// Used where k_subcycle2 can be used (can_fuse) and the metrics derive from HTN / HTE.
// In-kernel clock (MI355X_MICROARCH.md, DVFS item 6): a DIAGNOSTIC build (-DCICE4_AMD_STAMPS, scripts/build_ab.sh) stamps
// the shader-cycle counter (s_memtime) and the 100 MHz wall clock (s_memrealtime) once before and once after the loop
// of k_evp_resident / k_subcycle_skew; clock = d(cycles) / d(ticks) x 100 MHz, median over the workgroups
// (scripts/inkernel_clock.py).  The product build has no stamp: both helpers are empty there.  The stamps go to a
// buffer of their own that nothing else reads.
__device__ __forceinline__ void stamp_at(long long* stamps, int slot) {
#ifdef CICE4_AMD_STAMPS
  if (stamps && threadIdx.x == 0) {
    stamps[4 * (size_t)blockIdx.x + slot] = (long long)__builtin_amdgcn_s_memtime();
    stamps[4 * (size_t)blockIdx.x + slot + 2] = (long long)__builtin_amdgcn_s_memrealtime();
  }
#else
  (void)stamps; (void)slot;
#endif
}

}  // namespace

// Phase clock of the one-launch loop (diagnostic build only): cycles a workgroup's wavefront 0 spends between marked points
// of a subcycle, summed over the subcycles of a launch (scripts/resident_phases.py).  Nothing in the product build.
#ifdef CICE4_AMD_STAMPS
#define PHASE_DECL long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ph_t_ = (long long)__builtin_amdgcn_s_memtime();
#define PHASE(i) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); ph_[i] += tn_ - ph_t_; ph_t_ = tn_; }
#define PHASE_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define PHASE_STORE(buf) if ((buf) && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) (buf)[8 * (size_t)blockIdx.x + i_] = ph_[i_]; }
#define PHASE_COUNT(i, n) { ph_[i] += (n); }
// ... and a trace of a few tiles (GRAN loop): wall-clock ticks of every wavefront at six points of subcycles 60 .. 63
#define TRACE_DECL const int tr_slot_ = (r.stamps && tile % 29 == 7 && tile / 29 < 8) ? tile / 29 : -1; \
                   long long* const tr_ = r.stamps + 12 * (size_t)gridDim.x;
#define TRACE(i) if (tr_slot_ >= 0 && lx == 0 && k >= 60 && k < 64) tr_[((tr_slot_ * 12 + w) * 4 + (k - 60)) * 6 + (i)] = wall_clock64();
#else
#define PHASE_COUNT(i, n)
#define TRACE_DECL
#define TRACE(i)
#define PHASE_DECL
#define PHASE(i)
#define PHASE_DRAIN
#define PHASE_STORE(buf)
#endif

struct SkewArgs {
  SubArgs a;               // only what the kernel names is fetched from the argument block
  const double* st_in;     // u, v, 12 stresses of the current state: 14 planes of a.n doubles
  double* st_out;          // the other copy
  const double* uar;       // aiu, uocn, vocn, forcex, forcey, umassdtei, fm, uarear: 8 planes of a.n doubles
  int seg_rows;            // U-rows a workgroup owns (when there is no table)
  const int32_t* rows;     // [tiles][2]: first and last U-row (relative to jlo) a workgroup owns, or NULL
  int stagger_ticks;       // start delay per workgroup "generation", in 10 ns ticks (see the kernel)
  int stagger_mod;         // generations
  int prio_rotate;         // rotate the issue priority among the workgroups sharing a CU (see the kernel)
  int level_deal;          // how the time levels are dealt to the wavefronts of a workgroup (see the kernel)
  int fwd_rule;            // the on-rank ghost copies are exactly the east-west wrap of full-width blocks (Evp::init checked)
  long long* dbg;          // test aid: [2 * workgroups] start / end wall-clock ticks (10 ns), or NULL
  long long* stamps;       // -DCICE4_AMD_STAMPS (diagnostic build only): [4 * workgroups], see stamp_at
  // SKEW_WIDE: the read-only inputs interleaved cell by cell (Evp::prepare / init build them): uar4 = 4 planes of pairs
  // {aiu, uocn} {vocn, forcex} {forcey, umassdtei} {fm, uarear}; hnhe = one plane of pairs {HTN, HTE}; msk = one int per cell:
  // bit 0 icetmask == 1, bit 1 iceumask != 0.  A level's 14 loads per step become 4 + 1 of 16 bytes, strength, and one int.
  const double* uar4;
  const double* hnhe;
  const int32_t* msk;
  long long* phases;       // diagnostic build: [8 * K * workgroups] cycles per phase of a step, per level, summed over the sweep
  int own_shift;           // strip 0 owns this many lanes less (see the kernel's column geometry)
  // a launch over an explicit LIST of tiles (Evp::build_split: the sweep in front of a wide-halo refresh runs as two
  // launches, the segments the neighbours wait for first): [4 per tile] block, strip, first and last owned row relative
  // to jlo; this launch covers tile_count entries from tile_first on.  NULL: every tile of the (strip x segment) grid.
  const int32_t* tiles;
  int tile_first, tile_count;
  // [block][strip][row - jlo] != 0: the strip has something to compute in that row (k_skew_rowact), or NULL: a workgroup
  // shrinks its segment to its first .. last such row and leaves at once if there is none
  const unsigned char* rowact;
  const int32_t* run_next;   // [block][strip][row - jlo]: k_skew_runs
  const int32_t* run_end;
};

namespace {

// WS: wavefronts per SIMD the kernel is built for (bounds the registers)
//
// Addressing: every array of the launch is (one uniform base) + (one 32-bit byte offset per lane); the 14 planes of
// the state and the 8 planes of the momentum inputs are reached by adding plane * n * 8 to the lane offset
// (Evp::can_skew checks that 14 planes stay below 4 GB), so the kernel holds a handful of base pointers, not forty.
// PAIRS: u | v and the stresses two by two live interleaved cell by cell in the sweep's own copies of the state
// (Evp::st2: 7 planes of pairs {u, v} {s1, s2} ... {s11, s12}): level 0 fetches a row with 7 loads of 16 bytes instead of
// 14 of 8, the last level stores it with 7.  The levels at the two ends of the pipeline are the ones every step waits
// for (profiles/r04_sweep_phases*.txt), and what makes them slow is the NUMBER of vector-memory instructions they issue.
// The sweep that ends evp(dt) (LAST) stores in the ordinary plane layout, so that nothing has to be converted back.
// S: wavefronts per LEVEL.  S = 3: twelve wavefronts (K = 4), ONE workgroup per CU instead of three -- the same three
// wavefronts per SIMD and the same LDS (3 x 42 KB), but the three strips of 64 lanes lie side by side and overlap by two
// columns instead of each losing 2K: a workgroup spans TXW = 62 S + 2 columns and owns TXW - 2K of them (180 of 188
// against 3 x 56 of 192).  Rows travel from level to level through LDS indexed by the workgroup's COLUMN, so a wavefront
// finds the velocity of its first lane's western neighbour -- produced by the wavefront beside it -- where it finds its
// own.  Of the two shared columns the western wavefront owns the first (its lane 62: stress and momentum right), the
// eastern one the second (its lane 1); lane 63 of the one and lane 0 of the other compute nothing that is kept.
template <int K, bool LAST, bool DAMP, int WS, bool PAIRS = false, int S = 1>
__global__ __launch_bounds__(64 * K * S, (S == 1 ? WS : 1)) void k_subcycle_skew(const SkewArgs sa) {
  const SubArgs& a = sa.a;
  constexpr bool PIN = PAIRS, POUT = PAIRS && !LAST;
  constexpr int TXW = 62 * S + 2;       // columns a workgroup spans (S = 1: the 64 lanes of its one wavefront per level)
  // Columns: a level loses one lane per side (the stress needs the western neighbour's velocity, the momentum equation
  // the eastern neighbour's stress), so after K levels lanes K .. 63-K are right: a strip owns OWNW = 64 - 2K columns.
  // Strip 0 starts at the ring's seam: its first owned lane is ilo, whose western neighbour ihi sits TWO lanes away (G in
  // between) -- one lane more of rim, it owns lanes K+1 .. 63-K.  If that layout puts G into the east rim of a strip within
  // K-1 lanes of its last owned lane (the dependency path crosses G without gaining a level and would need a lane more),
  // strip 0 gives up sa.own_shift lanes and everything moves west until G sits on that strip's lane 63 (Evp::skew_strips).
  constexpr int OWNW = TXW - 2 * K;
  // EARLY: the hand-off of level 0 has THREE slots (row mod 3), s_sig0; the others two, s_sig[k - 1] for level k >= 1.
  // Level 0 can then put the stresses it has just formed into LDS at the END of its step (the slot was read two steps
  // ago) and fetch those of the next row into the same registers BEFORE the barrier: they are in flight during the
  // barrier and the first third of the next step instead of being waited for in the middle of it.
  constexpr bool EARLY = SKEW_EARLY && !(SKEW_TPASS && K <= 4) && S == 1;
  __shared__ double s_sig[EARLY ? (K > 2 ? K - 2 : 1) : K - 1][2][12][TXW];
  __shared__ double s_sig0[EARLY ? 3 : 1][EARLY ? 12 : 1][TX];
  __shared__ double s_uv[K - 1][2][2][TXW];
  // TP: the T-cell inputs of a row (HTN, HTE, strength, icetmask, iceumask) ride along with it: only level 0 fetches them
  // from memory, the others find them in LDS a step after the level before them held them (10.5 KB more per workgroup:
  // K = 4 stays at three workgroups per CU, 3 x 52.5 KB of 160)
  constexpr bool TP = SKEW_TPASS && K <= 4 && S == 1;
  __shared__ double s_tin[TP ? K - 1 : 1][2][3][TX];
  __shared__ int s_msk[TP ? K - 1 : 1][2][TX];
  const int per_blk = a.tiles_x * a.tiles_y;
  const int nt = sa.tiles ? sa.tile_count : per_blk * a.nblocks;
  const int chunk = (nt + 7) >> 3;
  int tile_lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile_lin >= nt) return;  // whole workgroup
  int b, rem, tyi, txi, ja_rel = 0, jb_rel = -1;
  if (sa.tiles) {
    tile_lin += sa.tile_first;
    b = sa.tiles[4 * tile_lin];
    txi = sa.tiles[4 * tile_lin + 1];
    ja_rel = sa.tiles[4 * tile_lin + 2];
    jb_rel = sa.tiles[4 * tile_lin + 3];
    rem = 0;
    tyi = 0;
  } else {
    b = tile_lin / per_blk;
    rem = tile_lin - b * per_blk;
    tyi = rem / a.tiles_x;
    txi = rem - tyi * a.tiles_x;
  }
  const int ilo = a.blk[6 * b + 0], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  const int lx = threadIdx.x & 63;
  // time level of this wavefront (uniform); dealt differently from workgroup to workgroup, so that the wavefronts
  // a SIMD holds are at different levels (level 0 waits for memory, level K-1 stores).  S > 1: wavefront = level x S +
  // sub-strip, so the S wavefronts of a level land on different SIMDs too
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int sub = S == 1 ? 0 : wv % S;
  // generation of this workgroup: workgroups are dispatched in blockIdx order, one per CU first
  const int gen = (int)(blockIdx.x / (gridDim.x / (unsigned)sa.stagger_mod + 1u));
  const int k = ((S == 1 ? wv : wv / S) + tile_lin + sa.level_deal * gen) % K;
  const int cw = sub * 62 + lx;          // this lane's column of the workgroup
  const int nx = a.nx;
  const unsigned n8 = (unsigned)(a.n * 8);                  // bytes between two planes
  const size_t base = (size_t)b * nx * a.ny;
  const double* const u_in = sa.st_in + (PIN ? 2 : 1) * base;    // block planes: u, v at +n, stress c at +(2+c) n; PAIRS: see above
  double* const s_out = sa.st_out + (POUT ? 2 : 1) * base;
  const double* const uar = sa.uar + base;
  const double* const htn = a.HTN + base;
  const double* const hte = a.HTE + base;
  const double* const stren = a.strength + base;
  const int32_t* const tmk = a.icetmask + base;
  const int32_t* const umk = a.iceumask + base;
  constexpr bool WIDE = SKEW_WIDE != 0;
  const double* const uar4 = sa.uar4 + 2 * base;           // pairs: 16 bytes per cell and plane
  const double* const hnhe = sa.hnhe + 2 * base;
  const int32_t* const msk = sa.msk + base;
  const unsigned n16 = (unsigned)(a.n * 16);               // bytes between two planes of pairs
  // u, v of the cell at (8-byte layout) byte offset q8 of the state a sweep reads / writes
  auto ld_uv = [&](unsigned q8, double& u, double& v) {
    if (PIN) {
      const dbl2 t = ld16(u_in, 2u * q8);
      u = t.x;
      v = t.y;
    } else {
      u = ld8(u_in, q8);
      v = ld8(u_in, q8 + n8);
    }
  };
  auto st_uv = [&](unsigned q8, double u, double v) {
    if (POUT) {
      st16(s_out, 2u * q8, u, v);
    } else {
      st8(s_out, q8, u);
      st8(s_out, q8 + n8, v);
    }
  };
  const bool cyc = a.ew_cyclic != 0;
  // column: ring position (0 = ilo, ncol = G) -> memory column, as in k_subcycle2
  const int ncol = ihi - ilo + 1;
  const int own0 = txi == 0 ? K + 1 + sa.own_shift : K;
  const int kraw = (txi == 0 ? 0 : (OWNW - 1 - sa.own_shift) + (txi - 1) * OWNW) + cw - own0;
  int col = -1;
  if (cyc) {
    int kk = kraw % (ncol + 1);
    if (kk < 0) kk += ncol + 1;
    col = ilo + kk;
  } else if (kraw >= -1 && kraw <= ncol) {
    col = ilo + kraw;
  }
  const bool col_ok = col >= 1;
  const bool isG = cyc && col == ihi + 1;
  const bool at_ilo = cyc && col == ilo;
  // (S > 1: of the two columns neighbouring wavefronts share, lane 62 of the western one owns the first, lane 1 of the
  //  eastern one the second)
  const bool own_lane = S == 1 || (lx >= (sub == 0 ? 0 : 1) && lx <= (sub == S - 1 ? 63 : 62));
  const bool own_col = own_lane && cw >= own0 && cw <= TXW - 1 - K && kraw >= 0 && kraw <= ncol;
  const bool tcol = col_ok && col >= ilo;                 // (col <= ihi + 1 holds)
  const bool ucol = col_ok && col >= ilo && col <= ihi;
  const unsigned co = (unsigned)(col_ok ? col - 1 : 0) * 8u;
  const unsigned nx8 = (unsigned)nx * 8u;
  const bool has_ilo = __any(at_ilo), has_G = __any(isG);   // most strips hold neither end of the ring
  // rows: this workgroup owns U-rows ja..jb (and T-rows ja..jb, the last segment T-row jhi+1 as well); every level
  // walks T-rows jt0..jt1, level k two rows behind level k-1
  int ja0 = jlo + tyi * sa.seg_rows, jb0 = min(ja0 + sa.seg_rows - 1, jhi);
  if (sa.tiles) {
    ja0 = jlo + ja_rel;
    jb0 = jlo + jb_rel;
  } else if (sa.rows) {   // segments of unequal length (Evp::build_skew_rows): per tile of a block
    ja0 = jlo + sa.rows[2 * rem];
    jb0 = jlo + sa.rows[2 * rem + 1];
  }
  // Of its segment ja0 .. jb0 a workgroup walks the RUNS of rows that hold anything to compute (k_skew_rowact / k_skew_runs;
  // gaps of up to 3K rows are bridged: a new run costs the pipeline's fill and drain and its cone of redundant rows), each
  // as a sweep of its own.  Every wavefront finds the same runs (uniform), so all leave a run -- and the kernel -- together.
  // Nothing is lost: a cell without ice is never written (both copies of the state hold the same value there).
  const bool ra = S == 1 && sa.rowact != nullptr;
  const size_t ro = ((size_t)b * a.tiles_x + txi) * (size_t)(a.ny - 2);
  const int32_t* const rnext = sa.run_next + ro - jlo;
  const int32_t* const rend = sa.run_end + ro - jlo;
  // The workgroups a CU holds are copies of one program started at the same moment: left alone they stay IN PHASE --
  // all compute together (sharing the SIMDs), then all wait together (barrier, LDS, memory) -- and the waiting of one
  // hides behind nothing.  Workgroups are dealt to the CUs of an XCD in turn, so consecutive ones (blockIdx / 8) that
  // share a CU are stagger_mod apart in that count: each starts a fraction of a step later than the one before.
  if (sa.stagger_ticks > 0) {
    const long long wait = (long long)((blockIdx.x >> 3) % (unsigned)sa.stagger_mod) * sa.stagger_ticks;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(2);
  }
  if (sa.dbg && threadIdx.x == 0) sa.dbg[2 * blockIdx.x] = wall_clock64();
  stamp_at(sa.stamps, 0);
  PHASE_DECL
  for (int cursor = ja0; cursor <= jb0;) {
  int ja = cursor, jb = jb0;
  if (ra) {
    const int f = rnext[cursor];               // relative to jlo; uniform addresses: scalar loads
    if (f < 0 || jlo + f > jb0) break;         // nothing left in this segment
    ja = jlo + f;
    jb = min(jlo + rend[ja], jb0);
  }
  cursor = jb + 1;
  const int jt0 = max(jlo, ja - (K - 1)), jt1 = min(jhi + 1, jb + K);
  const bool lastlev = k == K - 1;
#if SKEW_TRIM
  // What a level has to compute shrinks by one row per level at either end of the segment (level K-1: T-rows ja..jb+1,
  // level k: ja-(K-1-k) .. jb+(K-k)): rows beyond that feed nothing.  A level whose first row lies above the block's
  // first row takes one light step before it -- the velocities and HTN of the row below, nothing computed (`pick`).
  // The last level ends 2(K-1) steps after level 0 would have walked ITS last row: K-1 steps fewer per sweep.
  const int lo = max(jlo, ja - (K - 1 - k)), hi = min(jhi + 1, jb + (K - k));
  const int nsteps = (min(jhi + 1, jb + 1) - jt0 + 1) + 2 * (K - 1);
#else
  const int lo = jt0, hi = jt1;
  const int nsteps = (jt1 - jt0 + 1) + 2 * (K - 1);
#endif

  // row below the first T-row: unchanged during the sweep where it matters (a ghost row), harmless elsewhere
  double us = c0, vs = c0, usw = c0, vsw = c0, hn_s = c0;
  if (col_ok) {
    const unsigned q = (unsigned)(jt0 - 2) * nx8 + co;
    ld_uv(q, us, vs);
    hn_s = ld8(htn, q);
    if (col >= 2) ld_uv(q - 8u, usw, vsw);
  }
  double p0 = c0, pe1 = c0, p4 = c0, pe6 = c0;   // str of the previous T-row: (i,j,1) (i+1,j,2) (i,j,5) (i+1,j,7)
  // what the NEXT step works on, fetched one step ahead (the stresses are not: they are wanted halfway through
  // `stress` only, level 0 fetches them at the top of the step itself and the others find them in LDS)
  double nun = c0, nvn = c0, nhn = c0, nhe = c0, nhew = c0, nst = c0;
  int ntm = 0, num = 0, um_prev = 0;   // icetmask / iceumask of the next row (as loaded); iceumask of the row below
  double s[12];                        // stresses of this step's row; handed on at the top of the NEXT step
#pragma unroll
  for (int c = 0; c < 12; ++c) s[c] = c0;

#pragma clang loop unroll(disable)
  for (int t = -1; t < nsteps; ++t) {
    // The SIMD issues from its OLDEST ready wavefront first: of the workgroups sharing a CU the first one dispatched
    // runs at its own pace, the last one gets what is left and finishes long after -- alone on the CU, bound by
    // latency.  Rotating the priority from step to step lets them advance together.
#if SKEW_LOADPRIO
    __builtin_amdgcn_s_setprio(3);   // the loads of a step go out first, whoever's turn it is in the rotation
#else
    if (sa.prio_rotate) {
      const int p = (t + 1 + gen) % sa.stagger_mod;
      if (p == 0) __builtin_amdgcn_s_setprio(3);
      else if (p == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const int r = jt0 + t - 2 * k;                 // T-row of this step (uniform)
    const bool act = r >= lo && r <= hi;
    const bool pick = SKEW_TRIM && k > 0 && lo > jlo && r == lo - 1;   // the row below this level's first row: taken over, not computed
    const unsigned q = (unsigned)(r - 1) * nx8 + co;   // T-cell (col, r); the U-cell of this step is (col, r-1) = q - nx8
    const bool urow = act && r > lo;                   // (jlo <= r-1 <= jhi holds then)
    // ---- the stresses of the previous row go to the next level now, not when they were formed: the slot they go
    // into was read by that level during the previous step (two slots, one barrier per step)
    if (!lastlev && r - 1 >= lo && r - 1 <= hi && !(EARLY && k == 0) && own_lane) {
#pragma unroll
      for (int c = 0; c < 12; ++c) s_sig[EARLY ? k - 1 : k][(r - 1) & 1][c][cw] = s[c];
    }
    // ---- take over what was fetched for this row.  The empty asm is a use of those registers placed BEFORE this
    // step's loads are issued: the wait for them that the compiler needs (it cannot count loads across the back edge
    // and waits for everything) then falls here, where only loads of the previous step are in flight.  (EARLY: that use
    // sits at the end of the previous step, in front of level 0's early loads.)
    if (!EARLY) asm volatile("" : "+v"(nun), "+v"(nvn), "+v"(nhn), "+v"(nhe), "+v"(nhew), "+v"(nst), "+v"(ntm), "+v"(num));
    const int r3 = (int)((unsigned)(r + 3 * 4096) % 3u);   // slot of row r in the three-slot hand-off of level 0
    // the planes of an array are walked with a scalar pointer (base += plane stride: scalar adds, no vector
    // arithmetic per access); the stride is made opaque once per step, or the compiler would form all the plane bases
    // before the loop and keep them in (spilled) scalar registers
    size_t pstride = n8;
#if SKEW_OPAQUE_STRIDE
    asm volatile("" : "+s"(pstride));
#endif
    double un = nun, vn = nvn;
    const double hn = nhn, he = nhe, St = nst;
    double hew = nhew;
    const int tm_cur = ntm, um_cur = num;
    if (TP && !lastlev && r >= jt0 && r <= jt1) {   // read by the next level during the NEXT step, as its prefetch
      s_tin[k][r & 1][0][lx] = hn;
      s_tin[k][r & 1][1][lx] = he;
      s_tin[k][r & 1][2][lx] = St;
      s_msk[k][r & 1][lx] = (tm_cur == 1 ? 1 : 0) | (um_cur != 0 ? 2 : 0);
    }
    // ---- loads, oldest first: the momentum inputs of the U-row below (read-only: L2 hits for the levels behind
    // level 0; fetched whatever the mask says so that they do not wait for it) ...
    double xa, xuo, xvo, xfx, xfy, xum, xfm, xur;
    if (urow && ucol) {
      const unsigned qu = q - nx8;
      if (WIDE) {
        size_t pstride2 = n16;
#if SKEW_OPAQUE_STRIDE
        asm volatile("" : "+s"(pstride2));
#endif
        const char* pu = (const char*)uar4;
        const dbl2 w0 = ld16((const double*)pu, 2u * qu); pu += pstride2;
        const dbl2 w1 = ld16((const double*)pu, 2u * qu); pu += pstride2;
        const dbl2 w2 = ld16((const double*)pu, 2u * qu); pu += pstride2;
        const dbl2 w3 = ld16((const double*)pu, 2u * qu);
        xa = w0.x; xuo = w0.y; xvo = w1.x; xfx = w1.y; xfy = w2.x; xum = w2.y; xfm = w3.x; xur = w3.y;
      } else {
        const char* pu = (const char*)uar;
        xa = ld8((const double*)pu, qu); pu += pstride;
        xuo = ld8((const double*)pu, qu); pu += pstride;
        xvo = ld8((const double*)pu, qu); pu += pstride;
        xfx = ld8((const double*)pu, qu); pu += pstride;
        xfy = ld8((const double*)pu, qu); pu += pstride;
        xum = ld8((const double*)pu, qu); pu += pstride;
        xfm = ld8((const double*)pu, qu); pu += pstride;
        xur = ld8((const double*)pu, qu);
      }
    }
    // ... what the next step starts with (straight-line code: a row index clamped into the sweep instead of a branch,
    // so that the number of loads in flight is the same on every path) ...
    {
      const int rn = min(max(r + 1, jt0), jt1);
      const unsigned qn = (unsigned)(rn - 1) * nx8 + co;
      if (k == 0) {
        if (PIN) {
          ld_uv(qn, nun, nvn);
        } else {
          nun = ld8(u_in, qn);
          nvn = ld8((const double*)((const char*)u_in + pstride), qn);
        }
      }
      if (TP && k > 0) {
        nhn = s_tin[k - 1][rn & 1][0][lx];
        nhe = s_tin[k - 1][rn & 1][1][lx];
        nst = s_tin[k - 1][rn & 1][2][lx];
        const int m = s_msk[k - 1][rn & 1][lx];
        ntm = m & 1;
        num = m & 2;
        if (has_ilo && (rn < jlo || rn > jhi)) nhew = ld8(hte, qn - (col >= 2 ? 8u : 0u));
      } else if (WIDE) {
        const dbl2 hh = ld16(hnhe, 2u * qn);
        nhn = hh.x;
        nhe = hh.y;
        // HTE west of ilo: on the block's own rows the value of column ihi, two lanes away (below); a ghost row reads the
        // ghost column as it lies (only the lane at ilo uses it)
        if (has_ilo && (rn < jlo || rn > jhi)) nhew = ld8(hte, qn - (col >= 2 ? 8u : 0u));
        nst = ld8(stren, qn);
        const int m = ld4(msk, qn >> 1);
        ntm = m & 1;
        num = m & 2;
      } else {
        nhn = ld8(htn, qn);
        nhe = ld8(hte, qn);
        if (has_ilo && (rn < jlo || rn > jhi)) nhew = ld8(hte, qn - (col >= 2 ? 8u : 0u));
        nst = ld8(stren, qn);
        ntm = ld4(tmk, qn >> 1);
        num = ld4(umk, qn >> 1);
      }
    }
    // ... and, youngest, the stresses of this row: whoever waits for them waits for everything, which has arrived by then
    if (act) {
      if (k == 0) {
        if (PIN) {
          size_t ps2 = n16;
#if SKEW_OPAQUE_STRIDE
          asm volatile("" : "+s"(ps2));
#endif
          const char* ps = (const char*)u_in + ps2;
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const dbl2 t = ld16((const double*)ps, 2u * q);
            s[2 * c] = t.x;
            s[2 * c + 1] = t.y;
            ps += ps2;
          }
        } else if (!EARLY) {
          const char* ps = (const char*)u_in + 2 * pstride;
#pragma unroll
          for (int c = 0; c < 12; ++c) {
            s[c] = ld8((const double*)ps, q);
            ps += pstride;
          }
        }   // (EARLY: fetched at the end of the previous step)
      } else if (EARLY && k == 1) {
#pragma unroll
        for (int c = 0; c < 12; ++c) s[c] = s_sig0[r3][c][lx];
      } else {
#pragma unroll
        for (int c = 0; c < 12; ++c) s[c] = s_sig[EARLY ? k - 2 : k - 1][r & 1][c][cw];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    PHASE(0)         // 0: hand-off written, this step's loads issued
#if SKEW_LOADPRIO
    if (sa.prio_rotate) {
      const int p = (t + 1 + gen) % sa.stagger_mod;
      if (p == 0) __builtin_amdgcn_s_setprio(2);
      else if (p == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else {
      __builtin_amdgcn_s_setprio(0);
    }
#endif
    if (act || pick) {
      const bool tact = act && tcol && tm_cur == 1;
      // the rows of the level below (a ghost row never changes: it is read where it lives)
      if (k > 0) {
        if (r > jhi) {
          if (col_ok) ld_uv(q, un, vn);
        } else {
          un = s_uv[k - 1][r & 1][0][cw];
          vn = s_uv[k - 1][r & 1][1][cw];
        }
      }
      double uw, vw;
      {
        // wave shifts are taken by every lane, the choice between them is per lane.  West of ilo sits G in the ring:
        // ilo's western neighbour ihi is two lanes away where the ghost columns mirror (rows jlo..jhi, every
        // level: the ghost columns of the state a launch starts from are current); on a ghost row it is the ghost
        // column's own, unchanged value
        uw = up1z(un);
        vw = up1z(vn);
        if (has_ilo) {
          const double uw2 = up1z(uw), vw2 = up1z(vw);
          if (r >= jlo && r <= jhi) {
            if (at_ilo) {
              uw = uw2;
              vw = vw2;
            }
          } else if (at_ilo) {
            ld_uv(q - 8u, uw, vw);
          }
        }
      }
      if (act) {
        // (the strips that hold the seam of the ring used to fetch HTE(ihi) for the lane at ilo with a load of their own
        //  per level and step: 46 memory instructions per workgroup and step instead of 42, and 11 % more time per row)
        const double hs = up1z(he);
        double hs2 = hs;
        if (has_ilo) hs2 = up1z(hs);
        if (!at_ilo) hew = hs;
        else if (r >= jlo && r <= jhi) hew = hs2;
      }
      const bool uact = urow && ucol && um_prev != 0;
      // ---- stress (ice_dyn_evp.F90:1065-1289)
      StressOut o;
#pragma unroll
      for (int c = 0; c < 8; ++c) o.str[c] = c0;
      if (tact) {
        const double Dxt = p5 * (hn + hn_s);                 // ice_grid.F90:1184 (dxt)
        const double Dyt = p5 * (he + hew);                  // :1271 (dyt)
        const double Dxhy = p5 * (he - hew);                 // :347
        const double Dyhx = p5 * (hn - hn_s);                // :348
        const double Cyp = 1.5 * he - p5 * hew;              // :354
        const double Cxp = 1.5 * hn - p5 * hn_s;             // :355
        const double Cym = -(1.5 * hew - p5 * he);           // :357
        const double Cxm = -(1.5 * hn_s - p5 * hn);          // :358
        const double Tiny = puny * (Dxt * Dyt);              // :334, :346
        const bool diag = LAST && lastlev;
        stress_cell<LAST, DAMP>(a.sc, un, uw, usw, us, vn, vw, vsw, vs, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp, Cxm, Cym,
                                diag ? ld8(a.tarear + base, q) : c0, Tiny, St, s, o, diag);
        const bool own_t = own_col && ((r >= ja && r <= jb) || (r == jhi + 1 && jb == jhi));
        if (lastlev && own_t) {
          if (POUT) {
            size_t ps2 = n16;
#if SKEW_OPAQUE_STRIDE
            asm volatile("" : "+s"(ps2));
#endif
            char* po = (char*)s_out + ps2;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              st16((double*)po, 2u * q, s[2 * c], s[2 * c + 1]);
              po += ps2;
            }
          } else {
            char* po = (char*)s_out + 2 * pstride;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
              st8((double*)po, q, s[c]);
              po += pstride;
            }
          }
          if (LAST) {
            st8(a.divu + base, q, o.divu);
            st8(a.rdg_conv + base, q, o.rdg_conv);
            st8(a.rdg_shear + base, q, o.rdg_shear);
            st8(a.shear + base, q, o.shear);
            st8(a.prs_sig + base, q, o.prs_sig);
          }
        }
      }
      PHASE(1)       // 1: stress (with the wait for its stresses and inputs)
      // ---- momentum of U-row r-1 (:1390-1435): str of (i,j) carried, of (i+1, .) by wave shift
      const double e1 = down1z(o.str[1]), e3 = down1z(o.str[3]), e6 = down1z(o.str[6]), e7 = down1z(o.str[7]);
      double u1 = us, v1 = vs;   // velocity of row r-1 after this level
      if (uact) {
        const double sx = p0 + pe1 + o.str[2] + e3;      // :1415-1416 order
        const double sy = p4 + o.str[5] + pe6 + e7;      // :1417-1418 order
        const int ru = r - 1;
        UIn x;
        x.aiu = xa; x.uocn = xuo; x.vocn = xvo; x.forcex = xfx; x.forcey = xfy; x.umassdtei = xum; x.fm = xfm;
        x.uarear = xur;
        water_of(xuo, xvo, xfm, x.waterx, x.watery);
        if (lastlev) {
          if (own_col && ru >= ja && ru <= jb) {
            if (sa.fwd_rule) {
              // the ghost cells that mirror this cell on this rank are known without the table: column ilo is
              // mirrored by ihi+1, column ihi by ilo-1, same row (the table look-up costs the two strips at the ends
              // of the ring four dependent memory round trips per row)
              StepuOut ro;
              stepu_cell(us, vs, x.aiu, x.uocn, x.vocn, x.waterx, x.watery, x.forcex, x.forcey, x.umassdtei, x.fm,
                         x.uarear, sx, sy, ro);
              const unsigned qu = q - nx8;
              st_uv(qu, ro.u, ro.v);
              if (LAST) {
                st8(a.strintx + base, qu, ro.strintx);
                st8(a.strinty + base, qu, ro.strinty);
                st8(a.strocnx + base, qu, ro.taux);
                st8(a.strocny + base, qu, ro.tauy);
              }
              if (cyc && (col == ilo || col == ihi)) {
                const unsigned qg = (unsigned)(ru - 1) * nx8 + (unsigned)(col == ilo ? ihi : ilo - 2) * 8u;
                st_uv(qg, ro.u, ro.v);
              }
            } else {
              stepu_store_o<LAST>(a, x, base, q - nx8, col, ru, ilo, ihi, jlo, jhi, us, vs, sx, sy);
            }
          }
        } else {
          StepuOut ro;
          stepu_cell(us, vs, x.aiu, x.uocn, x.vocn, x.waterx, x.watery, x.forcex, x.forcey, x.umassdtei, x.fm,
                     x.uarear, sx, sy, ro);
          u1 = ro.u;
          v1 = ro.v;
        }
      }
      if (!lastlev && urow) {
        // the east ghost column G mirrors column ilo = the next lane (rows whose velocity is updated, :397-402)
        if (has_G) {
          const double gu = down1z(u1), gv = down1z(v1);
          if (isG) {
            u1 = gu;
            v1 = gv;
          }
        }
        if (own_lane) {
          s_uv[k][(r - 1) & 1][0][cw] = u1;
          s_uv[k][(r - 1) & 1][1][cw] = v1;
        }
      }
      p0 = o.str[0]; pe1 = e1; p4 = o.str[4]; pe6 = e6;
      us = un; vs = vn; usw = uw; vsw = vw;
      hn_s = hn;
      um_prev = um_cur;
    }
    if (EARLY) {
      // what the next step starts with has arrived by now (fetched at the top of this step): the use that makes the
      // compiler wait for it stands HERE, in front of level 0's early loads
      asm volatile("" : "+v"(nun), "+v"(nvn), "+v"(nhn), "+v"(nhe), "+v"(nhew), "+v"(nst), "+v"(ntm), "+v"(num));
      if (k == 0) {
        if (act) {
#pragma unroll
          for (int c = 0; c < 12; ++c) s_sig0[r3][c][lx] = s[c];
        }
        if (r + 1 >= lo && r + 1 <= hi) {
          const char* ps = (const char*)u_in + 2 * pstride;
#pragma unroll
          for (int c = 0; c < 12; ++c) {
            s[c] = ld8((const double*)ps, q + nx8);
            ps += pstride;
          }
        }
      }
    }
    PHASE(2)         // 2: momentum, stores, hand-off of u, v
    __syncthreads();
    PHASE(3)         // 3: barrier
  }
  if (!ra) break;
  }   // runs
  stamp_at(sa.stamps, 1);
#ifdef CICE4_AMD_STAMPS
  ph_[6] = (long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
  ph_[7] = (long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID: where this wavefront ran (scripts/sweep_placement.py)
  if (sa.phases && lx == 0)
    for (int i_ = 0; i_ < 8; ++i_) sa.phases[8 * ((size_t)blockIdx.x * K + k) + i_] = ph_[i_];
#endif
  if (sa.dbg && threadIdx.x == 0) sa.dbg[2 * blockIdx.x + 1] = wall_clock64();
}

// ---- list-driven, unfused forms with the reference's argument lists (tests) --------------
// ---- the WHOLE subcycle loop in one launch, state resident in registers (grids of at most one tile per CU) --------
//
// At gx1 a launch of k_subcycle2 is 2.2 us of launch gap + 6.3 us of loads and stores + 2 x 4.3 us of arithmetic, the
// phases in series.  Everything but the velocity is private to a T-cell or read-only, so a
// workgroup that stays on its CU for all ndte subcycles keeps the 12 stresses, the metrics, the strength and the
// ten U-cell inputs of its lanes in registers, and only u, v on tile edges travel: the producing tile writes them to
// an exchange copy of (u, v) and raises its progress word; a tile starts subcycle k+1 when the tiles it reads from have
// reached k.  No launch gap, no re-load of sigma, no redundant second stress evaluation on a rim.
//
// Geometry = k_subcycle with R = 1: tile = 64 x W T-cells, tiles overlap by one T-row / T-column (recomputed, same
// bits); lane lx < 63 of wavefront w < W-1 owns U-cell (i0+lx, j0+w).  Velocities a lane does not produce itself --
// column 63 and row W-1 (owned by the east / north tile), ghost cells (mirrors, written by the owner of the mirrored
// cell), the row below wavefront 0 and the column west of lane 0 -- are re-read from the exchange copy every subcycle.
// Hand-off (MI355X_MICROARCH.md, "Valid forms", first row of the table): every published byte is an agent-scope
// (sc1) store, every storing wavefront drains vmcnt, workgroup barrier, ONE lane stores the progress word (sc1);
// the consumer's wavefront 0 polls the progress words of its producers (sc1 loads, one lane each), workgroup barrier,
// then every load of exchanged bytes is an sc1 load.  Exchange buffers are double-buffered by subcycle parity: the
// dependence is symmetric (a tile reads from every tile that reads from it), so no tile gets two subcycles ahead of a
// reader.  Every spin is bounded by wall-clock ticks; on time-out (a tile that is not resident) the abort word is
// raised, every workgroup leaves, nothing of the caller's state has been touched (inputs are read from st[cur],
// results go to st[1-cur] after the last subcycle) and the host falls back to the launch-per-pair loop.
constexpr int RES_MAXDEP = 16;
constexpr int RES_NPEER = 8;     // neighbouring ranks of a block in a cartesian layout (edges and corners)
constexpr int RES_STRIDE = 32;   // progress words 128 B apart

struct ResArgs {
  SubArgs a;             // u_in, v_in, sig_in = st[cur] (read once); u_out, v_out, sig_out = st[1-cur] (final result)
  int nsub, last;        // subcycles of this launch; last != 0: the final one is subcycle ndte (diagnostics)
  unsigned epoch0;       // progress of a tile after subcycle k of this launch = epoch0 + k + 1
  unsigned* prog;        // [tiles * RES_STRIDE]
  unsigned* abort_flag;
  const int32_t* deps;   // [tiles][RES_MAXDEP] producer tiles, -1 padded
  double* xu[2];         // exchange copies: u at 0, v at a.n; subcycle k publishes into xu[k & 1]
  long long spin_ticks;  // wall_clock64() ticks (100 MHz) a poll may take
  // PEER: the block is one slab of a domain cut across ranks; the tiles on its first / last rows exchange with tiles
  // of the neighbouring ranks through the SAME protocol, the neighbour's exchange copies and progress words being
  // mapped into this address space (Evp::peer_connect): side 0 = the rank to the south, 1 = to the north
  const int32_t* rslot;  // [cells] -1: nothing; -2: a ghost cell whose source lives on another rank; >= 0: slot in rfwd
  const int32_t* rfwd;   // [slots][4] ghost cells on other ranks that mirror a cell: (neighbour << 28) | address, -1 padded
  double* pxu[RES_NPEER][2];   // [neighbour][parity]: the neighbour's exchange copies (v at + pn[neighbour])
  unsigned pn[RES_NPEER];      // the neighbour's plane size
  unsigned* prp[RES_NPEER];    // [neighbour]: where this rank's tiles publish their progress on it ([tile * RES_STRIDE])
  const unsigned* rprog; // progress words the neighbours' tiles publish here: deps <= -2 index it (-2 - dep)
  const int32_t* pub;    // [tiles] bit s: cells of this tile are mirrored on neighbour s
  // FOLD: a tripole north boundary on a one-block, one-rank domain (serial/ice_boundary.F90:705-869) for the velocity
  // (NE-corner location, vector kind).  The ghost cells the fold fills mirror, up to the sign, the final value of an owned
  // cell: they ride on rslot / rfwd (entries: (negate << 30) | address in THIS block).  The owned cells of the top row are
  // changed by the fold itself -- symmetric average with the partner across the pole, or a mirror image, or a sign
  // flip -- from the RAW values of the subcycle, which the tiles of the top row hand to each other in a phase of their
  // own (xraw, prog2, deps2) before anything is published.
  const int32_t* fslot;  // [cells] / [slots][4]: ghost cells of THIS block the fold fills from a cell -- as rslot / rfwd, entries
  const int32_t* ffwd;   //   (negate << 30) | address (tables of their own: across ranks rslot / rfwd name the neighbours' cells)
  const int32_t* ftab;   // [2 * cells] owned top-row cells: partner address (-1: none) and mode bits (F_*)
  double* xraw[2];       // [parity] raw top-row velocities (u at 0, v at a.n)
  unsigned* prog2;       // [tiles * RES_STRIDE] raw top row of subcycle k published = epoch0 + k + 1
  const int32_t* deps2;  // [tiles][4] tiles that hold the partners of this tile's top-row cells, -1 padded
  long long* stamps;     // -DCICE4_AMD_STAMPS (diagnostic build only): [4 * workgroups] see stamp_begin / stamp_end
  long long* phases;     // ... [8 * workgroups] cycles per phase of the subcycle, summed over the launch (PHASE)
  int prio_mode;         // issue priority of the workgroups that share a CU (dense shape): 0 none, 1 by generation, 2 rotating
  int prio_div;          // workgroups of one generation per XCD (= CUs per XCD)
  int prio_top;          // FOLD, dense shape: the tiles of the top row always issue first on their CU
  const int32_t* tile_map;   // one word (k_res_choose_map): which tile a workgroup takes, 0 = the tiles of an XCD are neighbours, 1 = blockIdx
  // GRAN: the hand-off by data-tagged granules (see k_evp_resident)
  const int32_t* src;    // [cells] the owned U-cell whose velocity a cell holds: itself, or the source of a ghost cell; -1: nobody's (constant during the loop)
  void* xg;              // two copies (subcycle parity) of [cells][2][2] granules {low half | tag, high half | tag} of u, then of v: 32 bytes per cell
  unsigned xg_half;      // bytes of one copy
  void* xgr;             // GRAN && FOLD: the same for the RAW velocities of the top row (the fold's own hand-off, before anything is published)
  int poll_delay, poll_sleep;   // GRAN: s_sleep(8) units (~0.2 us each) before the first poll of a subcycle / between two polls
  int fake_ew;                  // TIMING EXPERIMENT ONLY (wrong results): the wavefronts between the first and the last row never poll
};
enum { F_LO = 1, F_HI = 2, F_NEG = 4, F_SELF = 8, F_MIRROR = 16 };

// agent-scope (sc1) access at a uniform base + 32-bit byte offset: SGPR base + VGPR offset addressing, no 64-bit
// address arithmetic per access (the exchange copies of a one-block domain are far below 4 GB)
__device__ __forceinline__ double ld_agent(const double* base, unsigned off) {
  return __hip_atomic_load((double*)((char*)const_cast<double*>(base) + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* base, unsigned off, double v) {
  __hip_atomic_store((double*)((char*)base + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same across devices (a neighbouring rank's memory over xGMI, or its writes into ours): system scope
__device__ __forceinline__ double ld_sys(const double* base, unsigned off) {
  return __hip_atomic_load((double*)((char*)const_cast<double*>(base) + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(double* base, unsigned off, double v) {
  __hip_atomic_store((double*)((char*)base + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// GRAN hand-off: the data is the flag (cdna_hip_programming.md, Guideline 16, recipe R2).  A velocity component travels as
// two naturally aligned 8-byte granules {32 bits of the value, 32-bit tag = subcycle epoch}, written together by ONE 16-byte
// write-through (sc1) store -- each 8-byte half is untorn -- and read by ONE 16-byte sc1 load (L1 bypassed); the reader
// polls the very bytes it needs until both tags carry the epoch it waits for.  No drain, no progress word, no flag: one
// one-way trip through memory instead of store -> drain -> flag -> poll -> load.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_gran(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned soff, double x, unsigned tag) {
  u32x4 g;
  g.x = (unsigned)__double2loint(x); g.y = tag; g.z = (unsigned)__double2hiint(x); g.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)off, (int)soff, 16);   // aux 16 = sc1
  // The store reads its four data registers over several cycles, and the compiler (ROCm 7.2) holds that a buffer store
  // with an SGPR soffset needs no wait state before they are written again: on gfx950 it does.  Under register pressure the
  // u and the v granule are built in the SAME four registers, and `v_cndmask v20` one instruction after `buffer_store_dwordx4
  // v[18:21] ... s20` put the high word of v into the u granule -- tag right, value wrong, one run in twenty
  // (profiles/r05_resident_granules.txt, section 11).  The registers stay live, and untouched, through two wait states.
  asm volatile("s_nop 1" : : "v"(g) : "memory");
}
__device__ __forceinline__ u32x4 ld_gran(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)soff, 16);
}
#ifdef CICE4_AMD_STAMPS
// diagnostic build: the low 16 bits of a tag carry the 100 MHz wall clock of the moment the granule was stored (the epoch
// keeps the high 16): the reader learns how long the hand-off took (scripts/resident_phases.py)
__device__ __forceinline__ unsigned gran_tag(unsigned epoch) { return (epoch << 16) | ((unsigned)wall_clock64() & 0xffffu); }
__device__ __forceinline__ bool gran_ok(const u32x4& a, const u32x4& b, unsigned tag) {
  return ((a.y ^ tag) >> 16) == 0 && a.w == a.y && b.y == a.y && b.w == a.y;
}
#else
__device__ __forceinline__ unsigned gran_tag(unsigned epoch) { return epoch; }
__device__ __forceinline__ bool gran_ok(const u32x4& a, const u32x4& b, unsigned tag) {
  return a.y == tag && a.w == tag && b.y == tag && b.w == tag;
}
#endif
__device__ __forceinline__ double gran_val(const u32x4& a) { return __hiloint2double((int)a.z, (int)a.x); }

// Which tile map of k_evp_resident leaves the busiest CU with fewer tiles that hold ice (one workgroup, once per evp(dt)).
// A workgroup blockIdx b of the loop runs on XCD b & 7, CU (b >> 3) mod per_xcd; under map 0 it takes tile (b & 7) * chunk +
// (b >> 3), under map 1 tile b.  force: -1 choose, 0 / 1 that map.
__global__ __launch_bounds__(1024) void k_res_choose_map(int nt, int tiles_x, int tiles_y, int W, int nx, int ny, int per_xcd,
                                                         int force, const int32_t* __restrict__ blk,
                                                         const int32_t* __restrict__ tmk, const int32_t* __restrict__ umk,
                                                         int32_t* __restrict__ out, unsigned* __restrict__ cover_word) {
  __shared__ int s_ice[1024];
  __shared__ int s_cnt[2][256];
  __shared__ int s_max[2];
  const int t = threadIdx.x;
  if (t < 256) s_cnt[0][t] = s_cnt[1][t] = 0;
  if (t < 2) s_max[t] = 0;
  int ice = 0;
  if (t < nt) {
    const int per_blk = tiles_x * tiles_y, b = t / per_blk, rem = t - b * per_blk, tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ilo = blk[6 * b], ihi = blk[6 * b + 1], jlo = blk[6 * b + 2], jhi = blk[6 * b + 3];
    const int i0 = ilo + txi * (TX - 1), j0 = jlo + tyi * (W - 1);
    for (int j = j0; j <= min(j0 + W - 1, jhi + 1) && !ice; ++j)
      for (int i = i0; i <= min(i0 + TX - 1, ihi + 1); ++i) {
        const size_t q = (size_t)b * nx * ny + (size_t)(j - 1) * nx + (i - 1);
        if (tmk[q] == 1 || umk[q] != 0) { ice = 1; break; }
      }
  }
  s_ice[t] = ice;
  __syncthreads();
  const int chunk = (nt + 7) >> 3;
  if (t < 8 * chunk && per_xcd * 8 <= 256) {     // t as a blockIdx of the loop
    const int cu = (t & 7) * per_xcd + ((t >> 3) % per_xcd);
    const int t0 = (t & 7) * chunk + (t >> 3);
    if (t0 < nt && s_ice[t0]) atomicAdd(&s_cnt[0][cu], 1);
    if (t < nt && s_ice[t]) atomicAdd(&s_cnt[1][cu], 1);
  }
  __syncthreads();
  if (t < 256) {
    atomicMax(&s_max[0], s_cnt[0][t]);
    atomicMax(&s_max[1], s_cnt[1][t]);
  }
  __syncthreads();
  if (t == 0) out[0] = force >= 0 ? force : (s_max[1] < s_max[0] ? 1 : 0);
  // how many tiles hold ice at all (read by the host behind the loop: the NEXT evp(dt) picks its shape by it, run_resident)
  // (into the last of the eight words the host reads behind every loop anyway: tiles with ice << 16 | tiles)
  const int cnt = __syncthreads_count(ice);
  if (t == 0 && cover_word) *cover_word = ((unsigned)cnt << 16) | (unsigned)nt;
}

// PEER: see ResArgs.  Differences to the one-rank loop: progress runs epoch0 + 1 ("this launch has begun: its exchange
// copies are initialised", published before anything else and waited for before the first remote store), epoch0 + 2 + k
// after subcycle k; the LAST subcycle is exchanged as well, so that every tile ends up holding the final velocity of
// its halo cells and writes the ghost cells owned by other ranks into the result itself (no halo update afterwards);
// everything another rank writes or reads is a system-scope access.
// GRAN (one rank, no fold): the edge velocities travel as data-tagged granules (st_gran / ld_gran above).  A lane whose
// velocity -- or whose southern / western neighbour's -- is produced elsewhere polls exactly those cells; cells nobody
// produces (a ghost cell without a source: beyond an open / closed edge, facing an eliminated block) are not polled and
// keep the value they were loaded with.  No workgroup barrier inside the loop (rows travel through LDS behind a flag word per
// wavefront: see the loop), no dependency lists, no progress words.  The exchange copies are double-buffered by subcycle parity exactly as before (a tile cannot publish subcycle k + 2
// before every reader of its subcycle k has published k + 1, i.e. has read k), and tags only ever grow, across launches too.
template <int W, bool DAMP, bool PEER, bool FOLD = false, bool GRAN = false>
// (second bound: wavefronts per SIMD.  W = 4 runs three workgroups per CU in the dense shape -- one rank only --, W = 11,
// 12 put three wavefronts of one workgroup on a SIMD: both need the 168-register budget whatever the compiler would
// like to use)
__global__ __launch_bounds__(64 * W, (W == 4 && !PEER ? 3 : (64 * W + 255) / 256)) void k_evp_resident(const ResArgs r) {
  // (PEER && FOLD: the rank that holds the top slab of a tripole grid cut into full-width slabs -- its fold is its own affair,
  //  handled exactly as on one rank, its southern neighbour is reached as in any cross-rank loop)
  static_assert(!(GRAN && PEER), "granule hand-off: one rank");
  const SubArgs& a = r.a;
  __shared__ double s_uv[W][2][TX];
  __shared__ double s_edge[W][4][TX];
  __shared__ double s_x[W][8][TX];
  __shared__ double s_m[W][10][TX];             // nine metrics + strength
  __shared__ int s_fd[W][3][TX];
  __shared__ int s_rfd[PEER ? W : 1][4][TX];   // PEER: ghost cells on other ranks mirroring this lane's cell
  int rfr[4] = {-1, -1, -1, -1};                // FOLD: ghost cells the fold fills from this lane's cell (registers: with them in LDS three
                                                //       4-wavefront workgroups need 165 KB of a CU's 160)
  __shared__ int s_pub[RES_NPEER];              // PEER: this tile publishes to neighbour s
  __shared__ int s_abort;
  __shared__ int s_uf[GRAN ? W : 1], s_sf[GRAN ? W : 1];   // GRAN: subcycles whose row of u | v (s_uv) / of str (s_edge) a wavefront has put into LDS
  // tiles are numbered block by block (a one-rank domain of several blocks: every block is cut into tiles_x x tiles_y
  // tiles of the largest block's extent; the cells of a block are addressed from its own plane, which is what the
  // forwarding lists and the dependency lists use as well -- the FOLD form has one block)
  const int per_blk = a.tiles_x * a.tiles_y;
  const int nt = per_blk * (FOLD ? 1 : a.nblocks);
  const int chunk = (nt + 7) >> 3;
  // Which tile: the workgroups of an XCD are blockIdx & 7 == XCD, and the three that share a CU are 256 apart in blockIdx.
  // Map 0 gives an XCD a band of neighbouring tile rows (and a CU three tiles a few rows apart); map 1 (tile = blockIdx)
  // gives a CU three tiles a third of the grid apart -- under an ice cover that comes in latitude bands a CU then holds
  // one tile with ice and two without instead of three of a kind: gx1 size with ice on two polar caps 4.2 us per subcycle
  // against 5.45, fully covered 5.51 against 5.44 (profiles/r04_resident_tile_map.txt).  k_res_choose_map picks, once per
  // evp(dt), the map under which the busiest CU holds fewer tiles with ice.  Everything between tiles goes by TILE number.
  const int tile = (r.tile_map && *r.tile_map == 1) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile >= nt) return;  // whole workgroup
  const int b = FOLD ? 0 : tile / per_blk;
  const int rem = tile - b * per_blk;
  const int tyi = rem / a.tiles_x, txi = rem - tyi * a.tiles_x;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  const int i0 = ilo + txi * (TX - 1), j0 = jlo + tyi * (W - 1);
  if (i0 > ihi || j0 > jhi) return;   // a block smaller than the largest one (padded decomposition): no cell to own here, nobody waits for this tile
  const int lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nx = a.nx;
  const int i = i0 + lx, j = j0 + w;
  const bool in_i = i <= ihi + 1;
  const bool ok = in_i && j <= jhi + 1;
  const size_t q = ok ? (size_t)b * nx * a.ny + (size_t)(j - 1) * nx + (i - 1) : 0;
  const unsigned qb = (unsigned)q * 8u, nxb = (unsigned)nx * 8u;   // byte offsets into an exchange copy
  const bool own_i = (min(i, ihi) - i0) < (TX - 1);
  const bool uown = lx < TX - 1 && i <= ihi && w < W - 1 && j <= jhi;
  const bool uact = uown && a.iceumask[q];
  const bool foreign = ok && !uown;                 // velocity of this lane's own position comes from elsewhere
  const bool tact = ok && a.icetmask[q] == 1;
  const bool sown = tact && own_i && (min(j, jhi) - j0) < (W - 1);
  // the velocity south of this lane comes from the wavefront below through LDS where that wavefront owns it, else
  // (row j0-1, or a column owned by the east tile / the ghost column) from the exchange copy
  const bool south_h = ok && (w == 0 || !(lx < TX - 1 && i <= ihi));
  const bool west_h = lx == 0 && ok;                // column i0-1 (>= 1): west and south-west of lane 0
  // a late workgroup of an aborted launch leaves at once
  if (GRAN && lx == 0) s_uf[w] = s_sf[w] = 0;
  if (threadIdx.x == 0) {
    s_abort = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PEER) {   // by geometry, whether or not the cells carry ice: the neighbour's tiles wait for this tile's progress
      for (int sd = 0; sd < RES_NPEER; ++sd) s_pub[sd] = (r.pub[tile] >> sd) & 1;
    }
  }
  __syncthreads();
  if (s_abort) return;

  // ---- resident state ----
  double un = c0, vn = c0, us = c0, vs = c0, uwh = c0, vwh = c0, uswh = c0, vswh = c0;
  if (ok) {
    un = a.u_in[q];
    vn = a.v_in[q];
  }
  if (south_h) {
    us = a.u_in[q - nx];
    vs = a.v_in[q - nx];
  }
  if (west_h) {
    uwh = a.u_in[q - 1];
    vwh = a.v_in[q - 1];
    uswh = a.u_in[q - nx - 1];
    vswh = a.v_in[q - nx - 1];
  }
  double s[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) s[c] = tact ? a.sig_in[(size_t)c * a.n + q] : c0;
  // everything that is read once per subcycle waits in LDS: registers are for the stresses
  if (tact) {   // the nine metrics (from memory: re-deriving them from HTN / HTE every subcycle cost 2 %) and the strength
    s_m[w][0][lx] = a.dxt[q]; s_m[w][1][lx] = a.dyt[q]; s_m[w][2][lx] = a.dxhy[q]; s_m[w][3][lx] = a.dyhx[q];
    s_m[w][4][lx] = a.strength[q];
    s_m[w][5][lx] = a.cxp[q]; s_m[w][6][lx] = a.cyp[q]; s_m[w][7][lx] = a.cxm[q]; s_m[w][8][lx] = a.cym[q];
    s_m[w][9][lx] = a.tinyarea[q];
  }
  // the eight read-only inputs of the momentum equation wait in LDS (read back by the same lane once per subcycle)
  if (uact) {
    s_x[w][0][lx] = a.aiu[q]; s_x[w][1][lx] = a.uocn[q]; s_x[w][2][lx] = a.vocn[q]; s_x[w][3][lx] = a.forcex[q];
    s_x[w][4][lx] = a.forcey[q]; s_x[w][5][lx] = a.umassdtei[q]; s_x[w][6][lx] = a.fm[q]; s_x[w][7][lx] = a.uarear[q];
  }
  // velocities other tiles read: the tile's first / last owned row and column, and cells that ghosts mirror
  bool edge = false;
  {
    int fd0 = -1, fd1 = -1, fd2 = -1;
    if ((GRAN ? uown : uact) && a.ring_slot) {
      const int slot = a.ring_slot[q];
      if (slot >= 0) {
        fd0 = a.fwd[3 * slot];
        fd1 = a.fwd[3 * slot + 1];
        fd2 = a.fwd[3 * slot + 2];
      }
    }
    s_fd[w][0][lx] = fd0; s_fd[w][1][lx] = fd1; s_fd[w][2][lx] = fd2;
    // (GRAN: every owned edge cell is published, with or without ice -- see `produced` below)
    edge = (GRAN ? uown : uact) && (lx == 0 || lx == TX - 2 || w == 0 || w == W - 2 || i == ihi || j == jhi || fd0 >= 0);
    if (PEER) {
      int rf[4] = {-1, -1, -1, -1};
      if (uact) {
        const int rs = r.rslot[q];
        if (rs >= 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) rf[c] = r.rfwd[4 * rs + c];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) s_rfd[w][c][lx] = rf[c];
      edge = edge || rf[0] >= 0;
    }
    if (FOLD) {
      if (uown) {
        const int rs = r.fslot[q];
        if (rs >= 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) rfr[c] = r.ffwd[4 * rs + c];
        }
      }
      edge = edge || rfr[0] >= 0;
    }
  }
  // FOLD: the owned cells of the top row (with or without ice: the average with an ice-covered partner is not zero)
  const bool topc = FOLD && uown && j == jhi;
  const bool toptile = FOLD && j0 + (W - 2) >= jhi;          // this tile owns cells of the top row (uniform)
  int fpad = -1, fmode = 0;
  if (topc) {
    fpad = r.ftab[2 * q];
    fmode = r.ftab[2 * q + 1];
    edge = true;
  }
  // PEER: a ghost cell of this block whose source lives on another rank: some tile has to write its final value
  const bool rghost = PEER && ok && !uown && r.rslot[q] == -2;
  // GRAN: which of the velocities this lane takes from elsewhere have a producer (an owned U-cell, wherever it lies).
  // A lane polls at most two cells -- slot A: its own position, slot B: its southern neighbour; the western and the
  // south-western neighbour of lane 0 are polled by two OTHER lanes in a slot they have free (a lane outside the block has
  // both free; in a full row lanes 1 .. 62 are not foreign in wavefront 0 and take their south from LDS in the others), and
  // handed to lane 0 by v_readlane: 16 registers of loads in flight instead of 32.
  bool need_f = false, need_s = false;
  const unsigned qg = (unsigned)q * 32u, nxg = (unsigned)nx * 32u;     // byte offsets into a granule copy
  unsigned offA = qg, offB = qg - nxg;
  bool nA = false, nB = false;
  int lw = -1, lq = -1;              // (uniform) the lanes that poll lane 0's western / south-western cell, -1: not needed
  bool lw_a = false, lq_a = false;   // ... in their slot A (else B)
  if (GRAN) {
    // (by GEOMETRY, with or without ice: who reads from whom has to be symmetric -- a tile that reads nothing from a
    //  neighbour that reads from it could get two subcycles ahead of that reader and overwrite a granule it still needs)
    auto produced = [&](size_t cell) { return r.src[cell] >= 0; };
    need_f = foreign && produced(q);
    need_s = south_h && produced(q - nx);
    nA = need_f;
    nB = need_s;
    const bool nw_ = west_h && produced(q - 1), nq_ = west_h && produced(q - nx - 1);
    const unsigned q0g = (unsigned)__builtin_amdgcn_readfirstlane((int)qg);    // lane 0's cell
    const unsigned long long want_w = __ballot(nw_), want_q = __ballot(nq_);   // (bit 0 or nothing)
    auto lend = [&](unsigned off, int& lane, bool& in_a) {
      const unsigned long long fa = __ballot(!nA && lx != 0), fb = __ballot(!nB && lx != 0);
      in_a = fa != 0;
      lane = (int)__builtin_ctzll(in_a ? fa : (fb ? fb : 1ull));   // (a lender always exists: see above; lane 0 of nothing otherwise)
      if (lx == lane && lane != 0) {
        if (in_a) { offA = off; nA = true; }
        else { offB = off; nB = true; }
      }
    };
    if (want_w) lend(q0g - 32u, lw, lw_a);
    if (want_q) lend(q0g - nxg - 32u, lq, lq_a);
    if (r.fake_ew && w > 0 && w < W - 1) {   // (timing experiment: what the loop would cost if interior wavefronts had nothing to wait for)
      nA = nB = false;
      lw = lq = -1;
    }
  }
  StressOut o;
  StepuOut ro{};

  // three barriers per subcycle: (C) str rows, (D) stores drained before the progress word, (E) progress of the
  // producers seen -- (E) also publishes the wavefronts' own velocity rows (s_uv) to the wavefront above
  s_uv[w][0][lx] = un;
  s_uv[w][1][lx] = vn;
  __syncthreads();
  if (PEER) {   // this launch has begun (its exchange copies were initialised before it started)
    const unsigned begun = r.epoch0 + 1u;
    if (threadIdx.x == 0)
      __hip_atomic_store(r.prog + (size_t)tile * RES_STRIDE, begun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < RES_NPEER && s_pub[threadIdx.x])
      __hip_atomic_store(r.prp[threadIdx.x] + (size_t)tile * RES_STRIDE, begun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  stamp_at(r.stamps, 0);
  PHASE_DECL
  // The workgroups a CU holds (dense shape: three) are copies of one program that depend on their neighbours once per
  // subcycle: left alone they end up IN PHASE -- all three compute together, each at a third of the SIMD, then all
  // three wait for their hand-off together (profiles/r04_resident_phases.txt: 1.4 us of stress for 0.5 us of issue, then
  // 2.4 us in the poll).  Workgroups are dispatched in blockIdx order, one per CU of an XCD before any CU gets its second:
  // the "generation" of a workgroup (0: first on its CU) is also a band of neighbouring tile rows.  Priority by
  // generation lets one band compute at full speed while the other two are in their hand-off.
  const int gen = r.prio_div > 0 ? min(2, (int)(blockIdx.x >> 3) / r.prio_div) : 0;
  if (!GRAN && r.prio_mode == 1) {
    if (gen == 0) __builtin_amdgcn_s_setprio(3);
    else if (gen == 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (GRAN) {
    // ---- the free-running loop: no workgroup barrier, every wavefront waits for exactly what IT needs ----
    // Inside the workgroup a row travels through LDS behind a flag word per wavefront: s_uf[w] = subcycles whose u | v row
    // wavefront w has written to s_uv[w] (read by the wavefront above for its next stress), s_sf[w] = subcycles whose str row
    // it has written to s_edge[w] (read by the wavefront below for its momentum).  Neither array needs a second copy: the
    // writer of s_edge[w] needs u | v of subcycle k from the wavefront below before it can write again, and that wavefront
    // publishes it only after it has read s_edge[w]; the writer of s_uv[w] needs str of the next subcycle from the wavefront
    // above, which reads s_uv[w] first.  Between workgroups: granules (above).  While a wavefront waits for its granules the
    // other wavefronts of its SIMD compute: the hand-off latency no longer adds to the arithmetic of three wavefronts.
    // Every wait is bounded; a wavefront that gives up raises s_abort and the abort word and every wavefront leaves.
#ifndef GRAN_COPIES
#define GRAN_COPIES 2
#endif
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(r.xg, 0, GRAN_COPIES * r.xg_half, 0x00020000);
    const bool prio_work = r.prio_mode >= 5;     // issue priority by what a wavefront is doing: waiting 0, computing 2 (chain rows 3 in mode 5)
    const bool chain_hi = r.prio_mode == 5 && (w == 0 || w >= W - 2);
    auto prio_wait = [&]() { if (prio_work) __builtin_amdgcn_s_setprio(0); };
    auto prio_compute = [&]() {
      if (prio_work) {
        if (chain_hi) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(2);
      }
    };
    auto wait_row = [&](int* flag, int want) -> bool {
      long long t0 = 0;
      for (int it = 0;; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
        if (it == 0) prio_wait();
        if (__hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
        if (it == 0) t0 = wall_clock64();
        else if ((it & 63) == 0 && wall_clock64() - t0 > 2 * r.spin_ticks) {   // (twice: whoever waits for memory reports first)
          if (lx == 0 && __hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            r.abort_flag[1] = 5u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = 0u;     // wait 5 = a row in LDS
            r.abort_flag[4] = (unsigned)w; r.abort_flag[5] = (unsigned)want; r.abort_flag[6] = 0u;
          }
          __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          return false;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");   // (the rows are read after the flag)
      prio_compute();
      return true;
    };
    auto post_row = [&](int* flag, int value) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wavefront's LDS writes are done
      if (lx == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // issue priority: the wavefronts on the chain tile -> neighbouring tile -> tile (rows 0, W-2, W-1) first
    TRACE_DECL
    const bool chain = w == 0 || w >= W - 2;
    if (r.prio_mode == 1) {
      if (chain) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(1);
    }
    prio_compute();
#pragma clang loop unroll(disable)
    for (int k = 0; k < r.nsub; ++k) {
      const bool lastk = r.last && k == r.nsub - 1;
      if (r.prio_mode == 2) {
        const int p = (k + gen) % 3;
        if (p == 0) __builtin_amdgcn_s_setprio(3);
        else if (p == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      } else if (r.prio_mode == 3) {
        // whoever is BEHIND on its SIMD issues first: the wavefronts of a workgroup go to the SIMDs round-robin, w, w + 4, w + 8
        // share one; s_uf[] says how many subcycles each of them has finished
        int behind = 0, level = 0;
        for (int o = w & 3; o < W - 1; o += 4)
          if (o != w) {
            const int done = __hip_atomic_load(&s_uf[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            behind += done < k;
            level += done == k;
          }
        if (behind) __builtin_amdgcn_s_setprio(0);
        else if (level) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      } else if (r.prio_mode == 4) {
        const int p = (k + gen) & 3;
        if (p == 0) __builtin_amdgcn_s_setprio(3);
        else if (p == 1) __builtin_amdgcn_s_setprio(2);
        else if (p == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      PHASE(7)
      TRACE(0)
      // (A) velocities of the row below (LDS, from the wavefront below) and of the western neighbour
      if (w > 0) {
        if (!wait_row(&s_uf[w - 1], k)) return;
        if (!south_h) {
          us = s_uv[w - 1][0][lx];
          vs = s_uv[w - 1][1][lx];
        }
      }
      TRACE(1)
      double uw = up1(un), vw = up1(vn), usw = up1(us), vsw = up1(vs);
      if (lx == 0) {
        uw = uwh;
        vw = vwh;
        usw = uswh;
        vsw = vswh;
      }
      // (B) stress (ice_dyn_evp.F90:1065-1289)
#pragma unroll
      for (int c = 0; c < 8; ++c) o.str[c] = c0;
      if (tact) {
        const double Dxt = s_m[w][0][lx], Dyt = s_m[w][1][lx], Dxhy = s_m[w][2][lx], Dyhx = s_m[w][3][lx],
                     Cxp = s_m[w][5][lx], Cyp = s_m[w][6][lx], Cxm = s_m[w][7][lx], Cym = s_m[w][8][lx],
                     Tiny = s_m[w][9][lx];
        stress_cell<true, DAMP>(a.sc, un, uw, usw, us, vn, vw, vsw, vs, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp, Cxm, Cym,
                                lastk ? a.tarear[q] : c0, Tiny, s_m[w][4][lx], s, o, lastk);
      }
      const double e1 = down1(o.str[1]), e3 = down1(o.str[3]), e6 = down1(o.str[6]), e7 = down1(o.str[7]);
      if (w > 0) {
        s_edge[w][0][lx] = o.str[2];
        s_edge[w][1][lx] = e3;
        s_edge[w][2][lx] = o.str[5];
        s_edge[w][3][lx] = e7;
      }
      post_row(&s_sf[w], k + 1);
      TRACE(2)
      PHASE(0)       // 0: row below seen + stress
      // (C) momentum (:1390-1435): str of the eastern neighbour by wave shift, of the row above through LDS
      if (w < W - 1) {
        if (!wait_row(&s_sf[w + 1], k + 1)) return;
        TRACE(3)
        PHASE(1)     // 1: str row of the wavefront above seen
        if (uact) {
          const double sx = o.str[0] + e1 + s_edge[w + 1][0][lx] + s_edge[w + 1][1][lx];   // :1415-1416 order
          const double sy = o.str[4] + s_edge[w + 1][2][lx] + e6 + s_edge[w + 1][3][lx];   // :1417-1418 order
          const double uocn = s_x[w][1][lx], vocn = s_x[w][2][lx];
          double wx, wy;
          water_of(uocn, vocn, s_x[w][6][lx], wx, wy);
          stepu_cell(un, vn, s_x[w][0][lx], uocn, vocn, wx, wy,
                     s_x[w][3][lx], s_x[w][4][lx], s_x[w][5][lx], s_x[w][6][lx], s_x[w][7][lx], sx, sy, ro);
          un = ro.u;
          vn = ro.v;
        }
      }
      const unsigned par = (unsigned)(k & (GRAN_COPIES - 1)) * r.xg_half;
      if constexpr (FOLD) {
        // (C') the fold changes the owned cells of the top row (the halo update after stepu, ice_dyn_evp.F90:397-402, on the
        // degenerate row; serial/ice_boundary.F90:705-869): their RAW velocities of this subcycle travel as granules of
        // their own, every lane that has a partner across the pole polls that one cell -- in the last subcycle too
        if (__any(topc)) {
          // (FOUR copies by subcycle, not two: a mirror image is read by a cell its source does not read back, so the two
          //  tiles are held together only through the tiles between them)
          const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(r.xgr, 0, 4 * r.xg_half, 0x00020000);
          const unsigned tag2 = gran_tag(r.epoch0 + (unsigned)k + 1u);
          const unsigned par4 = (unsigned)(k & 3) * r.xg_half;
          if (topc) {
            st_gran(rsr, qg, par4, un, tag2);
            st_gran(rsr, qg + 16u, par4, vn, tag2);
          }
          bool pp = topc && fmode != 0 && !(fmode & F_SELF);
          u32x4 p0 = {0u, 0u, 0u, 0u}, p1 = p0;
          if (__any(pp)) {
            prio_wait();
            const long long t0 = wall_clock64();
            int it = 0;
            while (true) {
              if (pp) { p0 = ld_gran(rsr, (unsigned)fpad * 32u, par4); p1 = ld_gran(rsr, (unsigned)fpad * 32u + 16u, par4); }
              if (pp && gran_ok(p0, p1, tag2)) pp = false;
              if (!__any(pp)) break;
              asm volatile("" ::: "memory");
              int bad = __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (!bad && (++it & 7) == 0) bad = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (!bad && wall_clock64() - t0 > r.spin_ticks) {
                if (lx == (int)__builtin_ctzll(__ballot(pp)) &&
                    __hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                  r.abort_flag[1] = 6u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = (unsigned)k;   // wait 6 = a raw granule across the fold
                  r.abort_flag[4] = (unsigned)w; r.abort_flag[5] = (unsigned)lx; r.abort_flag[6] = tag2;
                }
                bad = 1;
              }
              if (__any(bad)) {
                __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            prio_compute();
          }
          if (topc && fmode) {
            if (fmode & F_SELF) {                   // unpaired point of the degenerate row: isign * itself (:861-863)
              un = -un;
              vn = -vn;
            } else {
              const double pu = gran_val(p0), pv = gran_val(p1);
              if (fmode & F_MIRROR) {               // the image of the partner
                un = (fmode & F_NEG) ? -pu : pu;
                vn = (fmode & F_NEG) ? -pv : pv;
              } else {                              // xavg = 0.5*(x1 + isign*x2), x1 = the pair's first member (:792-799)
                const double u1 = (fmode & F_LO) ? un : pu, u2 = (fmode & F_LO) ? pu : un;
                const double v1 = (fmode & F_LO) ? vn : pv, v2 = (fmode & F_LO) ? pv : vn;
                const double xu_ = 0.5 * (u1 + (-u2)), xv_ = 0.5 * (v1 + (-v2));
                un = (fmode & F_NEG) ? -xu_ : xu_;
                vn = (fmode & F_NEG) ? -xv_ : xv_;
              }
            }
          }
        }
      }
      if (k + 1 == r.nsub) break;
      // (D) the edge velocities of subcycle k leave first, then the row for the wavefront above
      const unsigned tag = gran_tag(r.epoch0 + (unsigned)k + 1u);
      if (edge) {
        st_gran(rs, qg, par, un, tag);
        st_gran(rs, qg + 16u, par, vn, tag);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int fd = s_fd[w][c][lx];
          if (fd >= 0) {
            st_gran(rs, (unsigned)fd * 32u, par, un, tag);
            st_gran(rs, (unsigned)fd * 32u + 16u, par, vn, tag);
          }
        }
        if (FOLD) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int rf = rfr[c];
            if (rf >= 0) {                          // a ghost cell the fold fills: this cell's value, negated or not
              const bool neg = (rf >> 30) & 1;
              const unsigned ro = (unsigned)(rf & 0x3fffffff) * 32u;
              st_gran(rs, ro, par, neg ? -un : un, tag);
              st_gran(rs, ro + 16u, par, neg ? -vn : vn, tag);
            }
          }
        }
      }
      s_uv[w][0][lx] = un;
      s_uv[w][1][lx] = vn;
      post_row(&s_uf[w], k + 1);
      TRACE(4)
      PHASE(2)       // 2: momentum, granule stores issued, row posted
      // (E) this wavefront polls the cells its own lanes need
      bool pa = nA, pb = nB;
      if (__any(pa || pb)) {
        prio_wait();
        const long long t0 = wall_clock64();
        int it = 0;
        for (int d = 0; d < r.poll_delay; ++d) __builtin_amdgcn_s_sleep(8);
        u32x4 a0 = {0u, 0u, 0u, 0u}, a1 = a0, b0 = a0, b1 = a0;
        while (true) {
          if (pa) { a0 = ld_gran(rs, offA, par); a1 = ld_gran(rs, offA + 16u, par); }
          if (pb) { b0 = ld_gran(rs, offB, par); b1 = ld_gran(rs, offB + 16u, par); }
          if (pa && gran_ok(a0, a1, tag)) pa = false;     // (a lane that has its granules stops loading: the registers keep them)
          if (pb && gran_ok(b0, b1, tag)) pb = false;
          PHASE_COUNT(4, 1)        // 4: passes of the poll (a count, not cycles)
          if (it == 0) { PHASE(3) }    // 3: the first pass of the poll
          if (!__any(pa || pb)) break;
          asm volatile("" ::: "memory");   // (the loads above are re-issued every pass)
          int bad = __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (!bad && (++it & 7) == 0) bad = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!bad && wall_clock64() - t0 > r.spin_ticks) {
            if (lx == (int)__builtin_ctzll(__ballot(pa || pb)) &&
                __hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              // who gave up first, and on what (read by the host for its message): wait 4 = a granule of subcycle k
              r.abort_flag[1] = 4u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = (unsigned)k;
              r.abort_flag[4] = (unsigned)w; r.abort_flag[5] = (unsigned)lx | (pa ? 256u : 0u) | (pb ? 512u : 0u);
              r.abort_flag[6] = tag;
            }
            bad = 1;
          }
          if (__any(bad)) {
            __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
          }
          __builtin_amdgcn_s_sleep(1);
          for (int d = 0; d < r.poll_sleep; ++d) __builtin_amdgcn_s_sleep(8);
        }
#ifdef CICE4_AMD_STAMPS
        {   // 5: ticks (10 ns) between the store of the granules this lane waited for and now -- the hand-off itself
          const unsigned now = (unsigned)wall_clock64() & 0xffffu;
          const unsigned ta = nA ? ((now - a0.y) & 0xffffu) : 0u, tb = nB ? ((now - b0.y) & 0xffffu) : 0u;
          PHASE_COUNT(5, (long long)(ta > tb ? ta : tb))
        }
#endif
        const double au = gran_val(a0), av = gran_val(a1), bu = gran_val(b0), bv = gran_val(b1);
        if (need_f) { un = au; vn = av; }
        if (need_s) { us = bu; vs = bv; }
        if (lw >= 0) {
          const double xu_ = __shfl(lw_a ? au : bu, lw), xv_ = __shfl(lw_a ? av : bv, lw);
          if (lx == 0) { uwh = xu_; vwh = xv_; }
        }
        if (lq >= 0) {
          const double xu_ = __shfl(lq_a ? au : bu, lq), xv_ = __shfl(lq_a ? av : bv, lq);
          if (lx == 0) { uswh = xu_; vswh = xv_; }
        }
        prio_compute();
      }
      TRACE(5)
      PHASE(6)       // 6: poll of the granules
    }
  } else {
#pragma clang loop unroll(disable)
  for (int k = 0; k < r.nsub; ++k) {
    const bool lastk = r.last && k == r.nsub - 1;
    if (FOLD && toptile && r.prio_top) {
      // the tiles of the top row have a second hand-off per subcycle (the raw velocities across the fold) and everybody
      // waits for them: they issue first on their CU, always
      __builtin_amdgcn_s_setprio(3);
    } else if (r.prio_mode == 2) {
      const int p = (k + gen) % 3;
      if (p == 0) __builtin_amdgcn_s_setprio(3);
      else if (p == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    PHASE_DRAIN      // (diagnostic build: the exchanged velocities have arrived before the clock of the stress starts)
    PHASE(7)         // 7: load of the exchanged velocities
    // (A) velocities of the row below and of the western neighbour
    if (!south_h && w > 0) {
      us = s_uv[w - 1][0][lx];
      vs = s_uv[w - 1][1][lx];
    }
    double uw = up1(un), vw = up1(vn), usw = up1(us), vsw = up1(vs);
    if (lx == 0) {
      uw = uwh;
      vw = vwh;
      usw = uswh;
      vsw = vswh;
    }
    // (B) stress (ice_dyn_evp.F90:1065-1289)
#pragma unroll
    for (int c = 0; c < 8; ++c) o.str[c] = c0;
    if (tact) {
      const double Dxt = s_m[w][0][lx], Dyt = s_m[w][1][lx], Dxhy = s_m[w][2][lx], Dyhx = s_m[w][3][lx],
                   Cxp = s_m[w][5][lx], Cyp = s_m[w][6][lx], Cxm = s_m[w][7][lx], Cym = s_m[w][8][lx],
                   Tiny = s_m[w][9][lx];
      stress_cell<true, DAMP>(a.sc, un, uw, usw, us, vn, vw, vsw, vs, Dxt, Dyt, Dxhy, Dyhx, Cxp, Cyp, Cxm, Cym,
                              lastk ? a.tarear[q] : c0, Tiny, s_m[w][4][lx], s, o, lastk);
    }
    // (C) momentum (:1390-1435): str of the eastern neighbour by wave shift, of the row above through LDS
    const double e1 = down1(o.str[1]), e3 = down1(o.str[3]), e6 = down1(o.str[6]), e7 = down1(o.str[7]);
    if (w > 0) {
      s_edge[w][0][lx] = o.str[2];
      s_edge[w][1][lx] = e3;
      s_edge[w][2][lx] = o.str[5];
      s_edge[w][3][lx] = e7;
    }
    PHASE(0)         // 0: stress
    __syncthreads();
    PHASE(1)         // 1: barrier (C)
    if (uact) {
      const double sx = o.str[0] + e1 + s_edge[w + 1][0][lx] + s_edge[w + 1][1][lx];   // :1415-1416 order
      const double sy = o.str[4] + s_edge[w + 1][2][lx] + e6 + s_edge[w + 1][3][lx];   // :1417-1418 order
      const double uocn = s_x[w][1][lx], vocn = s_x[w][2][lx];
      // waterx, watery are evp_prep2's own expressions of uocn, vocn (:915-916): recomputed, same bits
      double wx, wy;
      water_of(uocn, vocn, s_x[w][6][lx], wx, wy);
      stepu_cell(un, vn, s_x[w][0][lx], uocn, vocn, wx, wy,
                 s_x[w][3][lx], s_x[w][4][lx], s_x[w][5][lx], s_x[w][6][lx], s_x[w][7][lx], sx, sy, ro);
      un = ro.u;
      vn = ro.v;
    }
    if (FOLD && toptile) {
      // (C') the fold changes the owned cells of the top row: the tiles of that row hand each other the raw velocities of
      // this subcycle first (the halo update after stepu, ice_dyn_evp.F90:397-402, on the degenerate row)
      double* xr = r.xraw[k & 1];
      if (topc) {
        st_agent(xr, qb, un);
        st_agent(xr + a.n, qb, vn);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned target2 = r.epoch0 + (unsigned)k + 1u;
      if (threadIdx.x == 0)
        __hip_atomic_store(r.prog2 + (size_t)tile * RES_STRIDE, target2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w == 0) {
        const int dep = lx < 4 ? r.deps2[tile * 4 + lx] : -1;
        bool have = dep < 0;
        int bad = 0;
        const long long t0 = wall_clock64();
        while (true) {
          if (!have)
            have = (int)(__hip_atomic_load(r.prog2 + (size_t)dep * RES_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target2) >= 0;
          if (__all(have)) break;
          const unsigned long long miss = __ballot(!have);
          if (lx == 0) {
            bad = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!bad && wall_clock64() - t0 > r.spin_ticks) {
              if (__hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                r.abort_flag[1] = 3u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = (unsigned)k;   // wait 3 = (C')
                r.abort_flag[4] = (unsigned)miss; r.abort_flag[5] = 0u; r.abort_flag[6] = target2;
              }
              bad = 1;
            }
          }
          bad = __shfl(bad, 0);
          if (bad) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (lx == 0) s_abort = bad;
      }
      __syncthreads();
      if (s_abort) return;
      if (topc && fmode) {
        if (fmode & F_SELF) {                   // unpaired point of the degenerate row: isign * itself (:861-863)
          un = -un;
          vn = -vn;
        } else {
          const double pu = ld_agent(xr, (unsigned)fpad * 8u), pv = ld_agent(xr + a.n, (unsigned)fpad * 8u);
          if (fmode & F_MIRROR) {               // the image of the partner
            un = (fmode & F_NEG) ? -pu : pu;
            vn = (fmode & F_NEG) ? -pv : pv;
          } else {                              // xavg = 0.5*(x1 + isign*x2), x1 = the pair's first member (:792-799)
            const double u1 = (fmode & F_LO) ? un : pu, u2 = (fmode & F_LO) ? pu : un;
            const double v1 = (fmode & F_LO) ? vn : pv, v2 = (fmode & F_LO) ? pv : vn;
            const double xu_ = 0.5 * (u1 + (-u2)), xv_ = 0.5 * (v1 + (-v2));
            un = (fmode & F_NEG) ? -xu_ : xu_;
            vn = (fmode & F_NEG) ? -xv_ : xv_;
          }
        }
      }
    }
    if (!PEER && k + 1 == r.nsub) break;
    s_uv[w][0][lx] = un;     // read after (E); the reads of the previous values lie before (C)
    s_uv[w][1][lx] = vn;
    PHASE(2)         // 2: momentum
    // (D) publish the edge velocities of subcycle k, then the progress word
    double* xu = r.xu[k & 1];
    double* xv = xu + a.n;
    if (PEER && k == 0 && r.pub[tile] != 0) {
      // nothing may be stored into a neighbour's exchange copies before its launch has initialised them
      if (w == 0) {
        const int dep = lx < RES_MAXDEP ? r.deps[tile * RES_MAXDEP + lx] : -1;
        bool have = dep > -2;          // only the tiles of other ranks matter here
        int bad = 0;
        const long long t0 = wall_clock64();
        while (true) {
          if (!have) {
            const unsigned v = __hip_atomic_load(r.rprog + (size_t)(-2 - dep) * RES_STRIDE, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_SYSTEM);
            have = (int)(v - (r.epoch0 + 1u)) >= 0;
          }
          if (__all(have)) break;
          const unsigned long long miss = __ballot(!have);
          if (lx == 0) {
            bad = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!bad && wall_clock64() - t0 > r.spin_ticks) {
              if (__hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                // who gave up first, and on what (read by the host for its message): wait 1 = "the neighbour has begun"
                r.abort_flag[1] = 1u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = 0u;
                r.abort_flag[4] = (unsigned)miss; r.abort_flag[5] = (unsigned)(miss >> 32); r.abort_flag[6] = r.epoch0 + 1u;
              }
              bad = 1;
            }
          }
          bad = __shfl(bad, 0);
          if (bad) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (lx == 0) s_abort = bad;
      }
      __syncthreads();
      if (s_abort) return;
    }
    if (edge) {
      st_agent(xu, qb, un);
      st_agent(xv, qb, vn);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int fd = s_fd[w][c][lx];
        if (fd >= 0) {
          st_agent(xu, (unsigned)fd * 8u, un);
          st_agent(xv, (unsigned)fd * 8u, vn);
        }
      }
      if (PEER) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int rf = s_rfd[w][c][lx];
          if (rf >= 0) {
            const int side = (rf >> 28) & 7;
            const unsigned ro = (unsigned)(rf & 0x0fffffff) * 8u;
            double* pu = r.pxu[side][k & 1];
            st_sys(pu, ro, un);
            st_sys(pu + r.pn[side], ro, vn);
          }
        }
      }
      if (FOLD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int rf = rfr[c];
          if (rf >= 0) {                          // a ghost cell the fold fills: this cell's value, negated or not
            const bool neg = (rf >> 30) & 1;
            const unsigned ro = (unsigned)(rf & 0x3fffffff) * 8u;
            st_agent(xu, ro, neg ? -un : un);
            st_agent(xv, ro, neg ? -vn : vn);
          }
        }
      }
    }
    PHASE(3)         // 3: edge stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE(4)         // 4: stores drained
    __syncthreads();
    PHASE(5)         // 5: barrier (D)
    const unsigned target = r.epoch0 + (unsigned)k + (PEER ? 2u : 1u);
    if (threadIdx.x == 0)
      __hip_atomic_store(r.prog + (size_t)tile * RES_STRIDE, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PEER && threadIdx.x < RES_NPEER && s_pub[threadIdx.x])
      __hip_atomic_store(r.prp[threadIdx.x] + (size_t)tile * RES_STRIDE, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (E) wait until every producer of this tile's halo has published subcycle k
    if (w == 0) {
      const int dep = lx < RES_MAXDEP ? r.deps[tile * RES_MAXDEP + lx] : -1;
      bool have = dep == -1;
      int bad = 0;
      const long long t0 = wall_clock64();
      while (true) {
        if (!have) {
          unsigned v;
          if (PEER && dep <= -2)
            v = __hip_atomic_load(r.rprog + (size_t)(-2 - dep) * RES_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else
            v = __hip_atomic_load(r.prog + (size_t)dep * RES_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          have = (int)(v - target) >= 0;
        }
        if (__all(have)) break;
        const unsigned long long miss = __ballot(!have);
        if (lx == 0) {
          bad = (int)__hip_atomic_load(r.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!bad && wall_clock64() - t0 > r.spin_ticks) {
            if (__hip_atomic_exchange(r.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              r.abort_flag[1] = 2u; r.abort_flag[2] = (unsigned)tile; r.abort_flag[3] = (unsigned)k;   // wait 2 = (E)
              r.abort_flag[4] = (unsigned)miss; r.abort_flag[5] = (unsigned)(miss >> 32); r.abort_flag[6] = target;
            }
            bad = 1;
          }
        }
        bad = __shfl(bad, 0);
        if (bad) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (lx == 0) s_abort = bad;
    }
    __syncthreads();
    PHASE(6)         // 6: progress word, poll of the producers' words, barrier (E)
    if (s_abort) return;
    // every exchanged velocity is an agent-scope load (system scope where another rank may have written it)
    if (foreign) {
      un = PEER ? ld_sys(xu, qb) : ld_agent(xu, qb);
      vn = PEER ? ld_sys(xv, qb) : ld_agent(xv, qb);
    }
    if (south_h) {
      us = PEER ? ld_sys(xu, qb - nxb) : ld_agent(xu, qb - nxb);
      vs = PEER ? ld_sys(xv, qb - nxb) : ld_agent(xv, qb - nxb);
    }
    if (west_h) {
      uwh = PEER ? ld_sys(xu, qb - 8u) : ld_agent(xu, qb - 8u);
      vwh = PEER ? ld_sys(xv, qb - 8u) : ld_agent(xv, qb - 8u);
      uswh = PEER ? ld_sys(xu, qb - nxb - 8u) : ld_agent(xu, qb - nxb - 8u);
      vswh = PEER ? ld_sys(xv, qb - nxb - 8u) : ld_agent(xv, qb - nxb - 8u);
    }
  }
  }   // (!GRAN)

  stamp_at(r.stamps, 1);
  PHASE_STORE(r.phases)
  // ---- results (owners only), into the other copy of the state ----
  if (sown) {
#pragma unroll
    for (int c = 0; c < 12; ++c) a.sig_out[(size_t)c * a.n + q] = s[c];
    if (r.last) {
      a.divu[q] = o.divu;
      a.rdg_conv[q] = o.rdg_conv;
      a.rdg_shear[q] = o.rdg_shear;
      a.shear[q] = o.shear;
      a.prs_sig[q] = o.prs_sig;
    }
  }
  if (PEER) {   // the final velocities of ghost cells another rank owns (after the last exchange the halo registers hold them)
    if (rghost) {
      a.u_out[q] = un;
      a.v_out[q] = vn;
    }
    if (south_h && r.rslot[q - nx] == -2) {       // the ghost row below the block is nobody's own position
      a.u_out[q - nx] = us;
      a.v_out[q - nx] = vs;
    }
    if (west_h) {                                 // nor is the ghost column to the west (its corners)
      if (r.rslot[q - 1] == -2) {
        a.u_out[q - 1] = uwh;
        a.v_out[q - 1] = vwh;
      }
      if (r.rslot[q - nx - 1] == -2) {
        a.u_out[q - nx - 1] = uswh;
        a.v_out[q - nx - 1] = vswh;
      }
    }
  }
  if (FOLD && (uact || topc)) {   // the ghost cells the fold fills, and top-row cells without ice (changed by the fold all the same)
    a.u_out[q] = un;
    a.v_out[q] = vn;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int rf = rfr[c];
      if (rf >= 0) {
        const bool neg = (rf >> 30) & 1;
        a.u_out[rf & 0x3fffffff] = neg ? -un : un;
        a.v_out[rf & 0x3fffffff] = neg ? -vn : vn;
      }
    }
  }
  if (uact) {
    a.u_out[q] = un;
    a.v_out[q] = vn;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int fd = s_fd[w][c][lx];
      if (fd >= 0) {
        a.u_out[fd] = un;
        a.v_out[fd] = vn;
      }
    }
    if (r.last) {
      a.strintx[q] = ro.strintx;
      a.strinty[q] = ro.strinty;
      a.strocnx[q] = ro.taux;
      a.strocny[q] = ro.tauy;
    }
  }
}

struct StressListArgs {
  EvpScalars sc;
  int nx, ny, ksub, icellt;
  const int32_t *ti, *tj;
  const double *uvel, *vvel, *g[10], *strength;  // g: dxt,dyt,dxhy,dyhx,cxp,cyp,cxm,cym,tarear,tinyarea
  double *sig[12], *diag[5], *str;               // diag: shear,divu,prs_sig,rdg_conv,rdg_shear
};

template <bool LAST, bool DAMP>
__global__ __launch_bounds__(256) void k_stress_list(const StressListArgs a) {
  const int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= a.icellt) return;
  const int i = a.ti[ij], j = a.tj[ij], nx = a.nx;
  const size_t np = (size_t)nx * a.ny;
  const size_t q = (size_t)(j - 1) * nx + (i - 1), qw = q - 1, qs = q - nx, qsw = q - nx - 1;
  double s[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) s[c] = a.sig[c][q];
  StressOut o;
  stress_cell<LAST, DAMP>(a.sc, a.uvel[q], a.uvel[qw], a.uvel[qsw], a.uvel[qs], a.vvel[q], a.vvel[qw],
                          a.vvel[qsw], a.vvel[qs], a.g[0][q], a.g[1][q], a.g[2][q], a.g[3][q],
                          a.g[4][q], a.g[5][q], a.g[6][q], a.g[7][q], a.g[8][q], a.g[9][q],
                          a.strength[q], s, o);
#pragma unroll
  for (int c = 0; c < 12; ++c) a.sig[c][q] = s[c];
#pragma unroll
  for (int c = 0; c < 8; ++c) a.str[(size_t)c * np + q] = o.str[c];
  a.diag[2][q] = o.prs_sig;
  if (LAST) {
    a.diag[0][q] = o.shear;
    a.diag[1][q] = o.divu;
    a.diag[3][q] = o.rdg_conv;
    a.diag[4][q] = o.rdg_shear;
  }
}

struct StepuListArgs {
  int nx, ny, icellu;
  const int32_t *ui, *uj;
  const double *in[10], *str;  // in: aiu,uocn,vocn,waterx,watery,forcex,forcey,umassdtei,fm,uarear
  double* io[6];               // strocnx,strocny,strintx,strinty,uvel,vvel
};

__global__ __launch_bounds__(256) void k_stepu_list(const StepuListArgs a) {
  const int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= a.icellu) return;
  const int i = a.ui[ij], j = a.uj[ij], nx = a.nx;
  const size_t np = (size_t)nx * a.ny;
  const size_t q = (size_t)(j - 1) * nx + (i - 1), qe = q + 1, qn = q + nx, qne = q + nx + 1;
  const double sx = a.str[0 * np + q] + a.str[1 * np + qe] + a.str[2 * np + qn] + a.str[3 * np + qne];
  const double sy = a.str[4 * np + q] + a.str[5 * np + qn] + a.str[6 * np + qe] + a.str[7 * np + qne];
  StepuOut r;
  stepu_cell(a.io[4][q], a.io[5][q], a.in[0][q], a.in[1][q], a.in[2][q], a.in[3][q], a.in[4][q],
             a.in[5][q], a.in[6][q], a.in[7][q], a.in[8][q], a.in[9][q], sx, sy, r);
  a.io[4][q] = r.u;
  a.io[5][q] = r.v;
  a.io[2][q] = r.strintx;
  a.io[3][q] = r.strinty;
  a.io[0][q] = r.taux;
  a.io[1][q] = r.tauy;
}

// ---- once-per-step kernels ------------------------------------------------------------------
struct PrepArgs {
  EvpScalars sc;
  int nx, ny, nblocks;
  size_t n;
  const int32_t *blk, *tmask, *umask;
  const double *aice, *vice, *vsno, *aice0, *aicen, *vicen, *strairxT, *strairyT, *uocn, *vocn;
  const double *ss_tltx, *ss_tlty;
  const double *tarea, *uarea, *fcor;
  double *strairx, *strairy, *tmass, *umass, *aiu, *work1;
  int32_t *icetmask, *iceumask;
  double *rdg_conv, *rdg_shear, *divu, *shear, *prs_sig;
  double *umassdtei, *waterx, *watery, *forcex, *forcey, *fm, *strtltx, *strtlty, *strocnx, *strocny,
      *strintx, *strinty, *strength, *strocnxT, *strocnyT;
  double *u, *v, *sig;
  int kstrength, krdg_partic, krdg_redist;
  double mu_rdg;
  unsigned long long* counters;
};

__device__ __forceinline__ bool cell_of(const PrepArgs& a, size_t t, int& b, int& i, int& j) {
  if (t >= a.n) return false;
  const size_t np = (size_t)a.nx * a.ny;
  b = (int)(t / np);
  const size_t r = t - (size_t)b * np;
  j = (int)(r / a.nx) + 1;
  i = (int)(r - (size_t)(j - 1) * a.nx) + 1;
  return true;
}

__device__ __forceinline__ bool tmphm_at(const PrepArgs& a, size_t q) {  // :651-661
  if (!a.tmask[q]) return false;
  const double tm = rhoi * a.vice[q] + rhos * a.vsno[q];
  return (a.aice[q] > a_min) && (tm > m_min);
}

// evp :214-224 + evp_prep1 :586-694
__global__ __launch_bounds__(256) void k_prep1(const PrepArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  a.rdg_conv[t] = c0; a.rdg_shear[t] = c0; a.divu[t] = c0; a.shear[t] = c0; a.prs_sig[t] = c0;
  a.tmass[t] = a.tmask[t] ? (rhoi * a.vice[t] + rhos * a.vsno[t]) : c0;
  a.strairx[t] = a.strairxT[t];
  a.strairy[t] = a.strairyT[t];
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  int m = 0;
  if (i >= ilo && i <= ihi && j >= jlo && j <= jhi && a.tmask[t]) {
    bool any = false;
    for (int dj = -1; dj <= 1; ++dj)
      for (int di = -1; di <= 1; ++di) any = any || tmphm_at(a, t + (ptrdiff_t)dj * a.nx + di);
    m = any ? 1 : 0;
  }
  a.icetmask[t] = m;
}

// to_ugrid (ice_grid.F90:1580-1633) for two fields at once: work2 = 0 off the physical domain
__global__ __launch_bounds__(256) void k_to_ugrid2(const PrepArgs a, const double* __restrict__ f1,
                                                   const double* __restrict__ f2,
                                                   double* __restrict__ o1, double* __restrict__ o2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  double r1 = c0, r2 = c0;
  if (i >= ilo && i <= ihi && j >= jlo && j <= jhi) {
    const size_t e = t + 1, n = t + a.nx, ne = t + a.nx + 1;
    const double ta = a.tarea[t], tb = a.tarea[e], tc = a.tarea[n], td = a.tarea[ne], ua = a.uarea[t];
    r1 = p25 * (f1[t] * ta + f1[e] * tb + f1[n] * tc + f1[ne] * td) / ua;
    r2 = p25 * (f2[t] * ta + f2[e] * tb + f2[n] * tc + f2[ne] * td) / ua;
  }
  o1[t] = r1;
  o2[t] = r2;
}

// to_tgrid (ice_grid.F90:1684-1732) for two fields; ghost cells of the outputs are untouched
__global__ __launch_bounds__(256) void k_to_tgrid2(const PrepArgs a, const double* __restrict__ f1,
                                                   const double* __restrict__ f2,
                                                   double* __restrict__ o1, double* __restrict__ o2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  if (i >= ilo && i <= ihi && j >= jlo && j <= jhi) {
    const size_t w = t - 1, s = t - a.nx, sw = t - a.nx - 1;
    const double ua = a.uarea[t], ub = a.uarea[w], uc = a.uarea[s], ud = a.uarea[sw], ta = a.tarea[t];
    o1[t] = p25 * (f1[t] * ua + f1[w] * ub + f1[s] * uc + f1[sw] * ud) / ta;
    o2[t] = p25 * (f2[t] * ua + f2[w] * ub + f2[s] * uc + f2[sw] * ud) / ta;
  }
}

// evp_prep2 :703-938 (non-coupled, non-AusCOM branches), dense form: the compressed lists
// become the masks icetmask (T-cells, incl. N/E ghost ring) and iceumask (U-cells)
__global__ __launch_bounds__(256) void k_prep2(const PrepArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  double wx = c0, wy = c0, fx = c0, fy = c0, umd = c0;
  if (a.icetmask[t] == 0) {
#pragma unroll
    for (int c = 0; c < 12; ++c) a.sig[(size_t)c * a.n + t] = c0;
  }
  if (i >= ilo && i <= ihi && j >= jlo && j <= jhi) {
    const bool old = a.iceumask[t] != 0;
    const bool now = a.umask[t] && (a.aiu[t] > a_min) && (a.umass[t] > m_min);
    a.iceumask[t] = now ? 1 : 0;
    if (now) {
      if (!old) {
        a.u[t] = a.uocn[t];
        a.v[t] = a.vocn[t];
      }
      umd = a.umass[t] * a.sc.dtei;
      const double fmv = a.fcor[t] * a.umass[t];
      a.fm[t] = fmv;
      water_of(a.uocn[t], a.vocn[t], fmv, wx, wy);
#ifdef CICE4_AMD_AUSCOM   // :919-933: the tilt from the ocean model's surface slope when use_ocnslope is set
      const double tx = ocnslope ? -gravit * a.umass[t] * a.ss_tltx[t] : -fmv * a.vocn[t];
      const double ty = ocnslope ? -gravit * a.umass[t] * a.ss_tlty[t] : fmv * a.uocn[t];
#else
      const double tx = -fmv * a.vocn[t], ty = fmv * a.uocn[t];
#endif
      a.strtltx[t] = tx;
      a.strtlty[t] = ty;
      fx = a.strairx[t] + tx;
      fy = a.strairy[t] + ty;
    } else {
      a.u[t] = c0; a.v[t] = c0;
      a.strintx[t] = c0; a.strinty[t] = c0;
      a.strocnx[t] = c0; a.strocny[t] = c0;
    }
  }
  a.waterx[t] = wx; a.watery[t] = wy; a.forcex[t] = fx; a.forcey[t] = fy; a.umassdtei[t] = umd;
}

// ice_strength (ice_mechred.F90:1869-2036; asum_ridging :573, ridge_itd :773-1098) on the
// T-cell list of evp_prep2 (icetmask = 1 on ilo..ihi+1, jlo..jhi+1); zero elsewhere.
__global__ __launch_bounds__(256) void k_strength(const PrepArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  double st = c0;
  const bool phys = (i >= ilo && i <= ihi && j >= jlo && j <= jhi);
  if (a.kstrength != 1) {
    if (phys) st = 2.75e4 * a.vice[t] * exp_libm(-20.0 * (c1 - a.aice[t]));
  } else if (i >= ilo && i <= ihi + 1 && j >= jlo && j <= jhi + 1 && a.icetmask[t] == 1) {
    constexpr double Cf = 17.0, Cp = p5 * gravit * (rhow - rhoi) * rhoi / rhow;
    constexpr double Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0;
    constexpr double Gstari = c1 / Gstar, astari = c1 / astar;
    const size_t np = (size_t)a.nx * a.ny;
    const size_t qc = (size_t)b * NCAT * np + (t - (size_t)b * np);  // (nx,ny,ncat,nb) layout
    double Gsum[NCAT + 2], apartic[NCAT + 1], hrmin[NCAT + 1], hrmax[NCAT + 1], hrexp[NCAT + 1],
        krdg[NCAT + 1], an[NCAT + 1], vn[NCAT + 1];
    Gsum[0] = c0;
    Gsum[1] = (a.aice0[t] > puny) ? a.aice0[t] : Gsum[0];
    apartic[0] = c0;
#pragma unroll
    for (int n = 1; n <= NCAT; ++n) {
      an[n] = a.aicen[qc + (size_t)(n - 1) * np];
      vn[n] = a.vicen[qc + (size_t)(n - 1) * np];
      Gsum[n + 1] = (an[n] > puny) ? Gsum[n] + an[n] : Gsum[n];
      apartic[n] = c0; hrmin[n] = c0; hrmax[n] = c0; hrexp[n] = c0; krdg[n] = c1;
    }
    const double work = c1 / Gsum[NCAT + 1];
#pragma unroll
    for (int n = 0; n <= NCAT; ++n) Gsum[n + 1] = Gsum[n + 1] * work;
    if (a.krdg_partic == 0) {
#pragma unroll
      for (int n = 0; n <= NCAT; ++n) {
        const double g = Gsum[n + 1], gm = Gsum[n];
        if (g < Gstar)
          apartic[n] = Gstari * (g - gm) * (c2 - (gm + g) * Gstari);
        else if (gm < Gstar)
          apartic[n] = Gstari * (Gstar - gm) * (c2 - (gm + Gstar) * Gstari);
      }
    } else {
      const double xtmp = c1 / (c1 - exp_libm(-astari));
#pragma unroll
      for (int n = -1; n <= NCAT; ++n) Gsum[n + 1] = exp_libm(-Gsum[n + 1] * astari) * xtmp;
#pragma unroll
      for (int n = 0; n <= NCAT; ++n) apartic[n] = Gsum[n] - Gsum[n + 1];
    }
#pragma unroll
    for (int n = 1; n <= NCAT; ++n) {
      if (an[n] > puny) {
        double hi = vn[n] / an[n];
        if (a.krdg_redist == 0) {
          hrmin[n] = fmin(c2 * hi, hi + maxraft);
          hrmax[n] = c2 * sqrt(Hstar * hi);
          hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
          const double hrmean = p5 * (hrmin[n] + hrmax[n]);
          krdg[n] = hrmean / hi;
        } else {
          hi = fmax(hi, puny);
          hrmin[n] = fmin(c2 * hi, hi + maxraft);
          hrexp[n] = a.mu_rdg * sqrt(hi);
          krdg[n] = (hrmin[n] + hrexp[n]) / hi;
        }
      }
    }
    double aksum = apartic[0];
#pragma unroll
    for (int n = 1; n <= NCAT; ++n) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
    double s = c0;
#pragma unroll
    for (int n = 1; n <= NCAT; ++n) {
      if (an[n] > puny && apartic[n] > c0) {
        const double hi = vn[n] / an[n];
        double h2rdg;
        if (a.krdg_redist == 0)
          h2rdg = p333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) /
                  (hrmax[n] - hrmin[n]);
        else
          h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
        const double dh2rdg = -hi * hi + h2rdg / krdg[n];
        s = s + apartic[n] * dh2rdg;
      }
    }
    st = Cf * Cp * s / aksum;
  }
  a.strength[t] = st;
}

// evp_finish :1452-1549 (strocnxT/yT zeroed everywhere first)
__global__ __launch_bounds__(256) void k_finish(const PrepArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  if (!cell_of(a, t, b, i, j)) return;
  double xT = c0, yT = c0;
  const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jlo = a.blk[6 * b + 2], jhi = a.blk[6 * b + 3];
  if (i >= ilo && i <= ihi && j >= jlo && j <= jhi && a.iceumask[t]) {  // the indxu list
    const double du = a.uocn[t] - a.u[t], dv = a.vocn[t] - a.v[t];
    const double vrel = dragw * sqrt(du * du + dv * dv);
#ifdef CICE4_AMD_AUSCOM   // :1524-1536: rotate to the opposite direction in the Southern Hemisphere
    const bool south = a.fm[t] < 0.0;
    const double sx = a.strocnx[t] - vrel * (south ? a.u[t] * cosw + a.v[t] * sinw : a.u[t] * cosw - a.v[t] * sinw) * a.aiu[t];
    const double sy = a.strocny[t] - vrel * (south ? a.v[t] * cosw - a.u[t] * sinw : a.v[t] * cosw + a.u[t] * sinw) * a.aiu[t];
#else
    const double sx = a.strocnx[t] - vrel * (a.u[t] * cosw - a.v[t] * sinw) * a.aiu[t];
    const double sy = a.strocny[t] - vrel * (a.v[t] * cosw + a.u[t] * sinw) * a.aiu[t];
#endif
    a.strocnx[t] = sx;
    a.strocny[t] = sy;
    xT = sx / a.aiu[t];
    yT = sy / a.aiu[t];
  }
  a.strocnxT[t] = xT;
  a.strocnyT[t] = yT;
}

// The read-only inputs of the sweep kernel, interleaved cell by cell (SkewArgs::uar4 / hnhe / msk): once per evp(dt)
// (the momentum inputs and the masks) and once per grid (HTN | HTE).  16 bytes per lane and load instead of 8.
__global__ __launch_bounds__(256) void k_skew_pack(size_t n, const double* __restrict__ uar, const int32_t* __restrict__ tmk,
                                                   const int32_t* __restrict__ umk, double* __restrict__ uar4,
                                                   int32_t* __restrict__ msk) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uar4[2 * ((size_t)p * n + q)] = uar[(size_t)(2 * p) * n + q];
    uar4[2 * ((size_t)p * n + q) + 1] = uar[(size_t)(2 * p + 1) * n + q];
  }
  msk[q] = (tmk[q] == 1 ? 1 : 0) | (umk[q] != 0 ? 2 : 0);
}
// Which rows of a column strip of the sweep kernel hold anything to compute (SkewArgs::rowact): one wavefront per (block,
// strip, row), its lanes laid over the columns exactly as the sweep kernel lays them; a row counts if one of the strip's
// OWN columns has ice on the T-cell of that row or of the row above (whose stress the row's momentum equation reads), or
// an active U-cell.  Cells without ice are never written by any subcycle kernel (both copies of the state start out
// identical: Evp::prepare), so rows outside a workgroup's first .. last such row need no workgroup at all.
__global__ __launch_bounds__(256) void k_skew_rowact(int K, int strips, int own_shift, int nx, int ny, int ew_cyclic,
                                                     const int32_t* __restrict__ blk, const int32_t* __restrict__ msk,
                                                     unsigned char* __restrict__ act) {
  const int lx = threadIdx.x & 63, rows = ny - 2;
  const int rr = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);     // row relative to jlo
  const int txi = blockIdx.y, b = blockIdx.z;
  if (rr >= rows) return;
  const int ilo = blk[6 * b + 0], ihi = blk[6 * b + 1], jlo = blk[6 * b + 2], jhi = blk[6 * b + 3];
  const int r = jlo + rr;
  bool f = false;
  if (r <= jhi) {
    const int OWNW = 64 - 2 * K, ncol = ihi - ilo + 1;
    const int own0 = txi == 0 ? K + 1 + own_shift : K;
    const int kraw = (txi == 0 ? 0 : (OWNW - 1 - own_shift) + (txi - 1) * OWNW) + lx - own0;
    const bool own_col = lx >= own0 && lx <= 63 - K && kraw >= 0 && kraw <= ncol;
    if (own_col) {
      const int col = ilo + kraw;                                   // (<= ihi + 1: the ghost column's T-cell is computed too)
      const size_t q = (size_t)b * nx * ny + (size_t)(r - 1) * nx + (size_t)(col - 1);
      const int m0 = msk[q], m1 = msk[q + nx];
      f = m0 != 0 || (m1 & 1) != 0;
    }
    (void)ew_cyclic;
  }
  const unsigned long long any = __ballot(f);
  if (lx == 0) act[((size_t)b * strips + txi) * rows + rr] = any ? 1 : 0;
}
// ... and the RUNS of such rows (gaps of up to `gap` rows bridged), so that a workgroup finds its next run with two loads:
// nxt[row] = the first row >= row that holds anything (or -1), rend[row] = the last row of the run an active row belongs to.
// One thread per (block, strip), top row down.
__global__ __launch_bounds__(64) void k_skew_runs(int nstrips, int rows, int gap, const unsigned char* __restrict__ act,
                                                  int32_t* __restrict__ nxt, int32_t* __restrict__ rend) {
  const int sidx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (sidx >= nstrips) return;
  const unsigned char* a = act + (size_t)sidx * rows;
  int32_t* nx_ = nxt + (size_t)sidx * rows;
  int32_t* re_ = rend + (size_t)sidx * rows;
  int above = -1, above_end = -1;   // the nearest active row above the current one, and where its run ends
  for (int r = rows - 1; r >= 0; --r) {
    if (a[r]) {
      const int e = (above >= 0 && above - r - 1 <= gap) ? above_end : r;
      re_[r] = e;
      above = r;
      above_end = e;
    } else {
      re_[r] = -1;
    }
    nx_[r] = above;
  }
}
// plane layout (14 planes of n doubles: u, v, 12 stresses) <-> the sweep's pair layout (7 planes of n pairs)
__global__ __launch_bounds__(256) void k_to_pairs(size_t n, const double* __restrict__ in, double* __restrict__ out,
                                                  double* __restrict__ out2) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
#pragma unroll
  for (int p = 0; p < 7; ++p) {
    dbl2 t;
    t.x = in[(size_t)(2 * p) * n + q];
    t.y = in[(size_t)(2 * p + 1) * n + q];
    *(dbl2*)(out + 2 * ((size_t)p * n + q)) = t;
    if (out2) *(dbl2*)(out2 + 2 * ((size_t)p * n + q)) = t;
  }
}
__global__ __launch_bounds__(256) void k_from_pairs(size_t n, const double* __restrict__ in, double* __restrict__ out) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
#pragma unroll
  for (int p = 0; p < 7; ++p) {
    const dbl2 t = *(const dbl2*)(in + 2 * ((size_t)p * n + q));
    out[(size_t)(2 * p) * n + q] = t.x;
    out[(size_t)(2 * p + 1) * n + q] = t.y;
  }
}

__global__ __launch_bounds__(256) void k_skew_pack_grid(size_t n, const double* __restrict__ HTN, const double* __restrict__ HTE,
                                                        double* __restrict__ hnhe) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  hnhe[2 * q] = HTN[q];
  hnhe[2 * q + 1] = HTE[q];
}

__global__ __launch_bounds__(1024) void k_count_active(const PrepArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int b, i, j;
  unsigned long long nt = 0, nu = 0;
  if (cell_of(a, t, b, i, j)) {
    const int ilo = a.blk[6 * b], ihi = a.blk[6 * b + 1], jhi = a.blk[6 * b + 3];
    // rows this block owns (a wide-halo domain recomputes its overlap rows; they are not counted)
    const int ojlo = a.blk[6 * b + 4], ojhi = a.blk[6 * b + 5];
    const int tjhi = ojhi + (ojhi == jhi ? 1 : 0);
    if (i >= ilo && i <= ihi + 1 && j >= ojlo && j <= tjhi && a.icetmask[t] == 1) nt = 1;
    if (i >= ilo && i <= ihi && j >= ojlo && j <= ojhi && a.iceumask[t]) nu = 1;
  }
  for (int off = 32; off > 0; off >>= 1) {
    nt += __shfl_down(nt, off);
    nu += __shfl_down(nu, off);
  }
  // one pair of atomics per workgroup of 1,024, not per wavefront: atomics on one address are served one at a time
  // (13.6 ns each, therm.h), which made this kernel 3.2 ms at 0.1 degree
  __shared__ unsigned long long s_n[2][16];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_n[0][w] = nt; s_n[1][w] = nu; }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned long long sum = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) sum += s_n[threadIdx.x][k];
    if (sum) atomicAdd(&a.counters[threadIdx.x], sum);
  }
}

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// ---------------------------------------------------------------------------------------------
Evp::~Evp() {
  drop_graph();
  if (res_done_ev) (void)hipEventDestroy(res_done_ev);
  for (hipEvent_t e : {res_t0, res_t1, sub_t0, sub_t1})
    if (e) (void)hipEventDestroy(e);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_join) (void)hipEventDestroy(ev_join);
  if (stream2) (void)hipStreamDestroy(stream2);
  if (res_why) (void)hipHostFree(res_why);
}

void Evp::drop_graph() {
  if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
  graph_exec = nullptr;
  graph_key[0] = -1;
}

void Evp::set_option(const char* key, int value) {
  if (!std::strcmp(key, "waves")) {
    CICE_REQUIRE(value == 4 || value == 8 || value == 16, "waves must be 4, 8 or 16");
    waves = value;
  } else if (!std::strcmp(key, "rows_per_wave")) {
    CICE_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8, "rows_per_wave must be 1, 2, 4 or 8");
    rows_per_wave = value;
  } else if (!std::strcmp(key, "derive_metrics")) {
    derive_on = value != 0;
  } else if (!std::strcmp(key, "use_graph")) {
    use_graph = value != 0;
  } else if (!std::strcmp(key, "comm_graph")) {    // capture RCCL calls of a multi-rank loop too (default off)
    comm_graph = value != 0;
  } else if (!std::strcmp(key, "fuse")) {          // two subcycles per launch where the domain allows
    fuse_on = value != 0;
  } else if (!std::strcmp(key, "resident")) {      // whole loop in one launch where the grid fits (default on)
    resident_on = value != 0;
    if (value == 2) {                              // 2: also forget an earlier time-out
      resident_failed = false;
      res_level = 0;
      res_retry_in = 0;
    }
  } else if (!std::strcmp(key, "skew_fold")) {              // sweeps on a tripole grid (a band of top rows beside them)
    skew_fold_on = value != 0;
  } else if (!std::strcmp(key, "resident_blocks")) {        // the one-launch loop on a one-rank domain of several blocks
    res_blocks_on = value != 0;
  } else if (!std::strcmp(key, "resident_fold")) {          // a tripole north boundary inside the one-launch loop
    res_fold_on = value != 0;
  } else if (!std::strcmp(key, "resident_retry_steps")) {   // evp(dt) calls until a time-out is forgiven, 0 = never
    CICE_REQUIRE(value >= 0, "resident_retry_steps must be >= 0");
    res_retry_steps = value;
  } else if (!std::strcmp(key, "resident_peer_share")) {   // contexts that share this device in the cross-rank loop (tests: 2)
    CICE_REQUIRE(value >= 1 && value <= 8, "resident_peer_share must be 1 .. 8");
    res_peer_share = value;
  } else if (!std::strcmp(key, "resident_peer_agree")) {   // all-reduce the time-out flag over the ranks (needs cice_comm_init)
    res_peer_agree = value != 0;
  } else if (!std::strcmp(key, "resident_spin_us")) {   // how long a tile waits for its neighbours before giving up
    CICE_REQUIRE(value >= 0, "resident_spin_us must be >= 0");
    res_spin_us = value;
  } else if (!std::strcmp(key, "resident_prio")) {    // issue priority among the workgroups of a CU (dense shape): 0, 1, 2
    CICE_REQUIRE(value >= 0 && value <= 6, "resident_prio must be 0 .. 6");
    res_prio = value;
  } else if (!std::strcmp(key, "resident_dense")) {   // three 4-wavefront workgroups per CU where that fills the chip
    res_dense = value != 0;
  } else if (!std::strcmp(key, "resident_granules")) {   // edge velocities as data-tagged granules (one rank, no fold); 0: progress words,
    CICE_REQUIRE(value >= 0 && value <= 2, "resident_granules must be 0, 1 or 2");   // 1: by the ice cover of the last step, 2: always
    res_gran = value;
  } else if (!std::strcmp(key, "keep_state")) {       // cice_evp: see Evp::run
    CICE_REQUIRE(value >= 0 && value <= 2, "keep_state must be 0, 1 or 2");
    keep_state = value;
    io_valid = false;
  } else if (!std::strcmp(key, "lazy_stresses")) {
    lazy_sig = value != 0;
  } else if (!std::strcmp(key, "resident_waves")) {   // 0 = auto
    CICE_REQUIRE(value == 0 || value == 4 || value == 6 || value == 8 || value == 11 || value == 12,
                 "resident_waves must be 0, 4, 6, 8, 11 or 12");
    res_w_opt = value;
  } else if (!std::strcmp(key, "skew")) {          // K subcycles per sweep (k_subcycle_skew) where the domain allows
    skew_on = value != 0;
  } else if (!std::strcmp(key, "skew_levels")) {   // 0 = auto
#ifdef CICE4_AMD_EXPERIMENTS
    CICE_REQUIRE(value == 0 || value == 2 || value == 3 || value == 4 || value == 5 || value == 6 || value == 8,
                 "skew_levels must be 0, 2, 3, 4, 5, 6 or 8");
#else
    CICE_REQUIRE(value == 0 || value == 2 || value == 3 || value == 4,
                 "skew_levels must be 0, 2, 3 or 4 (5, 6, 8: measured slower, in -DCICE4_AMD_EXPERIMENTS builds only)");
#endif
    skew_k_opt = value;
  } else if (!std::strcmp(key, "skew_stagger_ns")) {   // start delay between the workgroups that share a CU
    CICE_REQUIRE(value >= 0 && value <= 100000, "skew_stagger_ns must be 0 .. 100000");
    skew_stagger_ns = value;
  } else if (!std::strcmp(key, "skew_gen_pct")) {   // segments of the workgroup dispatched first on a CU this much longer
    CICE_REQUIRE(value >= 0 && value <= 60, "skew_gen_pct must be 0 .. 60");
    skew_gen_pct = value;
  } else if (!std::strcmp(key, "resident_map")) {   // tile map of the one-launch loop: -1 chosen by the ice cover, 0 / 1 fixed (k_res_choose_map)
    CICE_REQUIRE(value >= -1 && value <= 1, "resident_map must be -1, 0 or 1");
    res_map_opt = value;
    res_map_stale = true;
  } else if (!std::strcmp(key, "skew_rowact")) {    // workgroups of the sweep shrink to the rows that hold ice (k_skew_rowact)
    rowact_opt = value != 0;
    skew_packed = false;
  } else if (!std::strcmp(key, "skew_balance")) {   // segments of the sweep follow the measured cost of their rows (balance_after_sweep)
    skew_balance = value != 0;
    skew_rows_key[0] = -1;                          // a new table: the static one, measured again if this is on
  } else if (!std::strcmp(key, "skew_balance_every")) {   // loops between two tuning phases
    CICE_REQUIRE(value >= 1, "skew_balance_every must be >= 1");
    bal_every = value;
  } else if (!std::strcmp(key, "skew_fill")) {      // longer segments for workgroups on CUs that hold fewer of them (build_skew_rows)
    CICE_REQUIRE(value >= 0 && value <= 100, "skew_fill must be 0 .. 100 (per cent)");
    skew_fill = value;
  } else if (!std::strcmp(key, "skew_debug")) {  // record start / end ticks of every workgroup of the sweep kernel
    skew_debug = value != 0;
  } else if (!std::strcmp(key, "stamps")) {      // diagnostic build (-DCICE4_AMD_STAMPS): cycle / wall-clock stamps per workgroup
    stamps_on = value != 0;
  } else if (!std::strcmp(key, "skew_split")) {  // the sweep in front of a wide-halo refresh as edge + interior launches
    split_on = value != 0;
  } else if (!std::strcmp(key, "skew_subs")) {    // wavefronts per level of the sweep kernel: 3 (default shape) or 1
#ifdef CICE4_AMD_EXPERIMENTS
    CICE_REQUIRE(value == 1 || value == 3, "skew_subs must be 1 or 3");
#else
    CICE_REQUIRE(value == 1, "skew_subs must be 1 (3: measured slower, in -DCICE4_AMD_EXPERIMENTS builds only)");
#endif
    skew_subs_opt = value;
    std::memset(strips_cache, 0, sizeof(strips_cache));
    tile_tabs.clear();
  } else if (!std::strcmp(key, "skew_pairs")) {   // the sweep's pair layout of u | v and the stresses (16-byte loads and stores)
    pairs_on = value != 0;
  } else if (!std::strcmp(key, "skew_split_probe")) {   // measurement aid, see Evp::tiles_for (wrong results by design)
    CICE_REQUIRE(value >= 0 && value <= 64, "skew_split_probe must be 0 .. 64");
    split_probe = value;
    tile_tabs.clear();
  } else if (!std::strcmp(key, "skew_trim_ext")) {  // sweeps on wide-halo slabs compute only the extension rows still needed
    trim_ext_on = value != 0;
  } else if (!std::strcmp(key, "skew_prio")) {   // rotate issue priorities among the workgroups of a CU
    skew_prio = value;
  } else if (!std::strcmp(key, "skew_blocks")) {   // 0 = default: workgroups per CU the sweep kernel is built for
#ifdef CICE4_AMD_EXPERIMENTS
    CICE_REQUIRE(value >= 0, "skew_blocks must be >= 0");
#else
    CICE_REQUIRE(value == 0 || value == 3, "skew_blocks must be 0 or 3 (2: in -DCICE4_AMD_EXPERIMENTS builds only)");
#endif
    skew_blocks_opt = value;
  } else if (!std::strcmp(key, "skew_seg_rows")) { // 0 = auto: rows a workgroup owns
    CICE_REQUIRE(value >= 0, "skew_seg_rows must be >= 0");
    skew_seg_opt = value;
  } else if (!std::strcmp(key, "skew_min_cells")) { // grids below this size keep k_subcycle2
    CICE_REQUIRE(value >= 0, "skew_min_cells must be >= 0");
    skew_min_cells = value;
  } else if (!std::strcmp(key, "fused_waves")) {   // 0 = auto
    CICE_REQUIRE(value == 0 || value == 8 || value == 12 || value == 13 || value == 14 || value == 16,
                 "fused_waves must be 0, 8, 12, 13, 14 or 16");
    waves2 = value;
  } else {
    throw Error{CICE_EINVAL, std::string("unknown option ") + key};
  }
  std::memset(strips_cache, 0, sizeof(strips_cache));   // (the strip layout depends on K and on the workgroup shape)
  drop_graph();
}

void Evp::init(const cice_evp_config& c, const cice_evp_grid& g) {
  CICE_REQUIRE(dom.nblocks() > 0, "cice_evp_init: no local blocks (call cice_domain_create first)");
  CICE_REQUIRE(c.ndte >= 1, "ndte must be >= 1");
  // the subcycle kernels address a block plane with 32-bit byte offsets
  CICE_REQUIRE((size_t)dom.nx_block * dom.ny_block * 8 < (1ull << 32), "cice_evp_init: block plane of 4 GB or more");
  cfg = c;
  n = (size_t)dom.nblocks() * dom.nx_block * dom.ny_block;
  std::vector<int32_t> hb;
  for (int gid : dom.local) {
    const Block& b = dom.all[gid];
    hb.insert(hb.end(), {b.ilo, b.ihi, b.jlo, b.jhi, b.own_jlo, b.own_jhi});
  }
  blk.alloc(hb.size());
  blk.upload(hb.data(), stream);
  {
    // Are the on-rank ghost copies exactly the east-west wrap of full-width blocks -- every cell (ilo, j) mirrored by
    // (ihi+1, j), every (ihi, j) by (ilo-1, j), j = jlo..jhi, and nothing else?  Then a kernel can forward a new
    // velocity to its ghost cells without the table.
    const int nx = dom.nx_block, ny = dom.ny_block;
    const size_t np = (size_t)nx * ny;
    bool ok = dom.ew == BND_CYCLIC && dom.nbx == 1;
    size_t expect = 0;
    for (int lb = 0; lb < dom.nblocks() && ok; ++lb) {
      const Block& b = dom.all[dom.local[lb]];
      expect += 2 * (size_t)(b.jhi - b.jlo + 1);
    }
    ok = ok && dom.hsrc.size() == expect;
    for (size_t e = 0; e < dom.hsrc.size() && ok; ++e) {
      const size_t sb = (size_t)dom.hsrc[e] / np, db = (size_t)dom.hdst[e] / np;
      const int sq = (int)((size_t)dom.hsrc[e] - sb * np), dq = (int)((size_t)dom.hdst[e] - db * np);
      const int si = sq % nx + 1, sj = sq / nx + 1, di = dq % nx + 1, dj = dq / nx + 1;
      const Block& b = dom.all[dom.local[sb]];
      ok = sb == db && sj == dj && sj >= b.jlo && sj <= b.jhi &&
           ((si == b.ilo && di == b.ihi + 1) || (si == b.ihi && di == b.ilo - 1));
    }
    fwd_is_ew_wrap = ok;
  }
  // the eight read-only inputs of the momentum equation share one allocation (k_subcycle_skew: one base pointer)
  uarena.alloc(8 * n);
  {
    DevBuf<double>* us8[8] = {&aiu, &uocn, &vocn, &forcex, &forcey, &umassdtei, &fm, &uarear};
    for (int k = 0; k < 8; ++k) us8[k]->view(uarena.p + (size_t)k * n, n);
  }
  struct G { DevBuf<double>* d; const double* h; };
  G gs[] = {{&dxt, g.dxt}, {&dyt, g.dyt}, {&dxhy, g.dxhy}, {&dyhx, g.dyhx}, {&cxp, g.cxp},
            {&cyp, g.cyp}, {&cxm, g.cxm}, {&cym, g.cym}, {&tarea, g.tarea}, {&uarea, g.uarea},
            {&tarear, g.tarear}, {&uarear, g.uarear}, {&tinyarea, g.tinyarea}, {&fcor, g.fcor}};
  for (G& x : gs) {
    CICE_REQUIRE(x.h != nullptr, "cice_evp_init: NULL grid array");
    x.d->alloc(n);
    x.d->upload(x.h, stream);
  }
  CICE_REQUIRE(g.tmask && g.umask, "cice_evp_init: NULL mask");
  tmask.alloc(n); tmask.upload(g.tmask, stream);
  umask.alloc(n); umask.upload(g.umask, stream);
  // Optional primary lengths: enable metric derivation only if it reproduces the nine metric
  // arrays BIT FOR BIT on every ocean T-cell the stress kernel can touch (ilo..ihi+1, jlo..jhi+1).
  derive_ok = false;
  if (g.HTN && g.HTE) {
    bool same = true;
    const int nx = dom.nx_block, ny = dom.ny_block;
    for (int lb = 0; lb < dom.nblocks() && same; ++lb) {
      const Block& b = dom.all[dom.local[lb]];
      const size_t base = (size_t)lb * nx * ny;
      for (int j = b.jlo; j <= b.jhi + 1 && same; ++j)
        for (int i = b.ilo; i <= b.ihi + 1; ++i) {
          const size_t q = base + (size_t)(j - 1) * nx + (i - 1);
          if (!g.tmask[q]) continue;
          const double hn = g.HTN[q], hs = g.HTN[q - nx], he = g.HTE[q], hw = g.HTE[q - 1];
          const double Dxt = p5 * (hn + hs), Dyt = p5 * (he + hw);
          same = Dxt == g.dxt[q] && Dyt == g.dyt[q] && p5 * (he - hw) == g.dxhy[q] &&
                 p5 * (hn - hs) == g.dyhx[q] && (1.5 * he - p5 * hw) == g.cyp[q] &&
                 (1.5 * hn - p5 * hs) == g.cxp[q] && -(1.5 * hw - p5 * he) == g.cym[q] &&
                 -(1.5 * hs - p5 * hn) == g.cxm[q] && puny * (Dxt * Dyt) == g.tinyarea[q];
          if (!same) break;
        }
    }
    if (same) {
      HTN.alloc(n); HTN.upload(g.HTN, stream);
      HTE.alloc(n); HTE.upload(g.HTE, stream);
      derive_ok = true;
    }
  }
  for (DevBuf<double>* d : {&aice, &vice, &vsno, &aice0, &strairxT, &strairyT, &uocn, &vocn, &ss_tltx,
                            &ss_tlty, &fm, &strtltx, &strtlty, &strocnx, &strocny, &strintx, &strinty,
                            &strairx, &strairy, &strength, &divu, &shear, &rdg_conv, &rdg_shear,
                            &prs_sig, &strocnxT, &strocnyT, &tmass, &umass, &aiu, &umassdtei, &waterx,
                            &watery, &forcex, &forcey, &work1}) {
    d->alloc(n);
    d->zero(stream);
  }
  work1.alloc(2 * n);
  aicen.alloc(NCAT * n); aicen.zero(stream);
  vicen.alloc(NCAT * n); vicen.zero(stream);
  for (int k = 0; k < 2; ++k) {  // init_evp :487-520: velocities, stresses = 0, iceumask = F
    st[k].alloc(14 * n); st[k].zero(stream);   // u, v, 12 stresses contiguous: one halo message
    uv[k].p = st[k].p;
    sig[k].p = st[k].p + 2 * n;
  }
  iceumask.alloc(n); iceumask.zero(stream);
  icetmask.alloc(n); icetmask.zero(stream);
  counters.alloc(2);
  // Tile shape by problem size (measured): small grids cannot fill the 1,024 SIMDs
  // and are latency-bound -> many small workgroups, one row per wavefront; large grids are
  // bandwidth-bound -> two rows per wavefront (fewer redundant overlap rows, same occupancy).
  {
    const long long cells = (long long)dom.nblocks() * (dom.nx_block - 2) * (dom.ny_block - 2);
    waves = 4;
    rows_per_wave = cells <= 400LL * 400LL ? 1 : 2;
  }
  cur = 0;
  CICE_HIP(hipStreamSynchronize(stream));
  ready = true;
  prepared = false;
  drop_graph();
}

void Evp::upload(const cice_evp_fields& f) {
  upload_some(f, 0);
  CICE_HIP(hipStreamSynchronize(stream));
}

// skip_io 1: the device copies of u, v, the stresses and iceumask are current; 2: and the seven flux fields (fm, strtlt,
// strocn, strint) are zero on the host (the caller's statement), so they are zeroed here instead of uploaded
void Evp::upload_some(const cice_evp_fields& f, int skip_io) {
  CICE_REQUIRE(ready, "cice_evp_upload before cice_evp_init");
  io_valid = false;
  struct U { DevBuf<double>* d; const double* h; bool io; };
  U us[] = {{&aice, f.aice, false}, {&vice, f.vice, false}, {&vsno, f.vsno, false}, {&aice0, f.aice0, false},
            {&aicen, f.aicen, false}, {&vicen, f.vicen, false}, {&strairxT, f.strairxT, false},
            {&strairyT, f.strairyT, false}, {&uocn, f.uocn, false}, {&vocn, f.vocn, false},
            {&ss_tltx, f.ss_tltx, false}, {&ss_tlty, f.ss_tlty, false}, {&fm, f.fm, true},
            {&strtltx, f.strtltx, true}, {&strtlty, f.strtlty, true}, {&strocnx, f.strocnx, true},
            {&strocny, f.strocny, true}, {&strintx, f.strintx, true}, {&strinty, f.strinty, true}};
  fan.fork(stream);
  for (U& x : us) {
    // after adopt_state the six state fields are on the device already: a NULL pointer keeps them
    const bool state6 = x.d == &aice || x.d == &vice || x.d == &vsno || x.d == &aice0 || x.d == &aicen || x.d == &vicen;
    if (adopted && state6 && x.h == nullptr) continue;
#ifndef CICE4_AMD_AUSCOM   // the stand-alone build forms the tilt from the currents (:919-922) and never reads the slope
    if (x.d == &ss_tltx || x.d == &ss_tlty) continue;
#endif
    if (skip_io == 2 && x.io) {
      CICE_HIP(hipMemsetAsync(x.d->p, 0, n * 8, stream));   // (on the main stream, which every later kernel follows)
      continue;
    }
    CICE_REQUIRE(x.h != nullptr, "cice_evp_upload: NULL field");
    x.d->upload(x.h, fan.next());
  }
  adopted = false;
  if (!skip_io) {
    CICE_REQUIRE(f.uvel && f.vvel && f.iceumask, "cice_evp_upload: NULL field");
    CICE_HIP(hipMemcpyAsync(uv[cur].p, f.uvel, n * 8, hipMemcpyHostToDevice, fan.next()));
    CICE_HIP(hipMemcpyAsync(uv[cur].p + n, f.vvel, n * 8, hipMemcpyHostToDevice, fan.next()));
    const double* hs[12] = {f.stressp_1, f.stressp_2, f.stressp_3, f.stressp_4, f.stressm_1, f.stressm_2,
                            f.stressm_3, f.stressm_4, f.stress12_1, f.stress12_2, f.stress12_3,
                            f.stress12_4};
    for (int c = 0; c < 12; ++c) {
      CICE_REQUIRE(hs[c] != nullptr, "cice_evp_upload: NULL stress");
      CICE_HIP(hipMemcpyAsync(sig[cur].p + (size_t)c * n, hs[c], n * 8, hipMemcpyHostToDevice, fan.next()));
    }
    iceumask.upload(f.iceumask, fan.next());
  }
  fan.join();
  prepared = false;
}

void Evp::download(cice_evp_fields& f) {
  CICE_REQUIRE(ready, "cice_evp_download before cice_evp_init");
  fan.fork(stream);
  download_some(f, 3);
  download_stresses(f);
}

// part 1: the fields prepare() leaves final (nothing in the subcycle loop or in finish() writes them); part 2: the rest
// but the stresses.  Enqueued on the side streams: the caller has forked them.
void Evp::download_some(cice_evp_fields& f, int part) {
  struct D { const DevBuf<double>* d; double* h; int part; };
  D ds[] = {{&fm, f.fm, 1}, {&strtltx, f.strtltx, 1}, {&strtlty, f.strtlty, 1}, {&strairx, f.strairx, 1},
            {&strairy, f.strairy, 1}, {&strength, f.strength, 1}, {&strocnx, f.strocnx, 2},
            {&strocny, f.strocny, 2}, {&strintx, f.strintx, 2}, {&strinty, f.strinty, 2}, {&divu, f.divu, 2},
            {&shear, f.shear, 2}, {&rdg_conv, f.rdg_conv, 2}, {&rdg_shear, f.rdg_shear, 2}, {&prs_sig, f.prs_sig, 2},
            {&strocnxT, f.strocnxT, 2}, {&strocnyT, f.strocnyT, 2}};
  if (part & 2) {
    CICE_HIP(hipMemcpyAsync(f.uvel, uv[cur].p, n * 8, hipMemcpyDeviceToHost, fan.next()));
    CICE_HIP(hipMemcpyAsync(f.vvel, uv[cur].p + n, n * 8, hipMemcpyDeviceToHost, fan.next()));
  }
  if (part & 1) iceumask.download(f.iceumask, fan.next());
  for (D& x : ds)
    if (x.h && (x.part & part)) x.d->download(x.h, fan.next());
}

// (with the side streams forked by the caller, or not: then it forks them itself)
void Evp::download_stresses(cice_evp_fields& f) {
  CICE_REQUIRE(ready, "cice_evp_download before cice_evp_init");
  if (!fan.forked) fan.fork(stream);
  double* hs[12] = {f.stressp_1, f.stressp_2, f.stressp_3, f.stressp_4, f.stressm_1, f.stressm_2,
                    f.stressm_3, f.stressm_4, f.stress12_1, f.stress12_2, f.stress12_3, f.stress12_4};
  for (int c = 0; c < 12; ++c) {
    CICE_REQUIRE(hs[c] != nullptr, "cice_evp_download: NULL stress");
    CICE_HIP(hipMemcpyAsync(hs[c], sig[cur].p + (size_t)c * n, n * 8, hipMemcpyDeviceToHost, fan.next()));
  }
  fan.join();
  CICE_HIP(hipStreamSynchronize(stream));
}

// cice_evp: `call evp(dt)` on host arrays (ice_dyn_evp.F90:119-432) as one pipeline.  The link is idle while the subcycle loop
// runs (0.6 ms at gx1 size, as long as all the copies together): the six fields prepare() leaves final, and iceumask, go down
// then.  "keep_state" 1: uvel, vvel, the 12 stresses and iceumask are where the last call left them -- the caller has not
// changed them on the host since (in the reference nothing but evp itself and the restart reader, before the first step,
// writes them) -- and are not uploaded again; 2: and the seven flux fields evp reads back outside its ice mask (fm,
// strtltx/y, strocnx/y, strintx/y) are ZERO on the host when evp is called, as init_history_dyn (ice_flux.F90:585-602,
// called at the top of every step: drivers/cice4/CICE_RunMod.F90) leaves them: zeroed on the device, not uploaded.
// "lazy_stresses": the stresses stay on the device until someone asks (download_stresses: the reference reads them on the
// host only for its history and restart files).
void Evp::run(double dt, cice_evp_fields& f, const std::function<void()>& while_looping) {
  upload_some(f, io_valid ? keep_state : 0);
  prepare(dt);
  // (not where the loop waits for other ranks' loops: ranks that share ONE device -- the rehearsals of tests/ranks_case.py --
  //  need a hardware queue each for their loops, and copy streams that are busy at that moment take queues away)
  const bool early = !halo.multi_rank();
  if (early) {
    fan.fork(stream);        // the side streams wait for prepare()
    download_some(f, 1);
    fan.detach();            // ... and nobody waits for them yet
  }
  if (while_looping) while_looping();   // (cice_transport_chain: the transport's state travels up now)
  subcycles(1, sc.ndte, nullptr);
  finish();
  fan.fork(stream);          // (the side streams run in order: the copies above are done before these)
  download_some(f, early ? 2 : 3);
  if (lazy_sig) {
    fan.join();
    CICE_HIP(hipStreamSynchronize(stream));
  } else {
    download_stresses(f);
  }
  io_valid = true;
}

void Evp::prepare(double dt) {
  CICE_REQUIRE(ready, "cice_evp_prepare before cice_evp_init");
  if (res_retry_in > 0 && --res_retry_in == 0) {   // a time-out of the one-launch loop, res_retry_steps calls ago
    resident_failed = false;
    res_level = 0;
  }
  EvpScalars nsc;
  nsc.set(dt, cfg.ndte, cfg.evp_damping);
  if (std::memcmp(&nsc, &sc, sizeof(sc)) != 0) drop_graph();  // scalars are baked into the graph
  sc = nsc;
  PrepArgs a{};
  a.sc = sc; a.nx = dom.nx_block; a.ny = dom.ny_block; a.nblocks = dom.nblocks(); a.n = n;
  a.blk = blk.p; a.tmask = tmask.p; a.umask = umask.p;
  a.aice = aice.p; a.vice = vice.p; a.vsno = vsno.p; a.aice0 = aice0.p; a.aicen = aicen.p;
  a.vicen = vicen.p; a.strairxT = strairxT.p; a.strairyT = strairyT.p; a.uocn = uocn.p; a.vocn = vocn.p;
  a.ss_tltx = ss_tltx.p; a.ss_tlty = ss_tlty.p;
  a.tarea = tarea.p; a.uarea = uarea.p; a.fcor = fcor.p;
  a.strairx = strairx.p; a.strairy = strairy.p; a.tmass = tmass.p; a.umass = umass.p; a.aiu = aiu.p;
  a.work1 = work1.p; a.icetmask = icetmask.p; a.iceumask = iceumask.p;
  a.rdg_conv = rdg_conv.p; a.rdg_shear = rdg_shear.p; a.divu = divu.p; a.shear = shear.p;
  a.prs_sig = prs_sig.p; a.umassdtei = umassdtei.p; a.waterx = waterx.p; a.watery = watery.p;
  a.forcex = forcex.p; a.forcey = forcey.p; a.fm = fm.p; a.strtltx = strtltx.p; a.strtlty = strtlty.p;
  a.strocnx = strocnx.p; a.strocny = strocny.p; a.strintx = strintx.p; a.strinty = strinty.p;
  a.strength = strength.p; a.strocnxT = strocnxT.p; a.strocnyT = strocnyT.p;
  a.u = uv[cur].p; a.v = uv[cur].p + n; a.sig = sig[cur].p;
  a.kstrength = cfg.kstrength; a.krdg_partic = cfg.krdg_partic; a.krdg_redist = cfg.krdg_redist;
  a.mu_rdg = cfg.mu_rdg; a.counters = counters.p;
  const dim3 g = grid1(n), blk256(256);
  hipLaunchKernelGGL(k_prep1, g, blk256, 0, stream, a);                       // :214-244
  halo.update_i4(icetmask.p, 1, n, LOC_CENTER, KIND_SCALAR);                  // :250-253
  hipLaunchKernelGGL(k_to_ugrid2, g, blk256, 0, stream, a, (const double*)tmass.p,
                     (const double*)aice.p, umass.p, aiu.p);                  // :259-260
  // t2ugrid_vector(strairx), (strairy) :276-277 = copy, halo update, to_ugrid
  CICE_HIP(hipMemcpyAsync(work1.p, strairx.p, n * 8, hipMemcpyDeviceToDevice, stream));
  CICE_HIP(hipMemcpyAsync(work1.p + n, strairy.p, n * 8, hipMemcpyDeviceToDevice, stream));
  halo.update_r8(work1.p, 2, n, true, LOC_CENTER, KIND_VECTOR);               // ice_grid.F90:1565
  hipLaunchKernelGGL(k_to_ugrid2, g, blk256, 0, stream, a, (const double*)work1.p,
                     (const double*)(work1.p + n), strairx.p, strairy.p);
  hipLaunchKernelGGL(k_prep2, g, blk256, 0, stream, a);                       // :280-316
  hipLaunchKernelGGL(k_strength, g, blk256, 0, stream, a);                    // :322-332
  halo.update_r8(strength.p, 1, n, true, LOC_CENTER, KIND_SCALAR);            // :337
  if (dom.overlap > 0) {   // overlap rows of u, v AND sigma from their owners; a tripole fold only ever touches u, v
    halo.update_r8(st[cur].p, 14, n, true, LOC_CENTER, KIND_SCALAR, 0.0, HALO_COPIES);
    if (halo.has_fold()) halo.update_r8(uv[cur].p, 2, n, false, LOC_NECORNER, KIND_VECTOR, 0.0, HALO_FOLD);
  }
  else halo.update_r8(uv[cur].p, 2, n, true, LOC_NECORNER, KIND_VECTOR);      // :340-343
  // both copies of the double-buffered fields start out identical: cells the subcycle
  // kernel never writes (outside the masks) then hold the same value in either copy
  CICE_HIP(hipMemcpyAsync(st[1 - cur].p, st[cur].p, 14 * n * 8, hipMemcpyDeviceToDevice, stream));
  copies_identical = true;   // (until the first subcycle kernel writes one of them)
  res_map_stale = true;      // new masks: the one-launch loop chooses its tile map again
  flips = 0;
  skew_packed = false;
  if (can_skew() || can_skew_fold()) skew_pack();   // the sweep kernel's interleaved inputs of this step
  CICE_HIP(hipGetLastError());
  prepared = true;
  counted = false;   // the diagnostic counts are formed when somebody asks (active_cells)
}

// the sweep kernel's interleaved read-only inputs (SkewArgs::uar4 / hnhe / msk) from this step's work arrays
void Evp::skew_pack() {
  if (uar4.n < 8 * n) {
    uar4.alloc(8 * n);
    skew_msk.alloc(n);
    hnhe.alloc(2 * n);
    hipLaunchKernelGGL(k_skew_pack_grid, grid1(n), dim3(256), 0, stream, n, (const double*)HTN.p, (const double*)HTE.p, hnhe.p);
  }
  hipLaunchKernelGGL(k_skew_pack, grid1(n), dim3(256), 0, stream, n, (const double*)uarena.p, (const int32_t*)icetmask.p,
                     (const int32_t*)iceumask.p, uar4.p, skew_msk.p);
  // rows with anything to compute, per column strip (this step's masks)
  rowact_strips = 0;
  if (rowact_on()) {
    const int K = skew_levels(), rows = dom.ny_block - 2;
    int shift = 0;
    const int strips = skew_strips(K, &shift);
    const size_t want = (size_t)dom.nblocks() * strips * rows;
    if (rowact.n < want) rowact.alloc(want);
    hipLaunchKernelGGL(k_skew_rowact, dim3((unsigned)((rows + 3) / 4), (unsigned)strips, (unsigned)dom.nblocks()), dim3(256), 0,
                       stream, K, strips, shift, dom.nx_block, dom.ny_block, 0, (const int32_t*)blk.p,
                       (const int32_t*)skew_msk.p, rowact.p);
    if (run_next.n < want) { run_next.alloc(want); run_end.alloc(want); }
    const int ns = dom.nblocks() * strips;
    hipLaunchKernelGGL(k_skew_runs, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, stream, ns, rows, 3 * K,
                       (const unsigned char*)rowact.p, run_next.p, run_end.p);
    rowact_strips = strips;
    rowact_k = K;
    rowact_host_stale = true;
  }
  skew_packed = true;
}

int Evp::resident_map() {
  if (res_map.n == 0 || res_map_stale) return -1;
  int32_t v = -1;
  CICE_HIP(hipStreamSynchronize(stream));
  CICE_HIP(hipMemcpy(&v, res_map.p, 4, hipMemcpyDeviceToHost));
  return v;
}

// Workgroups of the sweep shrink to the rows that hold ice (SkewArgs::rowact).  Not with three wavefronts per level (its
// strips are laid out differently).  Option "skew_rowact" / CICE4_AMD_SKEW_ROWACT=0|1.
bool Evp::rowact_on() const {
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_ROWACT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  if (!(env >= 0 ? env == 1 : rowact_opt)) return false;
  return skew_subs(skew_levels()) == 1;
}

// test aid: start / end wall-clock ticks (10 ns) of the workgroups of the last k_subcycle_skew launch
long long Evp::debug_read(const char* what, long long* out, long long cap) {
  const bool stamps = !std::strcmp(what, "stamps");
  if (!std::strcmp(what, "skew_rows")) {   // the sweep's segments as they stand: [tiles][3] strip, first / last U-row relative to jlo
    if (bal_nt > 0) {
      const long long nn = 3LL * bal_nt;
      for (long long i = 0; out && i < std::min(nn, cap); ++i) {
        const BalTile& bt = bal_tiles[(size_t)(i / 3)];
        out[i] = i % 3 == 0 ? bt.strip : (i % 3 == 1 ? bt.first : bt.last);
      }
      return nn;
    }
    const long long nt = (long long)rows_host.size() / 2, nn = 3 * nt, sx = std::max(1, bal_tiles_x);
    for (long long i = 0; out && i < std::min(nn, cap); ++i)
      out[i] = i % 3 == 0 ? (i / 3) % sx : rows_host[(size_t)(2 * (i / 3) + (i % 3 - 1))];
    return nn;
  }
  CICE_REQUIRE(stamps || !std::strcmp(what, "skew_times"), "unknown debug array");
  CICE_HIP(hipStreamSynchronize(stream));
  DevBuf<long long>& b = stamps ? stamp_buf : skew_dbg;
  const long long nn = stamps ? (long long)stamp_used : (long long)b.n;
  if (out && nn) CICE_HIP(hipMemcpy(out, b.p, (size_t)std::min(nn, cap) * 8, hipMemcpyDeviceToHost));
  return nn;
}

// [4 * workgroups] cycle / wall-clock stamps of the last launch of a stamped kernel, or NULL (option "stamps" off, or the
// product build, whose kernels hold no stamp: cice_evp_debug("stamps") then returns zeros)
long long* Evp::stamp_buffer(size_t workgroups) {
  if (!stamps_on) return nullptr;
  if (stamp_buf.n < 4 * workgroups) stamp_buf.alloc(4 * workgroups);
  stamp_used = 4 * workgroups;
  CICE_HIP(hipMemsetAsync(stamp_buf.p, 0, stamp_used * 8, stream));
  return stamp_buf.p;
}

// aggregate (source/ice_itd.F90:279-): the category sums the dynamics read, formed in the reference's order
__global__ __launch_bounds__(256) void k_aggregate(size_t np, int nb, int ncat, const double* __restrict__ aicen,
                                                   const double* __restrict__ vicen, const double* __restrict__ vsnon,
                                                   double* __restrict__ aice, double* __restrict__ vice,
                                                   double* __restrict__ vsno, double* __restrict__ aice0) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np * nb) return;
  const size_t b = t / np, q = t - b * np;
  double a = c0, v = c0, s = c0;
  for (int n = 0; n < ncat; ++n) {
    const size_t c = (b * ncat + n) * np + q;
    a = a + aicen[c];
    v = v + vicen[c];
    s = s + vsnon[c];
  }
  aice[t] = a; vice[t] = v; vsno[t] = s;
  aice0[t] = fmax(c1 - a, c0);
}

// The ice state of the dynamics taken from arrays that are ALREADY on the device (the batched thermodynamic step's):
// aicen, vicen copied, aice, vice, vsno, aice0 aggregated here; the next upload() leaves those six alone.
void Evp::adopt_state(const double* d_aicen, const double* d_vicen, const double* d_vsnon) {
  CICE_REQUIRE(ready, "cice_evp_adopt_thermo_state before cice_evp_init");
  CICE_HIP(hipMemcpyAsync(aicen.p, d_aicen, (size_t)NCAT * n * 8, hipMemcpyDeviceToDevice, stream));
  CICE_HIP(hipMemcpyAsync(vicen.p, d_vicen, (size_t)NCAT * n * 8, hipMemcpyDeviceToDevice, stream));
  const size_t np = (size_t)dom.nx_block * dom.ny_block;
  hipLaunchKernelGGL(k_aggregate, grid1(n), dim3(256), 0, stream, np, dom.nblocks(), NCAT, (const double*)aicen.p,
                     (const double*)vicen.p, d_vsnon, aice.p, vice.p, vsno.p, aice0.p);
  CICE_HIP(hipGetLastError());
  adopted = true;
}

bool Evp::derives_metrics() const { return derive_ok && derive_on; }

void Evp::active_cells(long long* nt, long long* nu) {
  CICE_REQUIRE(prepared, "cice_evp_active_cells before cice_evp_prepare");
  if (!counted) {   // the masks do not change during the subcycling
    PrepArgs a{};
    a.nx = dom.nx_block; a.ny = dom.ny_block; a.nblocks = dom.nblocks(); a.n = n; a.blk = blk.p;
    a.icetmask = icetmask.p; a.iceumask = iceumask.p; a.counters = counters.p;
    counters.zero(stream);
    hipLaunchKernelGGL(k_count_active, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, stream, a);
    CICE_HIP(hipGetLastError());
    counted = true;
  }
  unsigned long long h[2];
  counters.download(h, stream);
  CICE_HIP(hipStreamSynchronize(stream));
  if (nt) *nt = (long long)h[0];
  if (nu) *nu = (long long)h[1];
}

template <int W, int R, bool DERIVE>
static void launch_wrd(const SubArgs& a, bool last, bool damp, dim3 g, hipStream_t s) {
  const dim3 blk(64 * W);
  if (last) {
    if (damp) hipLaunchKernelGGL((k_subcycle<W, R, true, true, DERIVE>), g, blk, 0, s, a);
    else hipLaunchKernelGGL((k_subcycle<W, R, true, false, DERIVE>), g, blk, 0, s, a);
  } else {
    if (damp) hipLaunchKernelGGL((k_subcycle<W, R, false, true, DERIVE>), g, blk, 0, s, a);
    else hipLaunchKernelGGL((k_subcycle<W, R, false, false, DERIVE>), g, blk, 0, s, a);
  }
}

template <int W, int R>
static void launch_wr(const SubArgs& a, bool last, bool damp, dim3 g, hipStream_t s) {
  if (a.HTN) launch_wrd<W, R, true>(a, last, damp, g, s);
  else launch_wrd<W, R, false>(a, last, damp, g, s);
}

SubArgs Evp::make_args() const {
  SubArgs a{};
  a.sc = sc; a.nx = dom.nx_block; a.ny = dom.ny_block; a.n = n; a.nblocks = dom.nblocks();
  a.ew_cyclic = dom.ew == BND_CYCLIC ? 1 : 0;
  const bool fwd = halo.fwd_ok();
  a.ring_slot = fwd ? halo.d_ring_slot() : nullptr; a.fwd = halo.d_fwd();
  a.carry_top = halo.has_fold() ? 1 : 0;
  a.blk = blk.p; a.icetmask = icetmask.p; a.iceumask = iceumask.p;
  a.u_in = uv[cur].p; a.v_in = uv[cur].p + n; a.u_out = uv[1 - cur].p; a.v_out = uv[1 - cur].p + n;
  a.sig_in = sig[cur].p; a.sig_out = sig[1 - cur].p;
  for (int c = 0; c < 12; ++c) {
    a.sig_in_p[c] = a.sig_in + (size_t)c * n;
    a.sig_out_p[c] = a.sig_out + (size_t)c * n;
  }
  a.dxt = dxt.p; a.dyt = dyt.p; a.dxhy = dxhy.p; a.dyhx = dyhx.p; a.cxp = cxp.p; a.cyp = cyp.p;
  a.cxm = cxm.p; a.cym = cym.p; a.tarear = tarear.p; a.tinyarea = tinyarea.p; a.strength = strength.p;
  const bool dv = derive_ok && derive_on;
  a.HTN = dv ? HTN.p : nullptr; a.HTE = dv ? HTE.p : nullptr;
  a.aiu = aiu.p; a.uocn = uocn.p; a.vocn = vocn.p; a.waterx = waterx.p; a.watery = watery.p;
  a.forcex = forcex.p; a.forcey = forcey.p; a.umassdtei = umassdtei.p; a.fm = fm.p; a.uarear = uarear.p;
  a.divu = divu.p; a.rdg_conv = rdg_conv.p; a.rdg_shear = rdg_shear.p; a.shear = shear.p;
  a.prs_sig = prs_sig.p; a.strintx = strintx.p; a.strinty = strinty.p; a.strocnx = strocnx.p;
  a.strocny = strocny.p;
  return a;
}

void Evp::launch_subcycle(int ksub) {
  ++loop_launches;
  SubArgs a = make_args();
  const int trows = waves * rows_per_wave;
  // physical extent of a block (a wide-halo slab is bsy + 2*overlap rows tall)
  a.tiles_x = ((dom.nx_block - 2) + (TX - 1) - 1) / (TX - 1);
  a.tiles_y = ((dom.ny_block - 2) + (trows - 1) - 1) / (trows - 1);
  const int nt = a.tiles_x * a.tiles_y * a.nblocks;
  const dim3 g(8 * ((nt + 7) / 8));
  const bool last = (ksub == sc.ndte), damp = sc.evp_damping != 0;
  const int key = waves * 100 + rows_per_wave;
  switch (key) {
    case 801: launch_wr<8, 1>(a, last, damp, g, stream); break;
    case 802: launch_wr<8, 2>(a, last, damp, g, stream); break;
    case 804: launch_wr<8, 4>(a, last, damp, g, stream); break;
    case 401: launch_wr<4, 1>(a, last, damp, g, stream); break;
    case 402: launch_wr<4, 2>(a, last, damp, g, stream); break;
    case 404: launch_wr<4, 4>(a, last, damp, g, stream); break;
    case 408: launch_wr<4, 8>(a, last, damp, g, stream); break;
    case 1601: launch_wr<16, 1>(a, last, damp, g, stream); break;
    case 1602: launch_wr<16, 2>(a, last, damp, g, stream); break;
    default: throw Error{CICE_EINVAL, "unsupported (waves, rows_per_wave) combination"};
  }
  after_subcycle(ksub);
}

// The double-buffered fields now live in the other copy; ghost cells owned elsewhere.
void Evp::after_subcycle(int ksub) {
  const bool fwd = halo.fwd_ok();
  cur = 1 - cur;
  ++flips;
  copies_identical = false;
  // On-rank ghost cells were written by the kernel itself.  Rows owned by other blocks/ranks:
  // classic domain -> every subcycle (:397-402); wide-halo domain -> u, v and sigma every
  // `overlap` subcycles and after the last one (the overlap rows are recomputed in between and
  // lose one valid row per side per subcycle).
  if (dom.overlap > 0) {
    // (a tripole fold on the top slab: the refresh moves whole rows of all 14 planes and folds nothing; the fold of
    //  u, v follows it, after every subcycle, as on any tripole grid)
    if (ksub % dom.overlap == 0 || ksub == sc.ndte) halo.update_r8(st[cur].p, 14, n, /*wrap=*/!fwd, LOC_CENTER, KIND_SCALAR, 0.0, HALO_COPIES);
    else if (!fwd) halo.update_r8(uv[cur].p, 2, n, true, LOC_CENTER, KIND_SCALAR, 0.0, HALO_COPIES);
    if (halo.has_fold()) halo.update_r8(uv[cur].p, 2, n, false, LOC_NECORNER, KIND_VECTOR, 0.0, HALO_FOLD);
  } else if (halo.has_refresh() || !fwd || halo.has_fold()) {
    halo.update_r8(uv[cur].p, 2, n, /*wrap=*/!fwd, LOC_NECORNER, KIND_VECTOR);   // :397-402
  }
}

template <int W, bool DERIVE>
static void launch2_wd(const SubArgs& a, bool last, bool damp, dim3 g, hipStream_t s) {
  const dim3 blk(64 * W);
  if (last) {
    if (damp) hipLaunchKernelGGL((k_subcycle2<W, true, true, DERIVE>), g, blk, 0, s, a);
    else hipLaunchKernelGGL((k_subcycle2<W, true, false, DERIVE>), g, blk, 0, s, a);
  } else {
    if (damp) hipLaunchKernelGGL((k_subcycle2<W, false, true, DERIVE>), g, blk, 0, s, a);
    else hipLaunchKernelGGL((k_subcycle2<W, false, false, DERIVE>), g, blk, 0, s, a);
  }
}

template <int W>
static void launch2_w(const SubArgs& a, bool last, bool damp, dim3 g, hipStream_t s) {
  if (a.HTN) launch2_wd<W, true>(a, last, damp, g, s);
  else launch2_wd<W, false>(a, last, damp, g, s);
}

// Two subcycles per launch are possible when no ghost ROW of a local block changes between two
// consecutive subcycles: every block spans the full width (its E/W ghost columns mirror itself or
// lie beyond an open edge and are handled inside the kernel), and its N/S ghost rows either lie
// beyond an open/closed domain edge or belong to a wide-halo slab that is refreshed only every
// `overlap` (even) subcycles.
bool Evp::can_fuse() const {
  if (!fuse_on || !halo.fwd_ok()) return false;
  if (dom.nbx != 1) return false;
  if (dom.overlap > 0) return dom.overlap % 2 == 0 && !halo.has_fold();   // (the top slab of a tripole grid folds after every subcycle)
  // a tripole fold rewrites the top row and its ghost row after every subcycle
  return dom.nby == 1 && dom.ns != BND_CYCLIC && !dom.tripole() && !halo.has_refresh();
}

int Evp::fused_waves() const {
  if (waves2) return waves2;
  if (waves2_auto) return waves2_auto;   // domain and device do not change under an Evp
  // Workgroups are dealt evenly to the CUs, so a launch lasts about ceil(workgroups / CUs) x W
  // wavefront-times (measured: gx3 9.9 / 12.5 / 15.4 us, gx1 17.6 / 25.4 / 17.8 us, 0.1 degree
  // 944 / 968 / 857 us for W = 8 / 12 / 16).  Taller workgroups own a larger share of their rows
  // ((W-3)/W) but quantise worse on small grids.  W = 13 is there for gx1: 234 workgroups, one round,
  // and its four SIMDs carry 4 (two of them the light rim wavefronts), 3, 3, 3 wavefronts: +4 %.
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const long long cols = dom.nx_block - 1, rows = dom.ny_block - 1;
  const long long tx = (cols + OWN_LANES - 1) / OWN_LANES;
  int best = 8;
  long long best_cost = -1;
  for (int w : {8, 16, 12, 13, 14}) {
    const long long wg = tx * ((rows + (w - 3) - 1) / (w - 3)) * dom.nblocks();
    const long long cost = ((wg + ncu - 1) / ncu) * w;
    if (best_cost < 0 || cost < best_cost) {
      best = w;
      best_cost = cost;
    }
  }
  waves2_auto = best;
  return best;
}

// subcycles ksub and ksub+1
void Evp::launch_subcycle_pair(int ksub) {
  ++loop_launches;
  SubArgs a = make_args();
  const int W = fused_waves();
  a.tiles_x = ((dom.nx_block - 2) + 1 + OWN_LANES - 1) / OWN_LANES;
  a.tiles_y = ((dom.ny_block - 2) + 1 + (W - 3) - 1) / (W - 3);
  const int nt = a.tiles_x * a.tiles_y * a.nblocks;
  const dim3 g(8 * ((nt + 7) / 8));
  const bool last = (ksub + 1 == sc.ndte), damp = sc.evp_damping != 0;
  switch (W) {
    case 8: launch2_w<8>(a, last, damp, g, stream); break;
    case 12: launch2_w<12>(a, last, damp, g, stream); break;
    case 13: launch2_w<13>(a, last, damp, g, stream); break;
    case 14: launch2_w<14>(a, last, damp, g, stream); break;
    case 16: launch2_w<16>(a, last, damp, g, stream); break;
    default: throw Error{CICE_EINVAL, "fused_waves must be 8, 12, 13, 14 or 16"};
  }
  after_subcycle(ksub + 1);
}

// ---- K subcycles per sweep (k_subcycle_skew) ---------------------------------------------------------------------
// Where two subcycles per launch are possible, K are: the same condition (no ghost row of a local block changes
// during the launch) over K subcycles; and the metrics have to derive from HTN / HTE (the kernel has no other form).
// Small grids keep k_subcycle2 / the resident loop: a sweep needs rows to amortise its 2(K-1) + 2K - 1 extra steps.
bool Evp::can_skew() const {
  static const bool env_off = [] { const char* e = std::getenv("CICE4_AMD_SKEW"); return e && e[0] == '0'; }();
  if (!skew_on || env_off || !can_fuse() || !(derive_ok && derive_on)) return false;
  if (n * 8 * 14 >= (1ull << 32)) return false;   // the kernel reaches the 14 planes of the state by 32-bit offsets
  const long long cells = (long long)dom.nblocks() * (dom.nx_block - 2) * (dom.ny_block - 2);
  if (cells < skew_min_cells) return false;
  return skew_strips(skew_levels(), nullptr) > 0;   // (0: no column layout passes skew_layout_ok for this width)
}

int Evp::skew_levels() const { return skew_k_opt ? skew_k_opt : 4; }

// Wavefronts per level of the sweep kernel (its template parameter S): three for the default shape -- one 12-wavefront
// workgroup per CU whose strips lie side by side -- one for every other K (LDS) and where the option says so.
int Evp::skew_subs(int K) const {
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_SUBS"); return e ? std::atoi(e) : 0; }();
  const int want = env ? env : skew_subs_opt;
#ifdef CICE4_AMD_EXPERIMENTS
  return (K == 4 && skew_waves_per_simd(4) == 3 && want == 3) ? 3 : 1;
#else
  (void)want; (void)K;
  return 1;
#endif
}

// Is the column layout of the sweep kernel right for a ring of ncol + 1 positions (0 = ilo ... ncol - 1 = ihi, ncol = G)?
// Lane-level restatement of the kernel's dependencies: a level's stress at a lane needs the velocity of that lane and of
// its western neighbour (ihi, TWO lanes west, for the lane that holds ilo), G's velocity is the lane to its east, the
// momentum equation needs the stress of the lane to the east; a wavefront has no neighbour beyond its lanes 0 and 63,
// what it hands to the next level is what its owner lanes formed.  Every owned position must come out right after K levels.
static bool skew_layout_ok(int K, int S, int ncol, int shift, bool cyc) {
  const int txw = 62 * S + 2, ownw = txw - 2 * K, first = ownw - 1 - shift, npos = ncol + 1;
  if (first < 1) return false;
  const int nt = npos <= first ? 1 : 1 + (npos - first + ownw - 1) / ownw;
  std::vector<char> covered(npos, 0);
  for (int t = 0; t < nt; ++t) {
    const int own0 = t == 0 ? K + 1 + shift : K, start = t == 0 ? 0 : first + (t - 1) * ownw;
    auto pos_of = [&](int cw) { return start + cw - own0; };
    auto ring = [&](int p) { int r = p % npos; return r < 0 ? r + npos : r; };
    auto valid_col = [&](int cw) { const int p = pos_of(cw); return cyc || (p >= -1 && p <= ncol); };
    auto is_g = [&](int cw) { return cyc && ring(pos_of(cw)) == ncol; };
    auto is_ilo = [&](int cw) { return cyc && ring(pos_of(cw)) == 0; };
    // uo[cw]: the velocity of workgroup column cw handed to the current level is right (level 0: from memory)
    std::vector<char> uo(txw, 1), sv(txw, 0), un(txw, 0);
    for (int k = 0; k < K; ++k) {
      std::fill(sv.begin(), sv.end(), 0);
      std::fill(un.begin(), un.end(), 0);
      for (int sub = 0; sub < S; ++sub) {
        char st[64], mo[64];
        for (int l = 0; l < 64; ++l) {        // stress: own and western velocity (through the wave shift: inside the wavefront)
          const int cw = sub * 62 + l, w = is_ilo(cw) ? l - 2 : l - 1;
          bool self = uo[cw] != 0;
          if (k > 0 && is_g(cw)) self = l + 1 < 64 && uo[cw + 1];      // G mirrors the lane to its east (formed by the level before)
          st[l] = valid_col(cw) && w >= 0 && self && uo[sub * 62 + w];
        }
        for (int l = 0; l < 64; ++l) mo[l] = l + 1 < 64 && st[l] && st[l + 1];
        const int l0 = sub == 0 ? 0 : 1, l1 = sub == S - 1 ? 63 : 62;
        for (int l = l0; l <= l1; ++l) {      // owner lanes hand on
          sv[sub * 62 + l] = st[l];
          // (G's own "velocity" is never read: its reader takes the next lane's; a ghost column beyond an open edge keeps
          //  the velocity it has: the kernel hands it on unchanged)
          const int p = pos_of(sub * 62 + l);
          un[sub * 62 + l] = mo[l] || is_g(sub * 62 + l) || (!cyc && (p < 0 || p >= ncol));
        }
      }
      uo = un;
    }
    for (int cw = own0; cw <= txw - 1 - K; ++cw) {
      const int p = pos_of(cw);
      if (p < 0 || p > ncol) continue;
      // owned: the last level's stress (T-cell) and momentum (U-cell, not for G) at the owner lane
      if (!sv[cw] || (!(p == ncol) && !uo[cw])) return false;
      covered[p] = 1;
    }
  }
  for (int p = 0; p < npos; ++p)
    if (!covered[p]) return false;
  return true;
}

// host-only test hook (cice_debug_skew_layout): the layout rule without a device
bool evp_skew_layout_ok(int K, int S, int ncol, int shift, bool cyc) { return skew_layout_ok(K, S, ncol, shift, cyc); }

// Column strips of the sweep kernel (its geometry comment): strip 0 owns TXW - 2K - 1 - shift ring positions, every other
// strip TXW - 2K; positions 0 .. ncol (ncol = the ghost column G, whose T-cell has stresses of its own).  shift: the
// smallest number of lanes strip 0 has to give up for the layout to be right (skew_layout_ok): the seam of the ring
// (ihi, G, ilo) must not lie where a strip's east rim -- or, S > 1, the two columns neighbouring wavefronts share --
// would need one lane more.
int Evp::skew_strips(int K, int* shift_out) const {
  const int ncol = dom.nx_block - 2, S = skew_subs(K), txw = 62 * S + 2, ownw = txw - 2 * K;
  int& cached = strips_cache[K][S == 3];
  int& cshift = strips_cache_shift[K][S == 3];
  if (cached == 0) {
    int shift = 0;
    const bool cyc = dom.ew == BND_CYCLIC;
    while (shift < 2 * K + 4 && ncol >= K && !skew_layout_ok(K, S, ncol, shift, cyc)) ++shift;
    if (ncol < K) {
      shift = 0;                           // (blocks narrower than K columns: one strip, nothing to lay out)
    } else if (shift >= 2 * K + 4) {
      // no shift passes the checker: NOT a layout to sweep with (shift 0 is one it has just rejected).  Sentinel: can_skew() /
      // can_skew_fold() say no for this domain, it keeps k_subcycle2.
      cached = -1;
      cshift = 0;
      if (shift_out) *shift_out = 0;
      return 0;
    }
    const int f = ownw - 1 - shift, npos = ncol + 1;
    cached = npos <= f ? 1 : 1 + (npos - f + ownw - 1) / ownw;
    cshift = shift;
  }
  if (shift_out) *shift_out = cshift;
  return cached > 0 ? cached : 0;
}

// wavefronts per SIMD the kernel is built for (registers), and the workgroups per CU that follow from it and from
// the LDS a workgroup takes ((K-1) x 14 KB of 160 KB)
int Evp::skew_waves_per_simd(int K) const {
  if (K == 4 && skew_blocks_opt == 2) return 2;
  return K <= 6 ? 3 : 2;
}
int Evp::skew_blocks(int K) const {
  if (skew_subs(K) == 3) return 1;    // twelve wavefronts, 3 x 42 KB of LDS: the CU
  const int by_regs = skew_waves_per_simd(K) * 4 / K;
  const bool tp = SKEW_TPASS && K <= 4, early = SKEW_EARLY && !tp;
  const int by_lds = K > 1 ? (160 * 1024) / ((K - 1) * (tp ? 17920 : 14336) + (early ? 6144 : 0)) : 16;
  return std::max(1, std::min(by_regs, by_lds));
}

// rows a workgroup owns: as many workgroups as the chip holds at once (one round), segments not shorter than 4K rows
int Evp::skew_seg_rows(int K) const {
  const int rows = dom.ny_block - 2;
  if (skew_seg_opt) return std::min(skew_seg_opt, rows);
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const long long strips = (long long)skew_strips(K, nullptr) * dom.nblocks();
  long long nseg = (long long)ncu * skew_blocks(K) / std::max(1LL, strips);
  nseg = std::max(1LL, std::min(nseg, (long long)std::max(1, rows / (4 * K))));
  return (int)((rows + nseg - 1) / nseg);
}

// Rows per workgroup when the workgroups of a launch are not treated alike.  Workgroups are dispatched in blockIdx
// order: blockIdx & 7 is the XCD, and within an XCD the m-th workgroup goes to CU m mod 32 -- a CU holds workgroups m,
// m + 32, m + 64 of its XCD (measured: scripts/sweep_placement.py finds the triples b, b + 256, b + 512 on one CU);
// blockIdx -> tile is the XCD remap of the kernel.
// (1) skew_gen_pct: the SIMD issues from its oldest wavefront first, so of the workgroups sharing a CU the one dispatched
//     first finishes first: the "generation" g of a tile (0: first on its CU) gets weight 1 + (1 - g) * pct / 100.
// (2) skew_fill: a launch rarely has exactly as many workgroups as the chip has places -- 0.1 degree: 65 strips x 11
//     segments = 715 on 768, twelve segments would need 780 -- so some CUs hold one workgroup less than the others, and
//     their workgroups take shorter steps: T(n) ~ 2.0 + 1.1 n us for n workgroups on a CU.  A
//     workgroup gets rows in proportion to 1 / T(n of its CU): 26 % more on a CU that holds two instead of three.
// Both are normalised per column strip, so that the strip's segments still cover its rows exactly; any partition gives
// the same bits.
static constexpr int BAL_FIRST = 36, BAL_AGAIN = 8, BAL_RECOUNT = 6;   // sweeps measured after a new table / in a later tuning phase (balance_after_sweep)

int Evp::skew_fill_pct() const {   // option "skew_fill" / CICE4_AMD_SKEW_FILL: per cent more rows on a CU one workgroup short (0: off)
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_FILL"); return e ? std::atoi(e) : -1; }();
  return std::max(0, std::min(100, env >= 0 ? env : skew_fill));
}
bool Evp::skew_fill_on() const { return skew_fill_pct() > 0; }

// static weight (relative speed) of the workgroup that takes place `tile_lin` of a launch of nt workgroups: see build_skew_rows
double Evp::place_weight(int tile_lin, int nt, int gens, int per_xcd, bool fill) const {
  const int chunk = (nt + 7) >> 3, m = tile_lin % chunk;
  const int g = std::min(gens - 1, m / per_xcd);
  double w = 1.0 + (0.5 * (gens - 1) - g) * skew_gen_pct / 100.0;
  if (fill) {
    const int cnt = std::min(chunk, nt - (tile_lin / chunk) * chunk);    // workgroups of this tile's XCD that do anything
    int on_cu = 0;
    for (int q = m % per_xcd; q < cnt; q += per_xcd) ++on_cu;
    on_cu = std::min(on_cu, gens);                                       // (a second round: as if full)
    // (the workgroups of a CU that is not full run alike whatever their order: measured, scripts/sweep_placement.py)
    if (on_cu < gens) w = 1.0;
    // T(n) ~ a + b n with the ratio a / b that makes T(gens) / T(gens - 1) = 1 + pct / 100 (26 % <-> 2.0 + 1.1 n at gens = 3)
    const double f = skew_fill_pct() / 100.0, ab = 1.0 / f - (gens - 1);   // a / b
    w *= (ab + gens) / (ab + on_cu);
  }
  return w;
}

void Evp::build_skew_rows(int K, int tiles_x, int tiles_y, int nblocks, int seg_rows) {
  const int key[6] = {K, tiles_x, tiles_y, skew_gen_pct, seg_rows, skew_fill_pct()};
  if (skew_rows.n && !std::memcmp(key, skew_rows_key, sizeof(key))) return;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int per_blk = tiles_x * tiles_y, nt = per_blk * nblocks, per_xcd = std::max(1, ncu / 8);
  const int rows = dom.ny_block - 2, gens = std::max(1, skew_blocks(K));
  const bool fill = skew_fill_on() && nblocks == 1;
  std::vector<int32_t> tab((size_t)2 * per_blk);
  rows_w.assign((size_t)per_blk, 1.0);
  // one table for all blocks (they have the same shape): generation from the tile's place in block 0's dispatch order
  for (int tx = 0; tx < tiles_x; ++tx) {
    std::vector<double> w(tiles_y);
    double sum = 0;
    for (int ty = 0; ty < tiles_y; ++ty) {
      const int tile_lin = ty * tiles_x + tx;
      w[ty] = place_weight(tile_lin, nt, gens, per_xcd, fill);
      sum += w[ty];
      rows_w[(size_t)tile_lin] = w[ty];
    }
    double acc = 0;
    int prev = 0;
    for (int ty = 0; ty < tiles_y; ++ty) {
      acc += w[ty];
      int end = ty == tiles_y - 1 ? rows : (int)std::lround(rows * acc / sum);
      end = std::max(end, std::min(rows, prev + 1));       // at least one row each while rows last
      tab[2 * (size_t)(ty * tiles_x + tx)] = prev;
      tab[2 * (size_t)(ty * tiles_x + tx) + 1] = std::max(prev, end) - 1;   // empty segment: last < first
      prev = std::max(prev, end);
    }
  }
  skew_rows.alloc(tab.size());
  skew_rows.upload(tab.data(), stream);
  CICE_HIP(hipStreamSynchronize(stream));
  std::memcpy(skew_rows_key, key, sizeof(key));
  rows_host = tab;
  bal_tiles_x = tiles_x;
  bal_tiles_y = tiles_y;
  // the same as a LIST of tiles (SkewArgs::tiles), which is what the measured balancing works on: a strip may then hold
  // more tiles than another
  bal_tiles.clear();
  bal_got.clear();
  bal_nt = 0;
  if (balance_on() && nblocks == 1) {
    bal_strips = tiles_x;
    bal_slots = std::max(per_blk, ncu * gens);
    bal_gens = gens;
    bal_k = K;
    bal_per_xcd = per_xcd;
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int tx = 0; tx < tiles_x; ++tx)
        bal_tiles.push_back({tx, tab[2 * (size_t)(ty * tiles_x + tx)], tab[2 * (size_t)(ty * tiles_x + tx) + 1]});
    bal_list.alloc((size_t)4 * bal_slots);
    bal_upload(stream);
    bal_left = BAL_FIRST;      // a new table: measure it
    bal_recounted = false;
    bal_seen = 0;
  } else {
    bal_left = 0;
  }
  bal_since = 0;
}

void Evp::bal_upload(hipStream_t s) {
  bal_nt = (int)bal_tiles.size();
  std::vector<int32_t> h((size_t)4 * bal_nt);
  for (int p = 0; p < bal_nt; ++p) {
    h[4 * (size_t)p] = 0;
    h[4 * (size_t)p + 1] = bal_tiles[p].strip;
    h[4 * (size_t)p + 2] = bal_tiles[p].first;
    h[4 * (size_t)p + 3] = bal_tiles[p].last;
  }
  CICE_HIP(hipMemcpyAsync(bal_list.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
  CICE_HIP(hipStreamSynchronize(s));
  const bool fill = skew_fill_on();
  bal_w.resize((size_t)bal_nt);
  for (int p = 0; p < bal_nt; ++p) bal_w[p] = place_weight(p, bal_nt, bal_gens, bal_per_xcd, fill);
}

// ---- segments by MEASURED cost ---------------------------------------------------------------------------------------
// The static table above knows the chip; it does not know the rows.  A T-cell without ice costs a wavefront a few
// instructions, one with ice ~600: on a global grid most rows of a strip are open water and the strip's time is the time of
// its most ice-covered segment; and even on a fully covered grid the workgroups of a launch differ by more than the static
// weights say (seam strips, XCDs, the order on the CU: profiles/r04_sweep_wg_times.txt).  So the sweep kernel's start / end
// ticks per workgroup (SkewArgs::dbg, two clock reads per workgroup) are read back after a sweep and the strip's
// boundaries move: the cost of a row is taken as uniform within its old segment (duration x static weight / rows), the
// strip's total is dealt to its tiles in proportion to their static weights, the boundaries go half-way to where the
// running sum says (damping: a measurement has ~1.5 % of noise).  Any partition gives the same bits (the sweep tests run
// many), so this is tuning, not arithmetic.  When: the sweeps of the first loop after the table was built (eager, no
// graph, one synchronisation per sweep), and one loop in every `bal_every` after that (the ice edge moves).  One block
// per rank, the plain and the tripole-fold sweep (the lists of a wide-halo slab are not balanced).
// Option "skew_balance" / CICE4_AMD_SKEW_BALANCE=0|1.
bool Evp::balance_on() const {
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_BALANCE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  return (env >= 0 ? env == 1 : skew_balance != 0) && dom.nblocks() == 1;
}

// One strip's step of the measured balancing, as a pure function (Evp::balance_after_sweep; cice_debug_balance_strip for the
// CPU tests): n tiles with exclusive ends e[] (bottom to top), measured durations d[] and static weights w[] of their places,
// act[rows] != 0 where a row holds ice (or NULL: every row counts).  cost[rows] <- what a row costs; returns the strip's
// total (0: nothing to go by) and, if ne, the new ends: every boundary half-way to where the running sum of the cost reaches
// the tiles' shares.
double balance_strip(int rows, int n, const int* e, const double* d, const double* w, const unsigned char* act, double* cost,
                     int* ne) {
  double omega = 0, wsum = 0;
  for (int i = 0; i < n; ++i) {
    if (!(d[i] >= 0) || d[i] > 1e9) return 0.0;          // (0 is fine: a workgroup with no row that holds ice)
    wsum += w[i];
  }
  // a segment's cost lies on its rows that hold ice (a segment of open water: what it took is not the rows' cost)
  {
    int lo = 0;
    for (int i = 0; i < n; ++i) {
      int nact = 0;
      for (int r = lo; r < e[i]; ++r) nact += act ? (act[r] != 0) : 1;
      const double per_row = nact ? d[i] * w[i] / nact : 0.0;
      for (int r = lo; r < e[i]; ++r) cost[r] = (act ? act[r] != 0 : true) ? per_row : 0.0;
      if (nact) omega += d[i] * w[i];
      lo = std::max(lo, e[i]);
    }
    for (int r = lo; r < rows; ++r) cost[r] = 0.0;
  }
  if (!(omega > 0) || !ne) return omega > 0 ? omega : 0.0;
  const int minrows = rows >= 2 * n ? 2 : (rows >= n ? 1 : 0);
  double acc = 0, cum = 0;
  int r = 0, prev = 0;
  for (int i = 0; i < n; ++i) {
    acc += w[i];
    const double want = omega * acc / wsum;
    while (r < rows && cum + cost[r] < want) cum += cost[r++];
    const double x = r < rows && cost[r] > 0 ? r + (want - cum) / cost[r] : r;
    int end = i == n - 1 ? rows : (int)std::lround(e[i] + 0.5 * (x - e[i]));
    end = std::max(end, std::min(rows, prev + minrows));
    end = std::min(end, rows - minrows * (n - 1 - i));
    end = std::max(end, prev);
    ne[i] = end;
    prev = end;
  }
  return omega;
}

bool Evp::places_on() const {   // CICE4_AMD_SKEW_PLACES=0: off (A/B)
  static const bool env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_PLACES"); return !(e && e[0] == '0'); }();
  return env;
}

// The tiles of the list change places inside every XCD's part of it so that strip sx ends up with the share target[sx] /
// sum(target) of all places' static weights (balance_after_sweep): heaviest place first, to the tile whose strip lacks most
// weight per tile it has still to place.  bal_got <- the weight every strip got.  (Dealt ONCE per table: a second deal ten
// sweeps later, from the times the strips were then measured to take, narrows the strips' means from 3.2 to 2.0 % and makes
// the launch no shorter -- 255.4 against 254.6 us per subcycle, it disturbs the cuts inside the strips: profiles/r05_sweep_places_by_strip.txt)
void Evp::deal_places(const std::vector<double>& target) {
  const int ns = bal_strips, n2 = (int)bal_tiles.size(), chunk2 = (n2 + 7) >> 3;
  std::vector<double> pw((size_t)n2), wstar(ns, 0.0);
  for (int q = 0; q < n2; ++q) pw[(size_t)q] = place_weight(q, n2, bal_gens, bal_per_xcd, skew_fill_on());
  std::vector<int> cnt(ns, 0), placed(ns, 0);
  for (const BalTile& bt : bal_tiles) ++cnt[bt.strip];
  double tsum = 0, wtot = 0;
  for (int sx = 0; sx < ns; ++sx)
    if (cnt[sx]) tsum += target[sx];
  for (double v : pw) wtot += v;
  if (!(tsum > 0)) return;
  for (int sx = 0; sx < ns; ++sx)
    if (cnt[sx]) wstar[sx] = target[sx] / tsum * wtot;
  bal_got.assign(ns, 0.0);
  for (int c0 = 0; c0 < n2; c0 += chunk2) {
    const int c1 = std::min(n2, c0 + chunk2);
    std::vector<int> places;
    const std::vector<BalTile> tiles(bal_tiles.begin() + c0, bal_tiles.begin() + c1);
    for (int q = c0; q < c1; ++q) places.push_back(q);
    std::stable_sort(places.begin(), places.end(), [&](int a, int b) { return pw[(size_t)a] > pw[(size_t)b]; });
    std::vector<char> used(tiles.size(), 0);
    for (int q : places) {
      int best = -1;
      double need = -1e300;
      for (size_t i = 0; i < tiles.size(); ++i) {
        if (used[i]) continue;
        const int sx = tiles[i].strip;
        const double lack = (wstar[sx] - bal_got[sx]) / std::max(1, cnt[sx] - placed[sx]);
        if (lack > need) { need = lack; best = (int)i; }
      }
      if (best < 0) break;
      used[(size_t)best] = 1;
      bal_tiles[(size_t)q] = tiles[(size_t)best];
      bal_got[tiles[(size_t)best].strip] += pw[(size_t)q];
      ++placed[tiles[(size_t)best].strip];
    }
  }
}

void Evp::balance_after_sweep(hipStream_t s) {
  const int nt = bal_nt, chunk = (nt + 7) >> 3, ns = bal_strips;
  const size_t g = 8 * (size_t)((nt + 7) / 8);
  CICE_HIP(hipStreamSynchronize(s));
  std::vector<long long> t(2 * g);
  CICE_HIP(hipMemcpy(t.data(), skew_dbg.p, t.size() * 8, hipMemcpyDeviceToHost));
  const int rows = dom.ny_block - 2;
  // the rows that hold ice, per strip (this step's masks: fetched once per loop that is measured)
  const bool have_act = rowact_on() && rowact_strips == ns && rowact.n >= (size_t)ns * rows;
  if (have_act && rowact_host_stale) {
    rowact_host.resize((size_t)ns * rows);
    CICE_HIP(hipMemcpy(rowact_host.data(), rowact.p, rowact_host.size(), hipMemcpyDeviceToHost));
    rowact_host_stale = false;
  }
  // the tiles of every strip, bottom to top
  std::vector<std::vector<int>> of(ns);
  for (int p = 0; p < nt; ++p) of[bal_tiles[p].strip].push_back(p);
  for (auto& v : of) std::sort(v.begin(), v.end(), [&](int a, int b) { return bal_tiles[a].first < bal_tiles[b].first; });
  ++bal_seen;
  // Once per table, after a few measured sweeps: the places the launch leaves empty (715 tiles on 768 at 0.1 degree) go, one
  // by one, to the strip whose tiles take longest -- the two strips at the seam of the ring do 11 % more per row, and a strip
  // under more ice than another more still.  Greedy on (strip's cost) / (its tiles) minimises the maximum.
  const bool recount = !bal_recounted && bal_seen >= BAL_RECOUNT && nt < bal_slots;
  std::vector<int> want_cnt(ns);
  std::vector<double> strip_cost(ns, 0.0);
  std::vector<double> cost((size_t)std::max(rows, 1));
  std::vector<std::vector<double>> rowcost(recount ? ns : 0);
  bool changed = false;
  auto duration = [&](int p) {
    const size_t b = ((size_t)(p % chunk) << 3) | (size_t)(p / chunk);      // the kernel's XCD remap, inverted
    return (double)(t[2 * b + 1] - t[2 * b]);
  };
  for (int tx = 0; tx < ns; ++tx) {
    const std::vector<int>& tl = of[tx];
    const int ty_n = (int)tl.size();
    want_cnt[tx] = ty_n;
    if (!ty_n) continue;
    std::vector<double> d(ty_n), w(ty_n);
    std::vector<int> e(ty_n), ne(ty_n);
    for (int i = 0; i < ty_n; ++i) {
      d[i] = duration(tl[i]);
      w[i] = bal_w[tl[i]];
      e[i] = bal_tiles[tl[i]].last + 1;                 // exclusive end (an empty tile: = its first row)
    }
    const unsigned char* act = have_act ? rowact_host.data() + (size_t)tx * rows : nullptr;
    const double omega = balance_strip(rows, ty_n, e.data(), d.data(), w.data(), act, cost.data(), recount ? nullptr : ne.data());
    if (!(omega > 0)) continue;
    strip_cost[tx] = omega;
    if (recount) {
      rowcost[tx].assign(cost.begin(), cost.begin() + rows);
      continue;                                 // (the table is re-cut below, with the new numbers of tiles)
    }
    int prev = 0;
    for (int i = 0; i < ty_n; ++i) {
      BalTile& bt = bal_tiles[tl[i]];
      if (bt.first != prev || bt.last != ne[i] - 1) changed = true;
      bt.first = prev;
      bt.last = ne[i] - 1;
      prev = ne[i];
    }
  }
  if (recount) {
    bal_recounted = true;
    int spare = bal_slots - nt;
    // (costs in steps of 4 % of the largest, ties to the lower strip: strips that cost about the same get their extra tile
    //  as ONE run of neighbours -- tiles of neighbouring strips that cover the same rows at the same time share the columns
    //  they overlap in through the L2; scattered, the launch reads 8 % more: profiles/r04_sweep_reads_by_table.txt)
    {
      double top = 0;
      for (double v : strip_cost) top = std::max(top, v);
      if (top > 0)
        for (double& v : strip_cost)
          if (v > 0) v = std::max(1.0, std::floor(v / (0.04 * top))) * (0.04 * top);
    }
    for (; spare > 0; --spare) {
      int best = -1;
      double worst = 0;
      for (int tx = 0; tx < ns; ++tx)
        if (strip_cost[tx] > 0 && want_cnt[tx] < std::max((int)of[tx].size(), rows / (4 * bal_k)) &&   // (segments not shorter than 4K rows: skew_seg_rows)
            strip_cost[tx] / want_cnt[tx] > worst) {
          worst = strip_cost[tx] / want_cnt[tx];
          best = tx;
        }
      if (best < 0) break;
      ++want_cnt[best];
    }
    for (int tx = 0; tx < ns; ++tx) {
      const int have = (int)of[tx].size(), n = want_cnt[tx];
      if (n == have || rowcost[tx].empty()) continue;
      // the strip's rows dealt to n tiles of equal cost (new tiles take the places at the end of the list: weight as found)
      std::vector<int> tl = of[tx];
      for (int i = have; i < n; ++i) {
        tl.push_back((int)bal_tiles.size());
        bal_tiles.push_back({tx, 0, -1});
      }
      const std::vector<double>& rc = rowcost[tx];
      double total = 0;
      for (double v : rc) total += v;
      double cum = 0;
      int r = 0, prev = 0;
      for (int i = 0; i < n; ++i) {
        const double want = total * (i + 1) / n;
        while (r < rows && cum + rc[r] <= want) cum += rc[r++];
        int end = i == n - 1 ? rows : std::max(r, std::min(rows, prev + 1));
        end = std::min(end, rows - (n - 1 - i));
        end = std::max(end, prev);
        bal_tiles[tl[i]].first = prev;
        bal_tiles[tl[i]].last = end - 1;
        prev = end;
      }
      changed = true;
    }
    // WHICH place a tile takes.  A strip's time is its cost over the sum of its places' weights, and the strips that kept
    // one tile less (0.1 degree: 12 of 65) have 218-row segments where the others have 200: they get the places that are
    // dispatched first on their CUs (+15 %), the others more of the late ones.  Tiles change places only INSIDE an XCD's part
    // of the list (tiles of neighbouring strips over the same rows stay on one L2), heaviest place to the tile whose strip
    // is shortest of weight per unit of cost.  The re-cuts that follow work with the weights of the new places.
    if (places_on() && (int)bal_tiles.size() > ns) {
      std::vector<double> target(ns, 0.0);
      double cmean = 0;
      int nc2 = 0;
      for (int sx = 0; sx < ns; ++sx)
        if (strip_cost[sx] > 0) { cmean += strip_cost[sx]; ++nc2; }
      cmean = nc2 ? cmean / nc2 : 1.0;
      for (int sx = 0; sx < ns; ++sx) target[sx] = strip_cost[sx] > 0 ? strip_cost[sx] : cmean;
      deal_places(target);
      changed = true;
    }
  }
  if (changed) bal_upload(s);
  ++bal_sweeps;
  if (bal_left > 0) --bal_left;
}

#ifdef CICE4_AMD_EXPERIMENTS
template <bool PAIRS>
static void launch_skew_s3(const SkewArgs& sa, bool last, bool damp, dim3 g, hipStream_t s) {
  const dim3 blk(64 * 4 * 3);
  if (last) {
    if (damp) hipLaunchKernelGGL((k_subcycle_skew<4, true, true, 3, PAIRS, 3>), g, blk, 0, s, sa);
    else hipLaunchKernelGGL((k_subcycle_skew<4, true, false, 3, PAIRS, 3>), g, blk, 0, s, sa);
  } else {
    if (damp) hipLaunchKernelGGL((k_subcycle_skew<4, false, true, 3, PAIRS, 3>), g, blk, 0, s, sa);
    else hipLaunchKernelGGL((k_subcycle_skew<4, false, false, 3, PAIRS, 3>), g, blk, 0, s, sa);
  }
}
#endif

template <int K, int WS>
static void launch_skew_kb(const SkewArgs& sa, bool last, bool damp, dim3 g, hipStream_t s, bool pairs = false, int subs = 1) {
  const dim3 blk(64 * K);
  if constexpr (K == 4 && WS == 3) {
#ifdef CICE4_AMD_EXPERIMENTS
    if (subs == 3) {   // twelve wavefronts per workgroup: three per level, side by side
      if (pairs) launch_skew_s3<true>(sa, last, damp, g, s);
      else launch_skew_s3<false>(sa, last, damp, g, s);
      return;
    }
#endif
    (void)subs;
    if (pairs) {   // (the pair layout is built for the default shape only)
      if (last) {
        if (damp) hipLaunchKernelGGL((k_subcycle_skew<K, true, true, WS, true>), g, blk, 0, s, sa);
        else hipLaunchKernelGGL((k_subcycle_skew<K, true, false, WS, true>), g, blk, 0, s, sa);
      } else {
        if (damp) hipLaunchKernelGGL((k_subcycle_skew<K, false, true, WS, true>), g, blk, 0, s, sa);
        else hipLaunchKernelGGL((k_subcycle_skew<K, false, false, WS, true>), g, blk, 0, s, sa);
      }
      return;
    }
  }
  if (last) {
    if (damp) hipLaunchKernelGGL((k_subcycle_skew<K, true, true, WS>), g, blk, 0, s, sa);
    else hipLaunchKernelGGL((k_subcycle_skew<K, true, false, WS>), g, blk, 0, s, sa);
  } else {
    if (damp) hipLaunchKernelGGL((k_subcycle_skew<K, false, true, WS>), g, blk, 0, s, sa);
    else hipLaunchKernelGGL((k_subcycle_skew<K, false, false, WS>), g, blk, 0, s, sa);
  }
}

// The sweep's pair layout (k_subcycle_skew<.., PAIRS>): used on domains whose sweeps are the default shape (K = 4), whose
// east-west ghost copies are the plain wrap, and that have no refresh, no message and no fold between two sweeps (one rank,
// one full-width block: the 0.1-degree configuration on one GPU) -- the halo lists address planes.
bool Evp::pairs_ok() const {
  static const bool env_off = [] { const char* e = std::getenv("CICE4_AMD_SKEW_PAIRS"); return e && e[0] == '0'; }();
  if (!pairs_on || env_off || !SKEW_WIDE) return false;
  return can_skew() && skew_levels() == 4 && skew_waves_per_simd(4) == 3 && fwd_is_ew_wrap && !halo.has_refresh() &&
         !halo.has_fold() && dom.overlap == 0;
}
// the state moves into the pair copies (both: cells a sweep never writes have to hold the same value in either copy) ...
void Evp::to_pairs() {
  if (in_pairs) return;
  const dim3 g((unsigned)((n + 255) / 256));
  if (flips == 0 && copies_identical) {
    hipLaunchKernelGGL(k_to_pairs, g, dim3(256), 0, stream, n, (const double*)st[cur].p, st2[cur].p, st2[1 - cur].p);
  } else {
    hipLaunchKernelGGL(k_to_pairs, g, dim3(256), 0, stream, n, (const double*)st[cur].p, st2[cur].p, (double*)nullptr);
    hipLaunchKernelGGL(k_to_pairs, g, dim3(256), 0, stream, n, (const double*)st[1 - cur].p, st2[1 - cur].p, (double*)nullptr);
  }
  in_pairs = true;
}
// ... and back (the current copy; the other one keeps what it held, as after any subcycle)
void Evp::to_planes() {
  if (!in_pairs) return;
  hipLaunchKernelGGL(k_from_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, (const double*)st2[cur].p, st[cur].p);
  in_pairs = false;
}

// the arguments every launch of the sweep kernel shares
void Evp::skew_args(SkewArgs& sa, int K) {
  sa.a = make_args();
  sa.seg_rows = skew_seg_rows(K);
  sa.a.tiles_x = skew_strips(K, &sa.own_shift);
  sa.a.tiles_y = ((dom.ny_block - 2) + sa.seg_rows - 1) / sa.seg_rows;
  sa.prio_rotate = skew_prio;
  {
    static const int deal = [] { const char* e = std::getenv("CICE4_AMD_SKEW_DEAL"); return e ? std::atoi(e) : 0; }();
    sa.level_deal = deal;
  }
  sa.rows = nullptr;
  sa.fwd_rule = fwd_is_ew_wrap ? 1 : 0;
  sa.dbg = nullptr;
  sa.stamps = nullptr;
  sa.phases = nullptr;
  sa.stagger_ticks = skew_stagger_ns / 10;
  sa.stagger_mod = std::max(1, skew_blocks(K));
  sa.st_in = in_pairs ? st2[cur].p : st[cur].p;
  sa.st_out = st[1 - cur].p;       // (skew_launch: the pair copy unless the sweep is the last of evp(dt))
  sa.uar = uarena.p;
  sa.uar4 = uar4.p;
  sa.hnhe = hnhe.p;
  sa.msk = skew_msk.p;
  sa.tiles = nullptr;
  sa.tile_first = sa.tile_count = 0;
  sa.rowact = rowact_on() && rowact_strips == sa.a.tiles_x && rowact_k == K ? rowact.p : nullptr;
  sa.run_next = run_next.p;
  sa.run_end = run_end.p;
}

void Evp::skew_launch(const SkewArgs& sa0, int K, bool last, int nt, hipStream_t s) {
  const dim3 g(8 * ((nt + 7) / 8));
  const bool damp = sc.evp_damping != 0;
  const int WS = skew_waves_per_simd(K);
  SkewArgs sa = sa0;
  if (in_pairs && !last) sa.st_out = st2[1 - cur].p;
  switch (K * 10 + WS) {
    case 23: launch_skew_kb<2, 3>(sa, last, damp, g, s); break;
    case 33: launch_skew_kb<3, 3>(sa, last, damp, g, s); break;
    case 43: launch_skew_kb<4, 3>(sa, last, damp, g, s, in_pairs, skew_subs(4)); break;
#ifdef CICE4_AMD_EXPERIMENTS
    case 42: launch_skew_kb<4, 2>(sa, last, damp, g, s); break;
    case 53: launch_skew_kb<5, 3>(sa, last, damp, g, s); break;
    case 63: launch_skew_kb<6, 3>(sa, last, damp, g, s); break;
    case 82: launch_skew_kb<8, 2>(sa, last, damp, g, s); break;
#endif
    default: throw Error{CICE_EINVAL, "unsupported (skew_levels, skew_blocks) combination"};
  }
}

// subcycles ksub .. ksub+K-1
void Evp::launch_subcycle_skew(int ksub, int K, bool flip_and_halo, hipStream_t on) {
  ++loop_launches;
  SkewArgs sa{};
  skew_args(sa, K);
  if (skew_rows_on()) {   // (built by subcycles() before any capture: it uploads a table)
    build_skew_rows(K, sa.a.tiles_x, sa.a.tiles_y, sa.a.nblocks, sa.seg_rows);
    sa.rows = skew_rows.p;
  }
  int nt = sa.a.tiles_x * sa.a.tiles_y * sa.a.nblocks;
  const bool listed = sa.rows && balance_on() && bal_nt > 0 && bal_strips == sa.a.tiles_x;
  if (listed) {             // the measured table: a list of tiles (a strip may hold more of them than another)
    sa.rows = nullptr;
    sa.tiles = bal_list.p;
    sa.tile_first = 0;
    sa.tile_count = nt = bal_nt;
  }
  if (skew_debug) {
    const size_t want = 2 * (size_t)(8 * ((nt + 7) / 8));
    if (skew_dbg.n < want) skew_dbg.alloc(want);
    sa.dbg = skew_dbg.p;
  }
  {
    const size_t g = 8 * ((size_t)(nt + 7) / 8);
    sa.stamps = stamp_buffer((1 + 2 * (size_t)K) * g);      // [4 g] stamps, then [8 K g] phase sums per level
    sa.phases = sa.stamps ? sa.stamps + 4 * g : nullptr;
  }
  const bool measure = bal_left > 0 && !in_capture && listed && skew_dbg.n >= 2 * (size_t)(8 * ((nt + 7) / 8));
  if (measure) sa.dbg = skew_dbg.p;
  skew_launch(sa, K, ksub + K - 1 == sc.ndte, nt, on ? on : stream);
  if (measure) balance_after_sweep(on ? on : stream);
  if (in_pairs && ksub + K - 1 == sc.ndte) in_pairs = false;   // the last sweep of evp(dt) stores planes
  if (flip_and_halo) after_subcycle(ksub + K - 1);
}

// ---- the sweep in front of a wide-halo refresh: the rows the neighbours wait for FIRST ------------------------------
// A slab of a domain cut across ranks (cice_domain_create_slabs, overlap H) refreshes its H + 1 outer rows from their
// owners every H subcycles: pack -> one message per neighbour -> unpack, ~3.6 MB per neighbour at 0.1 degree / 8 ranks
// (DESIGN.md section 7).  Run after the sweep it costs its full time; but only the H + 1 owned rows at either end of
// the slab are sent, and the rows that are received are not read before the NEXT sweep.  So the last sweep before a
// refresh runs as TWO launches over an explicit tile list (SkewArgs::tiles): the edge segments -- owned rows
// own_jlo .. own_jlo + H and own_jhi - H .. own_jhi, where there is a neighbour -- on the main stream, followed there by
// the refresh; the interior segments on a second stream, beside both.  The extension rows themselves are not computed
// in this sweep at all: the refresh overwrites them (all 14 planes, whole rows).  Edge segments are short (H + 1 rows
// against ~30), so the edge launch ends at about half the interior's time and the exchange has the other half.
// Same arithmetic on every owned row as the one-launch sweep: a segment's rows depend on the rows of the state the sweep
// starts from, not on which launch computes its neighbours.
// Cost, measured on one GPU at the 8-rank geometry of the 0.1-degree grid (profiles/r04_split_probe.txt): the sweep in
// front of the refresh takes 48.4 us per subcycle in this form against 39.7 as one launch over the same 300 rows (short
// edge segments, a fork and a join between two streams) = +35 us per refresh; it pays where the exchange it hides costs
// more than that.  Nothing has run between two devices, so it is OFF by default and bench.py --gpus N decides by timing.
bool Evp::can_split() const {
  // off unless asked for: option "skew_split" or CICE4_AMD_SKEW_SPLIT=1 (bench.py --gpus N times both forms and keeps the faster)
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_SKEW_SPLIT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  if (env == 0 || (!split_on && env != 1) || in_capture) return false;
  return can_trim();
}

// sweeps over tile lists (extension rows trimmed to what the next sweeps need): wide-halo slabs with neighbours
bool Evp::can_trim() const {
  static const bool env_off = [] { const char* e = std::getenv("CICE4_AMD_SKEW_TRIM_EXT"); return e && e[0] == '0'; }();
  if (!trim_ext_on || env_off) return false;
  if (dom.overlap <= 0 || !halo.multi_rank() || halo.has_fold() || dom.nblocks() < 1) return false;
  const int K = skew_levels(), H = dom.overlap;
  for (int gid : dom.local) {
    const Block& b = dom.all[gid];
    if (b.own_jhi - b.own_jlo + 1 < 2 * (H + 1) + 4 * K) return false;   // edges and at least one interior segment of 4K rows
  }
  return true;
}

// Tile list of a sweep on a wide-halo slab domain that computes the owned rows and `ext` extension rows beyond them at
// either end (where there is a neighbour): after a sweep only as many extension rows have to be right as there are
// subcycles left until the next refresh -- none in front of the refresh itself.  split: the segments the neighbours wait
// for (the first / last H + 1 owned rows) come first in the list (edge tiles), the rest behind them (ext = 0 only).
const Evp::TileTab& Evp::tiles_for(int K, int ext, bool split) {
  // (split_probe: a measurement aid -- a one-block domain WITHOUT neighbours is cut as if it were an interior slab with
  //  split_probe overlap rows at either end: the kernel cost of the edge + interior form at the deployment's geometry)
  const int H = split_probe > 0 ? split_probe : dom.overlap, S = skew_strips(K, nullptr), nb = dom.nblocks();
  for (const auto& t : tile_tabs)
    if (t->K == K && t->ext == ext && t->split == split && t->S == S && t->nb == nb) return *t;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  std::vector<int32_t> edge, inner;
  std::vector<std::array<int, 4>> seg_tmp;
  int nedge_seg = 0;
  for (int l = 0; l < nb; ++l) {
    Block b = dom.all[dom.local[l]];
    if (split_probe > 0) {
      b.own_jlo = b.jlo + split_probe;
      b.own_jhi = b.jhi - split_probe;
    }
    int lo = std::max(b.jlo, b.own_jlo - ext), hi = std::min(b.jhi, b.own_jhi + ext);   // rows this sweep computes
    if (split && b.own_jlo > b.jlo) {        // a neighbour to the south: its refresh reads our first H + 1 owned rows
      for (int t = 0; t < S; ++t) edge.insert(edge.end(), {l, t, b.own_jlo - b.jlo, b.own_jlo + H - b.jlo});
      lo = b.own_jlo + H + 1;
      ++nedge_seg;
    }
    if (split && b.own_jhi < b.jhi) {
      for (int t = 0; t < S; ++t) edge.insert(edge.end(), {l, t, b.own_jhi - H - b.jlo, b.own_jhi - b.jlo});
      hi = b.own_jhi - H - 1;
      ++nedge_seg;
    }
    inner.insert(inner.end(), {l, lo, hi, 0});
  }
  // as many workgroups as the chip holds at once, like the one-launch sweep (skew_seg_rows)
  const long long slots = (long long)ncu * skew_blocks(K);
  std::vector<int32_t> tab = edge;
  for (size_t e = 0; e + 3 < inner.size(); e += 4) {
    const int l = inner[e], lo = inner[e + 1], hi = inner[e + 2], rows = hi - lo + 1;
    long long nseg = (slots - (long long)nedge_seg * S) / std::max(1LL, (long long)S * nb);
    nseg = std::max(1LL, std::min(nseg, (long long)std::max(1, rows / (4 * K))));
    const Block& b = dom.all[dom.local[l]];   // (jlo is what the table is relative to: the same with or without the probe)
    // Round 5: the segments of a strip are cut by the STATIC weights of the places their workgroups take on the chip
    // (place_weight: dispatch order on the CU, CUs that hold one workgroup less), as build_skew_rows cuts the one-block
    // sweep -- one block per rank, lists without edge tiles in front (their places are taken by the edge launch).  Any
    // partition of a strip's rows gives the same bits.
    const bool weighted = nb == 1 && edge.empty() && (skew_gen_pct > 0 || skew_fill_on()) && skew_seg_opt == 0;
    const int nt_all = (int)nseg * S, gens = std::max(1, skew_blocks(K)), per_xcd = std::max(1, ncu / 8);
    for (int t = 0; t < S; ++t) {
      std::vector<double> w((size_t)nseg, 1.0);
      double sum = 0;
      for (int g = 0; g < (int)nseg; ++g) {
        if (weighted) w[(size_t)g] = place_weight(g * S + t, nt_all, gens, per_xcd, skew_fill_on());
        sum += w[(size_t)g];
      }
      double acc = 0;
      int prev = 0;
      for (int g = 0; g < (int)nseg; ++g) {
        acc += w[(size_t)g];
        int end = g == (int)nseg - 1 ? rows : (int)std::lround(rows * acc / sum);
        end = std::max(end, std::min(rows, prev + 1));
        // (sorted into place g * S + t of the block's part of the list below: segment-major, as the equal cut was)
        seg_tmp.push_back({g, t, lo + prev - b.jlo, lo + std::max(prev, end) - 1 - b.jlo});
        prev = std::max(prev, end);
      }
    }
    std::sort(seg_tmp.begin(), seg_tmp.end(), [](const std::array<int, 4>& x, const std::array<int, 4>& y) {
      return x[0] != y[0] ? x[0] < y[0] : x[1] < y[1];
    });
    for (const auto& e4 : seg_tmp) tab.insert(tab.end(), {l, e4[1], e4[2], e4[3]});
    seg_tmp.clear();
  }
  std::unique_ptr<TileTab> t(new TileTab);
  t->K = K; t->ext = ext; t->split = split; t->S = S; t->nb = nb;
  t->edge = (int)(edge.size() / 4);
  t->total = (int)(tab.size() / 4);
  t->tab.alloc(tab.size());
  t->tab.upload(tab.data(), stream);
  CICE_HIP(hipStreamSynchronize(stream));
  tile_tabs.push_back(std::move(t));
  return *tile_tabs.back();
}

// every table a range of subcycles will use, and the second stream: allocations and uploads stay outside captures
void Evp::build_split(int K) {
  const int H = dom.overlap;
  for (int ext = 0; ext <= H; ++ext) (void)tiles_for(K, ext, false);   // (a few KB each; any value can occur at the end of a step)
  if (can_split()) {
    (void)tiles_for(K, 0, true);
    if (!stream2) CICE_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    if (!ev_fork) CICE_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    if (!ev_join) CICE_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
}

// a sweep that leaves `ext` extension rows right (one launch over a tile list)
void Evp::launch_subcycle_skew_ext(int ksub, int K, int ext) {
  ++loop_launches;
  const TileTab& t = tiles_for(K, ext, false);
  SkewArgs sa{};
  skew_args(sa, K);
  sa.tiles = t.tab.p;
  sa.tile_first = 0;
  sa.tile_count = t.total;
  skew_launch(sa, K, ksub + K - 1 == sc.ndte, t.total, stream);
  after_subcycle(ksub + K - 1);
}

void Evp::launch_subcycle_skew_split(int ksub, int K) {
  loop_launches += 2;
  const TileTab& t = tiles_for(K, 0, true);
  const int split_edge = t.edge, split_total = t.total;
  SkewArgs sa{};
  skew_args(sa, K);
  sa.tiles = t.tab.p;
  const bool last = ksub + K - 1 == sc.ndte;
  // the interior beside everything that follows on the main stream
  CICE_HIP(hipEventRecord(ev_fork, stream));
  CICE_HIP(hipStreamWaitEvent(stream2, ev_fork, 0));
  sa.tile_first = 0;
  sa.tile_count = split_edge;
  if (split_edge) skew_launch(sa, K, last, split_edge, stream);
  sa.tile_first = split_edge;
  sa.tile_count = split_total - split_edge;
  skew_launch(sa, K, last, sa.tile_count, stream2);
  CICE_HIP(hipEventRecord(ev_join, stream2));
  after_subcycle(ksub + K - 1);          // flips the copies; the refresh follows the edge launch on the main stream
  CICE_HIP(hipStreamWaitEvent(stream, ev_join, 0));
}

// ---- K subcycles per sweep on a grid with a tripole fold ----------------------------------------------------------
// The fold couples the two halves of the top row after EVERY subcycle; a sweep has no subcycle finished anywhere before
// its end.  But the north boundary only reaches K rows down in K subcycles.  So: the sweep runs as on an open boundary
// (rows above jhi - K come out wrong), and beside it a BAND of the top 2K + 1 rows runs the K subcycles one at a time,
// k_subcycle + the halo update with its fold, on buffers of its own.  The band's lower rows see a stale row below them
// and go wrong from the other side, one row per subcycle -- after K subcycles its rows above jhi - K are right.  They
// replace the sweep's.  Per sweep: 2 + 2K small launches beside one large one -- BESIDE in time as well: the sweep goes to
// a second stream (a parallel branch of the captured graph) AFTER the band's first copy, so that its workgroups are placed
// first; the band's launches stay on the main stream and run where and when the sweep leaves room -- in practice in its tail,
// as its workgroups finish (letting the band start ahead of the sweep displaces sweep workgroups into a second round: +29 %,
// DESIGN.md section 3.2); only the last copy waits for the sweep.
// CICE4_AMD_SKEW_FOLD_BESIDE=0: one after the other on one stream (round 3's form), for A/B.
__global__ __launch_bounds__(256) void k_band_rows(double* __restrict__ dst, double* __restrict__ dst2,
                                                   const double* __restrict__ src, size_t n, size_t off, size_t len) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // 14 planes x len cells starting at `off` in each
  if (t >= 14 * len) return;
  const size_t p = t / len, q = off + (t - p * len);
  const double v = src[p * n + q];
  dst[p * n + q] = v;
  if (dst2) dst2[p * n + q] = v;
}

bool Evp::can_skew_fold() const {
  static const bool env_off = [] { const char* e = std::getenv("CICE4_AMD_SKEW"); return e && e[0] == '0'; }();
  if (!skew_on || !skew_fold_on || env_off || !fuse_on || !halo.fwd_ok() || !(derive_ok && derive_on)) return false;
  if (!halo.has_fold() || dom.nblocks() != 1) return false;
  // one block of the whole grid, or the top slab of a wide-halo domain (its overlap rows are refreshed between sweeps)
  // (launch_range never lets a launch straddle a refresh: where K subcycles do not fit before the next one, the rest
  //  runs one by one)
  if (dom.overlap == 0 && (halo.multi_rank() || halo.has_refresh())) return false;
  if (n * 8 * 14 >= (1ull << 32)) return false;
  const Block& bl = dom.all[dom.local[0]];
  if (bl.jhi - bl.jlo + 1 < 4 * skew_levels() + 4) return false;
  // (the band's ten small launches per sweep cost more on a small grid: 1000 x 800 50 us per subcycle against 54 through
  //  one launch per subcycle, 720 x 600 38 against 34)
  const long long cells = (long long)(dom.nx_block - 2) * (dom.ny_block - 2);
  if (cells < std::max(skew_min_cells, skew_min_cells ? 800000LL : 0LL)) return false;
  return skew_strips(skew_levels(), nullptr) > 0;
}

void Evp::ensure_band(int K) {
  const Block& bl = dom.all[dom.local[0]];
  for (int k = 0; k < 2; ++k)
    if (band[k].n < 14 * n) band[k].alloc(14 * n);
  if (blk_band.n == 0 || band_k != K) {
    std::vector<int32_t> hb = {bl.ilo, bl.ihi, bl.jhi - 2 * K, bl.jhi, 0, 0};
    blk_band.alloc(hb.size());
    blk_band.upload(hb.data(), stream);
    CICE_HIP(hipStreamSynchronize(stream));
    band_k = K;
  }
  if (!stream2) CICE_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));   // (outside any capture)
  if (!ev_fork) CICE_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  if (!ev_join) CICE_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
}

void Evp::launch_subcycle_skew_fold(int ksub, int K) {
  const Block& bl = dom.all[dom.local[0]];
  const int nx = dom.nx_block;
  const int jb = bl.jhi - 2 * K, jm = bl.jhi - K + 1;       // first row of the band, first row taken from it
  CICE_REQUIRE(band[0].n >= 14 * n && band_k == K, "sweep with a fold: the band has not been set up");
  // 1. the band's two copies start as the state of rows jb-1 .. jhi+1
  {
    const size_t off = (size_t)(jb - 2) * nx, len = (size_t)(bl.jhi + 1 - (jb - 1) + 1) * nx;
    hipLaunchKernelGGL(k_band_rows, dim3((unsigned)((14 * len + 255) / 256)), dim3(256), 0, stream, band[0].p, band[1].p,
                       (const double*)st[cur].p, n, off, len);
  }
  // 2. the sweep, as on an open north boundary (no halo update after it: the band brings the top rows); it reads the
  //    same copy of the state as step 1 and writes the other one, which the band does not touch before step 4
  static const bool beside = [] { const char* e = std::getenv("CICE4_AMD_SKEW_FOLD_BESIDE"); return !(e && e[0] == '0'); }();
  if (beside) {
    CICE_HIP(hipEventRecord(ev_fork, stream));
    CICE_HIP(hipStreamWaitEvent(stream2, ev_fork, 0));
    launch_subcycle_skew(ksub, K, /*flip_and_halo=*/false, stream2);
    CICE_HIP(hipEventRecord(ev_join, stream2));
  } else {
    launch_subcycle_skew(ksub, K, /*flip_and_halo=*/false);
  }
  // 3. the band, one subcycle at a time
  bool joined = false;
  for (int s = 0; s < K; ++s) {
    SubArgs a = make_args();
    double* in = band[s & 1].p;
    double* out = band[(s + 1) & 1].p;
    a.u_in = in; a.v_in = in + n; a.sig_in = in + 2 * n;
    a.u_out = out; a.v_out = out + n; a.sig_out = out + 2 * n;
    a.blk = blk_band.p;
    a.diag_jmin = jm;
    const int trows = waves * rows_per_wave;
    a.tiles_x = ((dom.nx_block - 2) + (TX - 1) - 1) / (TX - 1);
    a.tiles_y = ((bl.jhi - jb + 1) + (trows - 1) - 1) / (trows - 1);
    const int nt = a.tiles_x * a.tiles_y;
    const dim3 g(8 * ((nt + 7) / 8));
    const bool last = (ksub + s == sc.ndte), damp = sc.evp_damping != 0;
    // the last subcycle of evp(dt) also writes the diagnostics (strain rates, stress divergence ...): the sweep writes
    // them for every row, the band for its rows above jm -- the band's have to land second
    if (beside && last) { CICE_HIP(hipStreamWaitEvent(stream, ev_join, 0)); joined = true; }
    switch (waves * 100 + rows_per_wave) {
      case 801: launch_wr<8, 1>(a, last, damp, g, stream); break;
      case 802: launch_wr<8, 2>(a, last, damp, g, stream); break;
      case 804: launch_wr<8, 4>(a, last, damp, g, stream); break;
      case 401: launch_wr<4, 1>(a, last, damp, g, stream); break;
      case 402: launch_wr<4, 2>(a, last, damp, g, stream); break;
      case 404: launch_wr<4, 4>(a, last, damp, g, stream); break;
      case 408: launch_wr<4, 8>(a, last, damp, g, stream); break;
      case 1601: launch_wr<16, 1>(a, last, damp, g, stream); break;
      case 1602: launch_wr<16, 2>(a, last, damp, g, stream); break;
      default: throw Error{CICE_EINVAL, "unsupported (waves, rows_per_wave) combination"};
    }
    halo.update_r8(out, 2, n, /*wrap=*/false, LOC_NECORNER, KIND_VECTOR, 0.0, HALO_FOLD);   // the fold (:397-402)
  }
  // 4. rows jm .. jhi+1 of the band replace the sweep's
  if (beside && !joined) CICE_HIP(hipStreamWaitEvent(stream, ev_join, 0));
  {
    const size_t off = (size_t)(jm - 1) * nx, len = (size_t)(bl.jhi + 1 - jm + 1) * nx;
    hipLaunchKernelGGL(k_band_rows, dim3((unsigned)((14 * len + 255) / 256)), dim3(256), 0, stream, st[1 - cur].p,
                       (double*)nullptr, (const double*)band[K & 1].p, n, off, len);
  }
  cur = 1 - cur;
  ++flips;
  if (dom.overlap > 0) {   // top slab of a wide-halo domain: the refresh that is due after this sweep's last subcycle
    const int last = ksub + K - 1;
    if (last % dom.overlap == 0 || last == sc.ndte)
      halo.update_r8(st[cur].p, 14, n, /*wrap=*/false, LOC_CENTER, KIND_SCALAR, 0.0, HALO_COPIES);
  }
  CICE_HIP(hipGetLastError());
}

// ---- resident loop ------------------------------------------------------------------------------------------------
// One block on this rank, nothing to exchange with other ranks or across a tripole fold during the subcycling, on-rank
// ghosts served by forwarding, and at most one tile per CU (the hand-off form used is the one measured for one
// workgroup per CU, and every tile has to be resident for the progress words to advance).
bool Evp::can_reside() const {
  static const bool env_off = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT"); return e && e[0] == '0'; }();
  if (!resident_on || resident_failed || env_off) return false;
  if (!halo.fwd_ok() || dom.nblocks() < 1 || dom.overlap > 0 || halo.multi_rank()) return false;
  if (dom.nblocks() == 1) {
    if (halo.has_refresh()) return false;
  } else {
    // several blocks of one rank: every ghost cell is an on-rank copy (forwarded by its producer), lies beyond an open /
    // closed edge, or faces an eliminated land block (constant: the fill value of the halo update in prepare()); the
    // fold inside the loop is built for one block
    static const bool env_one = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_BLOCKS"); return e && e[0] == '0'; }();
    if (!res_blocks_on || env_one || halo.has_fold()) return false;
    // (slab domains keep the copies between their blocks in the REFRESH list, even without overlap rows: those ghost rows
    //  have no forwarding producer -- such a domain keeps its launch per subcycle)
    if (halo.has_onrank_refresh()) return false;
  }
  if (halo.has_fold() && !res_fold_on) return false;   // the fold inside the loop: option "resident_fold"
  return resident_waves() > 0;
}

// Shape of the resident launch.  One workgroup per CU (W wavefronts, the shortest W that gives every tile its own CU),
// or -- "dense" -- three 4-wavefront workgroups per CU (one wavefront per SIMD each, 3 x 148 VGPRs and 3 x 42 KB of LDS
// fit): the same three wavefronts per SIMD as 11- or 12-wavefront tiles, but while one workgroup waits for its
// hand-off the other two compute.  Dense needs EVERY slot of the chip (gx1: 768 tiles on 256 CUs); if the dispatcher
// does not place them all, the launch times out and the next one uses one workgroup per CU (res_level).
bool Evp::granules_on() const {
  static const int env = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_GRANULES"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  if (env >= 0) return env == 1;
  return res_gran == 2 || (res_gran == 1 && !res_sparse);     // 1: unless the last step's ice cover left most tiles empty (run_resident)
}

bool Evp::resident_dense() const {   // more tiles than CUs: several workgroups per CU
  const int W = resident_waves();
  if (W == 0) return false;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const long long tx = ((dom.nx_block - 2) + (TX - 1) - 1) / (TX - 1);
  return dom.nblocks() * tx * (((dom.ny_block - 2) + (W - 1) - 1) / (W - 1)) > ncu;
}

int Evp::resident_waves() const {
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int ncu_dev = ncu;
  const int share = halo.multi_rank() ? std::max(1, res_peer_share) : 1;   // contexts sharing this device (tests)
  ncu = std::max(1, ncu / share);
  const long long tx = ((dom.nx_block - 2) + (TX - 1) - 1) / (TX - 1);
  auto tiles_raw = [&](int w) { return dom.nblocks() * tx * (((dom.ny_block - 2) + (w - 1) - 1) / (w - 1)); };
  // Several ranks' loops on ONE device (tests only): every workgroup needs a CU of its own and the dispatcher deals the
  // workgroups of a launch round-robin over 8 XCDs x 4 shader engines without moving them between engines afterwards --
  // the `share` launches together must fit every engine (3 launches of 80 workgroups = 3 + 3 + 3 on an engine of 8 CUs
  // did not: one workgroup never started, the loops timed out every few dozen calls; scripts/soak_peer.py)
  auto tiles = [&](int w) -> long long {
    const long long t = tiles_raw(w);
    if (share > 1) {
      const long long per_xcd = (t + 7) / 8, per_se = (per_xcd + 3) / 4;
      if (per_se * share > std::max(1, ncu_dev / 32)) return (long long)ncu + 1;   // does not fit: as if too many tiles
    }
    return t;
  };
  // (the cross-rank and the fold variants carry one more table in LDS and a few more registers: no 12-wavefront
  //  workgroups, no three workgroups per CU)
  const bool plain = !halo.multi_rank() && !halo.has_fold();
  static const bool fold_dense = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_FOLD_DENSE"); return !(e && e[0] == '0'); }();
  const bool dense_ok = res_dense && res_level == 0 && tiles(4) > ncu && tiles(4) <= 3LL * ncu && (plain || (fold_dense && !halo.multi_rank()));
  if (res_w_opt) return (tiles(res_w_opt) <= ncu || (res_w_opt == 4 && dense_ok)) && !(res_w_opt == 12 && !plain) ? res_w_opt : 0;
  int single = 0;
  for (int w : {4, 6, 8, 11, 12})    // the shortest workgroup that still gives every tile its own CU
    if (tiles(w) <= ncu && !(w == 12 && !plain)) {
      single = w;
      break;
    }
  // wavefronts on the busiest SIMD: dense ceil(tiles / CUs), single ceil(W / 4); a tie goes to dense (hand-offs overlap)
  // -- unless the free-running granule loop runs (round 5): ONE workgroup per CU whose wavefronts each wait for exactly what
  // they need beats three barrier-coupled workgroups per CU (gx1: 5.0 us per subcycle against 5.3; the granule loop in the
  // dense shape: 6.8, its polls crowd the CU's memory queue -- profiles/r05_resident_granules.txt)
  if (single && !halo.multi_rank() && granules_on()) return single;     // (also under a tripole fold: round 5)
  if (dense_ok && (single == 0 || (tiles(4) + ncu - 1) / ncu <= (single + 3) / 4)) return 4;
  return single;
}

// ---- the resident loop across ranks (PEER) ----------------------------------------------------------------------
// One block per rank in ANY cartesian layout (round 5; before: one full-width slab per rank): full-width j-slabs, i-slabs
// (bld/config.nci.access-om.360x300: 6 x 1), 2 x 2 tasks (comp_ice:34-46) -- up to eight neighbouring ranks, the ones
// across a cyclic edge and the diagonal ones included.
// The tiles on a block's edges exchange their edge velocities with the neighbouring rank's tiles exactly as
// tiles of one rank do -- stores into the reader's exchange copy, a progress word, a poll -- only that the reader's
// exchange copy and progress words live in another device's memory, mapped into this address space (peer_connect: a
// plain device pointer for two contexts on one GPU, a pointer obtained from the neighbour's IPC handle across GPUs).
// No pack kernel, no RCCL call, no unpack kernel inside the subcycling; the host enqueues ONE launch per evp(dt).
constexpr int RP_MAX = 4096;   // progress words per side a rank keeps for its neighbour's tiles

void Evp::peer_export(void* out[3]) {
  CICE_REQUIRE(ready, "cice_evp_peer_export before cice_evp_init");
  peer_alloc();
  out[0] = res_xu[0].p;
  out[1] = res_xu[1].p;
  out[2] = res_rprog.p;
}

// What a NEIGHBOURING DEVICE writes or polls during a launch -- both exchange copies (its tiles store their edge velocities
// into them with system-scope stores, ours read them with system-scope loads) and the progress words its tiles publish
// here -- lives in fine-grained device memory (DevBuf::alloc_fine); everything else of the loop (own progress words, abort
// word, dependency lists, the state) is only ever touched by this device and stays ordinary hipMalloc memory.
// CICE4_AMD_PEER_COARSE=1 restores plain hipMalloc for an A/B (scripts/peer_two_slabs.py).
void Evp::peer_alloc() {
  static const bool coarse = [] { const char* e = std::getenv("CICE4_AMD_PEER_COARSE"); return e && e[0] == '1'; }();
  for (int k = 0; k < 2; ++k) {
    if (coarse) {
      if (res_xu[k].n < 2 * n) res_xu[k].alloc(2 * n);
    } else if (res_xu[k].n < 2 * n || !res_xu[k].fine) {
      res_xu[k].alloc_fine(2 * n);
    }
  }
  if (res_rprog.n == 0) {
    if (coarse) res_rprog.alloc((size_t)RES_NPEER * RP_MAX * RES_STRIDE);
    else res_rprog.alloc_fine((size_t)RES_NPEER * RP_MAX * RES_STRIDE);
    res_rprog.zero(stream);
    CICE_HIP(hipStreamSynchronize(stream));
  }
}

// The ranks this rank's block exchanges ghost cells with, ascending: the order of the `neighbour` index everywhere in the
// loop (ResArgs::pxu, the blocks of res_rprog, bit numbers of `pub`).
std::vector<int> Evp::peer_ranks() const {
  std::vector<int> v;
  for (const HaloMsg& m : dom.recv) v.push_back(m.peer);
  for (const HaloMsg& m : dom.send) v.push_back(m.peer);
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  return v;
}

void Evp::peer_connect_rank(int prank, void* xu0, void* xu1, void* rprog, long long peer_n) {
  CICE_REQUIRE(ready, "cice_evp_peer_connect before cice_evp_init");
  CICE_REQUIRE(xu0 && xu1 && rprog && peer_n > 0 && peer_n < (1ll << 28), "bad peer buffers");
  const std::vector<int> pr = peer_ranks();
  CICE_REQUIRE((int)pr.size() <= RES_NPEER, "resident EVP loop across ranks: more than eight neighbouring ranks");
  const auto it = std::find(pr.begin(), pr.end(), prank);
  CICE_REQUIRE(it != pr.end(), "cice_evp_peer_connect: that rank is not a neighbour of this rank's block");
  peer_alloc();
  Peer& p = peers[it - pr.begin()];
  p.xu[0] = (double*)xu0;
  p.xu[1] = (double*)xu1;
  p.rprog = (unsigned*)rprog;
  p.n = (unsigned)peer_n;
  res_w = 0;   // rebuild the dependency lists
}

// the older form: side 0 = the rank that owns the block to the south, 1 = to the north (one full-width slab per rank)
void Evp::peer_connect(int side, void* xu0, void* xu1, void* rprog, long long peer_n) {
  CICE_REQUIRE(ready, "cice_evp_peer_connect before cice_evp_init");
  CICE_REQUIRE(side == 0 || side == 1, "side must be 0 (south) or 1 (north)");
  const Block& bl = dom.all[dom.local[0]];
  int jb = bl.jb + (side == 0 ? -1 : 1);
  if (dom.ns == BND_CYCLIC) jb = (jb + dom.nby) % dom.nby;
  CICE_REQUIRE(jb >= 0 && jb < dom.nby, "cice_evp_peer_connect: this slab has no neighbour on that side");
  int prank = -1;
  for (const Block& b : dom.all)
    if (b.ib == bl.ib && b.jb == jb) prank = b.owner;
  CICE_REQUIRE(prank >= 0 && prank != dom.rank, "cice_evp_peer_connect: no other rank owns the block on that side");
  peer_connect_rank(prank, xu0, xu1, rprog, peer_n);
}

bool Evp::can_reside_peer() const {
  if (!resident_on || resident_failed || !halo.multi_rank()) return false;
  if (!halo.fwd_ok() || dom.nblocks() < 1 || dom.overlap > 0) return false;
  // a tripole grid: full-width slabs, one per rank -- the partner of every top-row cell then lies on the rank of the top
  // slab itself, which runs the PEER && FOLD form of the loop; the other ranks run as on any grid
  if ((dom.tripole() || halo.has_fold()) && (dom.nbx != 1 || dom.nblocks() != 1 || !res_fold_on)) return false;
  if (halo.has_onrank_refresh()) return false;
  // any number of blocks per rank, every block of the same size (a neighbour's tiles are numbered block by block on OUR
  // tile grid).  Ghost cells that face an ELIMINATED land block have neither a message nor an on-rank source: nobody
  // produces them, nobody waits for them, they keep the fill value the halo update of prepare() gave them -- as in the
  // one-rank loop and in the reference (its halo update fills them, mpi/ice_boundary.F90:1145-1404).
  // every neighbour this block has must be connected
  const std::vector<int> pr = peer_ranks();
  if (pr.empty() || (int)pr.size() > RES_NPEER) return false;
  for (size_t k = 0; k < pr.size(); ++k)
    if (!peers[k].rprog) return false;
  return resident_waves() > 0;
}

void Evp::build_resident_peer(int W) {
  // One or SEVERAL blocks per rank (round 5: tiles are numbered block by block, as in build_resident; a cell's address is
  // its address in the rank's arrays; ghost cells between two blocks of this rank are forwarded by the on-rank lists of
  // Halo, ghost cells owned by another rank by rslot / rfwd).
  const int nx = dom.nx_block, ny = dom.ny_block, nb = dom.nblocks();
  const int tiles_x = ((nx - 2) + (TX - 1) - 1) / (TX - 1), tiles_y = ((ny - 2) + (W - 1) - 1) / (W - 1);
  const int per_blk = tiles_x * tiles_y, nt = per_blk * nb;
  CICE_REQUIRE(nt <= RP_MAX, "resident EVP loop across ranks: too many tiles");
  const size_t np = (size_t)nx * ny;
  // the neighbours' view of the same decomposition (host logic only): their send / receive lists pair up with ours
  // element by element (Domain::build visits the ghost cells in one global order)
  std::map<int, Domain> pd;
  auto peer_dom = [&](int prank) -> const Domain& {
    auto it = pd.find(prank);
    if (it == pd.end()) {
      Domain d;
      const char* msg = dom.from_map
                            ? d.create_map(dom.nxg, dom.nyg, dom.bsx, dom.bsy, dom.ew, dom.ns, prank, dom.nranks,
                                           dom.map_owner.data(), dom.map_lid.data())
                            : d.create(dom.nxg, dom.nyg, dom.bsx, dom.bsy, dom.ew, dom.ns, prank, dom.npx, dom.npy);
      CICE_REQUIRE(!msg[0], "resident EVP loop across ranks: cannot rebuild a neighbour's decomposition");
      it = pd.emplace(prank, std::move(d)).first;
    }
    return it->second;
  };
  auto msg_with = [](const std::vector<HaloMsg>& v, int peer) -> const HaloMsg* {
    for (const HaloMsg& m : v)
      if (m.peer == peer) return &m;
    return nullptr;
  };
  // my ghost cell -> (neighbour, its tile that produces it); my edge cell -> ghost cells of neighbours
  const std::vector<int> pranks = peer_ranks();
  auto slot_of = [&](int prank) { return (int)(std::find(pranks.begin(), pranks.end(), prank) - pranks.begin()); };
  std::vector<int32_t> rsrc_tile(np * nb, -1);       // dep code (<= -2) of a ghost cell owned by another rank
  std::vector<int32_t> rslot(np * nb, -1), rfwd;
  for (const HaloMsg& m : dom.recv) {
    const Domain& d = peer_dom(m.peer);
    const HaloMsg* ps = msg_with(d.send, dom.rank);
    CICE_REQUIRE(ps && ps->addr.size() == m.addr.size(), "resident EVP loop across ranks: message lists do not pair up");
    CICE_REQUIRE(d.nx_block == nx && d.ny_block == ny, "resident EVP loop across ranks: the neighbour's blocks have another size");
    CICE_REQUIRE((long long)per_blk * d.nblocks() <= RP_MAX, "resident EVP loop across ranks: the neighbour has too many tiles");
    const int side = slot_of(m.peer);
    for (size_t e = 0; e < m.addr.size(); ++e) {
      const int q = m.addr[e];
      const int sq = ps->addr[e], plb = (int)((size_t)sq / np), sqq = (int)((size_t)sq - (size_t)plb * np);
      const int si = sqq % nx + 1, sj = sqq / nx + 1;
      const Block& pb = d.all[d.local[plb]];
      const int ptile = plb * per_blk + ((sj - pb.jlo) / (W - 1)) * tiles_x + (si - pb.ilo) / (TX - 1);
      CICE_REQUIRE(ptile >= 0 && ptile < RP_MAX, "resident EVP loop across ranks: neighbour tile out of range");
      rsrc_tile[q] = -2 - (side * RP_MAX + ptile);
      rslot[q] = -2;
    }
  }
  for (const HaloMsg& m : dom.send) {
    const Domain& d = peer_dom(m.peer);
    const HaloMsg* pr = msg_with(d.recv, dom.rank);
    CICE_REQUIRE(pr && pr->addr.size() == m.addr.size(), "resident EVP loop across ranks: message lists do not pair up");
    const int side = slot_of(m.peer);
    for (size_t e = 0; e < m.addr.size(); ++e) {
      const int q = m.addr[e];
      const int gq = pr->addr[e];
      CICE_REQUIRE(gq < (1 << 28), "resident EVP loop across ranks: neighbour plane too large");
      int32_t& sl = rslot[q];
      if (sl < 0) {
        sl = (int32_t)(rfwd.size() / 4);
        rfwd.insert(rfwd.end(), {-1, -1, -1, -1});
      }
      int k = 0;
      while (k < 4 && rfwd[4 * sl + k] >= 0) ++k;
      CICE_REQUIRE(k < 4, "resident EVP loop across ranks: a cell is mirrored by more than four remote ghost cells");
      rfwd[4 * sl + k] = (side << 28) | gq;
    }
  }
  if (rfwd.empty()) rfwd.assign(4, -1);
  std::vector<int32_t> src_of(np * nb, -1);
  for (size_t e = 0; e < dom.hsrc.size(); ++e) src_of[dom.hdst[e]] = dom.hsrc[e];
  // the top slab of a tripole grid: its fold is this rank's own affair, exactly as on one rank (the ghost row above it
  // gets its sources from the fold, the top row its partners; can_reside_peer has checked that the block spans the width)
  if (halo.has_fold()) build_resident_fold(src_of, tiles_x, W);
  auto owner = [&](int b, int i, int j) -> int {   // block, 1-based cell -> tile producing its velocity: local >= 0, remote <= -2, -1 nobody
    if (i < 1 || i > nx || j < 1 || j > ny) return -1;
    size_t q = (size_t)b * np + (size_t)(j - 1) * nx + (i - 1);
    if (rsrc_tile[q] <= -2) return rsrc_tile[q];
    if (src_of[q] >= 0) q = (size_t)src_of[q];
    const int sb = (int)(q / np);
    const size_t qq = q - (size_t)sb * np;
    const int si = (int)(qq % nx) + 1, sj = (int)(qq / nx) + 1;
    const Block& sbl = dom.all[dom.local[sb]];
    if (si < sbl.ilo || si > sbl.ihi || sj < sbl.jlo || sj > sbl.jhi) return -1;
    return sb * per_blk + ((sj - sbl.jlo) / (W - 1)) * tiles_x + (si - sbl.ilo) / (TX - 1);
  };
  std::vector<int32_t> deps((size_t)nt * RES_MAXDEP, -1);
  for (int t = 0; t < nt; ++t) {
    const int b = t / per_blk, rem = t - b * per_blk;
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const Block& bl = dom.all[dom.local[b]];
    const int ilo = bl.ilo, ihi = bl.ihi, jlo = bl.jlo, jhi = bl.jhi;
    const int i0 = ilo + txi * (TX - 1), j0 = jlo + tyi * (W - 1);
    if (i0 > ihi || j0 > jhi) continue;   // (a block smaller than the largest: the kernel's workgroup leaves at once)
    int nd = 0;
    auto add = [&](int i, int j) {
      const int o = owner(b, i, j);
      if (o == -1 || o == t) return;
      for (int k = 0; k < nd; ++k)
        if (deps[(size_t)t * RES_MAXDEP + k] == o) return;
      CICE_REQUIRE(nd < RES_MAXDEP, "resident EVP loop: a tile has more producers than RES_MAXDEP");
      deps[(size_t)t * RES_MAXDEP + nd++] = o;
    };
    for (int w = -1; w < W; ++w)
      for (int lx = -1; lx < TX; ++lx) {
        const int i = i0 + lx, j = j0 + w;
        if (i > ihi + 1 || j > jhi + 1) continue;
        const bool uown = lx >= 0 && w >= 0 && lx < TX - 1 && i <= ihi && w < W - 1 && j <= jhi;
        if (!uown) add(i, j);
      }
  }
  std::vector<int32_t> pub(nt, 0);   // tiles whose OWNED U-cells are mirrored on neighbour s: bit s
  // where this rank's tiles publish their progress on neighbour s: the block of ITS res_rprog that belongs to US, i.e. the
  // place of this rank in the neighbour's own ascending list of neighbours
  peer_back.assign(RES_NPEER, 0);
  for (size_t k = 0; k < pranks.size(); ++k) {
    const Domain& d = peer_dom(pranks[k]);
    std::vector<int> v;
    for (const HaloMsg& m : d.recv) v.push_back(m.peer);
    for (const HaloMsg& m : d.send) v.push_back(m.peer);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    const auto it = std::find(v.begin(), v.end(), dom.rank);
    CICE_REQUIRE(it != v.end(), "resident EVP loop across ranks: a neighbour does not list this rank");
    peer_back[k] = (int)(it - v.begin());
  }
  for (int b = 0; b < nb; ++b) {
    const Block& bl = dom.all[dom.local[b]];
    for (int j = bl.jlo; j <= bl.jhi; ++j)
      for (int i = bl.ilo; i <= bl.ihi; ++i) {
        const int sl = rslot[(size_t)b * np + (size_t)(j - 1) * nx + (i - 1)];
        if (sl < 0) continue;
        const int t = b * per_blk + ((j - bl.jlo) / (W - 1)) * tiles_x + (i - bl.ilo) / (TX - 1);
        for (int k = 0; k < 4; ++k)
          if (rfwd[4 * sl + k] >= 0) pub[t] |= 1 << ((rfwd[4 * sl + k] >> 28) & 7);
      }
  }
  res_pub.alloc(pub.size());
  res_pub.upload(pub.data(), stream);
  res_deps.alloc(deps.size());
  res_deps.upload(deps.data(), stream);
  res_rslot.alloc(rslot.size());
  res_rslot.upload(rslot.data(), stream);
  res_rfwd.alloc(rfwd.size());
  res_rfwd.upload(rfwd.data(), stream);
  res_prog.alloc((size_t)(nt + 1) * RES_STRIDE);
  res_prog.zero(stream);
  peer_alloc();
  CICE_HIP(hipStreamSynchronize(stream));
  res_w = W;
  res_tiles = nt;
  res_peer_built = true;
}

// producer tiles of every tile's halo: the cells it re-reads each subcycle, traced through the on-rank ghost copies
// The tripole fold (serial/ice_boundary.F90:705-869) inside the one-launch loop, for u and v (NE corner, vector).  From
// the Domain's own fold lists -- buffer fill, symmetric pairs, copy-out with the location's offsets -- every cell the
// fold writes gets its expression in the raw velocities of the subcycle: s * raw(A), or s * 0.5 * (raw(A) - raw(B)).
//  * owned cells of the top row: the kernel evaluates the expression itself (ftab: partner + mode), the raw values of
//    the partners come through xraw / prog2 from the tiles in deps2;
//  * ghost cells: the same expression as some owned cell's final value, up to the sign -- that cell forwards its value
//    to the ghost position (rslot / rfwd, bit 30 = negate) and becomes the ghost cell's source for the dependency lists.
// Anything else (an expression no owned cell carries, more than four images of a cell) is not a domain for this loop.
void Evp::build_resident_fold(std::vector<int32_t>& src_of, int tiles_x, int W) {
  const int nx = dom.nx_block, ny = dom.ny_block;
  const Block& bl = dom.all[dom.local[0]];
  const int ilo = bl.ilo, ihi = bl.ihi, jlo = bl.jlo, jhi = bl.jhi;
  const size_t np = (size_t)nx * ny;
  const int l = LOC_NECORNER - 1, sgn = -1;
  const size_t nbuf = (size_t)dom.fold_rows() * dom.nxg;
  std::vector<int32_t> baddr(nbuf, -1), plo(nbuf, -1), phi(nbuf, -1);
  for (size_t e = 0; e < dom.fold_lsrc.size(); ++e) baddr[dom.fold_bidx[e]] = dom.fold_lsrc[e];
  for (size_t e = 0; e < dom.fold_lo[l].size(); ++e) {
    const int32_t lo = dom.fold_lo[l][e], hi = dom.fold_hi[l][e];
    plo[lo] = lo; phi[lo] = hi; plo[hi] = lo; phi[hi] = hi;
  }
  auto owned = [&](int32_t q) {
    const int i = q % nx + 1, j = q / nx + 1;
    return i >= ilo && i <= ihi && j >= jlo && j <= jhi;
  };
  struct Ex { int pair; int32_t A, B; int s; };   // pair: s * 0.5 * (raw(A) - raw(B)), else s * raw(A)
  auto same = [](const Ex& x, const Ex& y) { return x.pair == y.pair && x.A == y.A && (!x.pair || x.B == y.B); };
  const Domain::FoldOut& fo = dom.fold_out[l];
  std::vector<Ex> ex(fo.dst.size());
  for (size_t e = 0; e < fo.dst.size(); ++e) {
    const int32_t b = fo.src[e];
    CICE_REQUIRE(b >= 0 && (size_t)b < nbuf && baddr[b] >= 0, "resident EVP loop: the fold reads a buffer cell nobody fills");
    if (plo[b] >= 0) {
      CICE_REQUIRE(baddr[plo[b]] >= 0 && baddr[phi[b]] >= 0, "resident EVP loop: fold pair outside the buffer");
      ex[e] = Ex{1, baddr[plo[b]], baddr[phi[b]], b == plo[b] ? sgn : 1};   // sgn * x, or sgn * (sgn * x)
    } else {
      ex[e] = Ex{0, baddr[b], -1, sgn};
    }
  }
  // owned cells the fold writes: all in the top row; every other owned cell keeps its raw value
  std::vector<int32_t> ftab(2 * np, 0);
  for (size_t q = 0; q < np; ++q) ftab[2 * q] = -1;
  std::vector<int> ex_of(np, -1);
  for (size_t e = 0; e < fo.dst.size(); ++e) {
    const int32_t d = fo.dst[e];
    if (!owned(d)) continue;
    CICE_REQUIRE(d / nx + 1 == jhi, "resident EVP loop: the fold writes an owned cell below the top row");
    CICE_REQUIRE(ex_of[d] < 0, "resident EVP loop: the fold writes a cell twice");
    ex_of[d] = (int)e;
    const Ex& x = ex[e];
    CICE_REQUIRE(owned(x.A) && (!x.pair || owned(x.B)), "resident EVP loop: the fold reads a cell this block does not own");
    int mode;
    int32_t partner = -1;
    if (x.pair) {
      CICE_REQUIRE(d == x.A || d == x.B, "resident EVP loop: a top-row cell averaged from two other cells");
      mode = (d == x.A ? F_LO : F_HI) | (x.s < 0 ? F_NEG : 0);
      partner = d == x.A ? x.B : x.A;
    } else if (x.A == d) {
      CICE_REQUIRE(x.s < 0, "resident EVP loop: identity in the fold");
      mode = F_SELF;
    } else {
      mode = F_MIRROR | (x.s < 0 ? F_NEG : 0);
      partner = x.A;
    }
    CICE_REQUIRE(partner < 0 || (int)(partner / nx) + 1 == jhi, "resident EVP loop: fold partner below the top row");
    ftab[2 * d] = partner;
    ftab[2 * d + 1] = mode;
  }
  // ghost cells: find the owned cell whose final value is the same expression
  std::vector<std::vector<int32_t>> images(np);   // per owned cell: (neg << 30) | ghost address
  for (size_t e = 0; e < fo.dst.size(); ++e) {
    const int32_t g = fo.dst[e];
    if (owned(g)) continue;
    const Ex& x = ex[e];
    int32_t src = -1;
    int rel = 1;
    if (!x.pair && owned(x.A) && ex_of[x.A] < 0) {     // an owned cell the fold leaves alone: final = raw
      src = x.A;
      rel = x.s;
    } else {                                             // a cell of the top row with the same expression
      for (int32_t c : {x.A, x.pair ? x.B : x.A}) {
        if (c < 0 || !owned(c) || ex_of[c] < 0) continue;
        // candidates: the members themselves and, for a mirror image, the partner that mirrors this member
        for (int32_t cand : {c, ftab[2 * c]}) {
          if (cand < 0 || ex_of[cand] < 0) continue;
          const Ex& y = ex[ex_of[cand]];
          if (same(x, y)) { src = cand; rel = x.s * y.s; }
        }
      }
    }
    CICE_REQUIRE(src >= 0, "resident EVP loop: a ghost cell of the fold mirrors no owned cell's value");
    // a regular copy into the same ghost cell (east-west wrap of the top row) is overwritten by the fold in the reference;
    // here both producers store: they have to store the same value
    if (src_of[g] >= 0 && src_of[g] != src) {
      const int32_t o = src_of[g];
      bool eq = false;
      if (ex_of[o] >= 0) eq = same(x, ex[ex_of[o]]) && x.s == ex[ex_of[o]].s;
      else eq = !x.pair && x.A == o && x.s > 0;
      CICE_REQUIRE(eq, "resident EVP loop: wrap and fold disagree on a ghost cell");
      continue;                                         // the wrap's producer already forwards exactly this value
    }
    src_of[g] = src;
    images[src].push_back((rel < 0 ? (1 << 30) : 0) | g);
  }
  std::vector<int32_t> rslot(np, -1), rfwd;
  for (size_t q = 0; q < np; ++q) {
    if (images[q].empty()) continue;
    CICE_REQUIRE(images[q].size() <= 4, "resident EVP loop: a cell with more than four images across the fold");
    rslot[q] = (int32_t)(rfwd.size() / 4);
    for (int c = 0; c < 4; ++c) rfwd.push_back(c < (int)images[q].size() ? images[q][c] : -1);
  }
  if (rfwd.empty()) rfwd.assign(4, -1);
  // tiles holding the partners of a tile's top-row cells
  const int tiles_y = ((ny - 2) + (W - 1) - 1) / (W - 1), nt = tiles_x * tiles_y;
  auto tile_of = [&](int32_t q) { return ((int)(q / nx) + 1 - jlo) / (W - 1) * tiles_x + ((int)(q % nx) + 1 - ilo) / (TX - 1); };
  std::vector<int32_t> deps2((size_t)nt * 4, -1);
  for (int i = ilo; i <= ihi; ++i) {
    const int32_t d = (int32_t)((jhi - 1) * nx + (i - 1));
    if (ftab[2 * d] < 0) continue;
    const int t = tile_of(d), o = tile_of(ftab[2 * d]);
    if (o == t) continue;
    int k = 0;
    while (k < 4 && deps2[(size_t)t * 4 + k] >= 0 && deps2[(size_t)t * 4 + k] != o) ++k;
    CICE_REQUIRE(k < 4, "resident EVP loop: a top-row tile with partners in more than four tiles");
    deps2[(size_t)t * 4 + k] = o;
  }
  res_ftab.alloc(ftab.size()); res_ftab.upload(ftab.data(), stream);
  res_fslot.alloc(rslot.size()); res_fslot.upload(rslot.data(), stream);
  res_ffwd.alloc(rfwd.size()); res_ffwd.upload(rfwd.data(), stream);
  res_deps2.alloc(deps2.size()); res_deps2.upload(deps2.data(), stream);
  res_prog2.alloc((size_t)nt * RES_STRIDE); res_prog2.zero(stream);
  for (int k = 0; k < 2; ++k)
    if (res_xraw[k].n < 2 * n) { res_xraw[k].alloc(2 * n); res_xraw[k].zero(stream); }
  CICE_HIP(hipStreamSynchronize(stream));   // the host vectors go out of scope
}

void Evp::build_resident(int W) {
  // One rank, one or SEVERAL blocks (source/ice_blocks.F90:133-330: any block size; comp_ice:34-46 gives the serial
  // build many small ones): tiles are numbered block by block, a cell's address is its address in the rank's arrays
  // (block plane + position), which is what the on-rank ghost copies dom.hsrc / hdst and the forwarding lists of
  // Halo use.  A ghost cell is produced by the tile that owns its SOURCE cell -- in whichever block that lies.
  const int nx = dom.nx_block, ny = dom.ny_block, nb = dom.nblocks();
  const size_t np = (size_t)nx * ny;
  const int tiles_x = ((nx - 2) + (TX - 1) - 1) / (TX - 1), tiles_y = ((ny - 2) + (W - 1) - 1) / (W - 1);
  const int per_blk = tiles_x * tiles_y, nt = per_blk * nb;
  std::vector<int32_t> src_of(np * nb, -1);
  for (size_t e = 0; e < dom.hsrc.size(); ++e) src_of[dom.hdst[e]] = dom.hsrc[e];
  auto owner = [&](int b, int i, int j) -> int {   // block, 1-based cell -> tile that produces its velocity, -1: nobody (constant)
    if (i < 1 || i > nx || j < 1 || j > ny) return -1;
    size_t q = (size_t)b * np + (size_t)(j - 1) * nx + (i - 1);
    if (src_of[q] >= 0) q = (size_t)src_of[q];
    const int sb = (int)(q / np);
    const size_t qq = q - (size_t)sb * np;
    const int si = (int)(qq % nx) + 1, sj = (int)(qq / nx) + 1;
    const Block& s = dom.all[dom.local[sb]];
    if (si < s.ilo || si > s.ihi || sj < s.jlo || sj > s.jhi) return -1;
    return sb * per_blk + ((sj - s.jlo) / (W - 1)) * tiles_x + (si - s.ilo) / (TX - 1);
  };
  if (halo.has_fold()) {
    CICE_REQUIRE(nb == 1, "resident EVP loop: the tripole fold inside the loop is built for one block");
    build_resident_fold(src_of, tiles_x, W);   // ghost cells the fold fills now have a source as well
  }
  std::vector<int32_t> deps((size_t)nt * RES_MAXDEP, -1);
  for (int t = 0; t < nt; ++t) {
    const int b = t / per_blk, rem = t - b * per_blk;
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const Block& bl = dom.all[dom.local[b]];
    const int ilo = bl.ilo, ihi = bl.ihi, jlo = bl.jlo, jhi = bl.jhi;
    const int i0 = ilo + txi * (TX - 1), j0 = jlo + tyi * (W - 1);
    if (i0 > ihi || j0 > jhi) continue;   // (a block smaller than the largest: the kernel's workgroup leaves at once)
    int nd = 0;
    auto add = [&](int i, int j) {
      const int o = owner(b, i, j);
      if (o < 0 || o == t) return;
      for (int k = 0; k < nd; ++k)
        if (deps[(size_t)t * RES_MAXDEP + k] == o) return;
      CICE_REQUIRE(nd < RES_MAXDEP, "resident EVP loop: a tile has more producers than RES_MAXDEP");
      deps[(size_t)t * RES_MAXDEP + nd++] = o;
    };
    for (int w = -1; w < W; ++w)
      for (int lx = -1; lx < TX; ++lx) {
        const int i = i0 + lx, j = j0 + w;
        if (i > ihi + 1 || j > jhi + 1) continue;
        const bool uown = lx >= 0 && w >= 0 && lx < TX - 1 && i <= ihi && w < W - 1 && j <= jhi;
        if (!uown) add(i, j);
      }
  }
  res_deps.alloc(deps.size());
  res_deps.upload(deps.data(), stream);
  res_prog.alloc((size_t)(nt + 1) * RES_STRIDE);
  res_prog.zero(stream);
  res_epoch = 0;
  for (int k = 0; k < 2; ++k)
    if (res_xu[k].n < 2 * n) res_xu[k].alloc(2 * n);
  {
    // granule hand-off: whose velocity a cell holds (the kernel polls a cell only if that U-cell carries ice), and the two
    // granule copies -- zeroed once: tags start at 1 and only ever grow
    std::vector<int32_t> src((size_t)np * nb, -1);
    for (int b = 0; b < nb; ++b) {
      for (int j = 1; j <= ny; ++j)
        for (int i = 1; i <= nx; ++i) {
          const size_t q = (size_t)b * np + (size_t)(j - 1) * nx + (i - 1);
          const size_t sq = src_of[q] >= 0 ? (size_t)src_of[q] : q;
          const int sb = (int)(sq / np);
          const size_t qq = sq - (size_t)sb * np;
          const int si = (int)(qq % nx) + 1, sj = (int)(qq / nx) + 1;
          const Block& sbl = dom.all[dom.local[sb]];
          if (si >= sbl.ilo && si <= sbl.ihi && sj >= sbl.jlo && sj <= sbl.jhi) src[q] = (int32_t)sq;
        }
    }
    res_src.alloc(src.size());
    res_src.upload(src.data(), stream);
    CICE_REQUIRE((unsigned long long)n * 64ull < (1ull << 32), "resident EVP loop: the granule copies are addressed by 32-bit offsets");
    if (res_xg.n < 16 * n) {      // (room for four copies: GRAN_COPIES)
      res_xg.alloc(16 * n);
      res_xg.zero(stream);
    }
    if (halo.has_fold() && res_xgr.n < 16 * n) {     // the raw top row across the fold: four copies (see the kernel)
      CICE_REQUIRE((unsigned long long)n * 128ull < (1ull << 32), "resident EVP loop: the raw granule copies are addressed by 32-bit offsets");
      res_xgr.alloc(16 * n);
      res_xgr.zero(stream);
    }
  }
  CICE_HIP(hipStreamSynchronize(stream));
  res_w = W;
  res_tiles = nt;
  res_peer_built = false;
}

template <int W>
static void launch_res(const ResArgs& r, bool damp, bool peer, dim3 g, hipStream_t s) {
  if (peer) {
    if constexpr (W <= 11) {   // (12 wavefronts + the table of remote ghost cells do not fit the LDS)
      if (r.ftab) {
        if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, true, true>), g, dim3(64 * W), 0, s, r);
        else hipLaunchKernelGGL((k_evp_resident<W, false, true, true>), g, dim3(64 * W), 0, s, r);
      } else {
        if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, true>), g, dim3(64 * W), 0, s, r);
        else hipLaunchKernelGGL((k_evp_resident<W, false, true>), g, dim3(64 * W), 0, s, r);
      }
    } else {
      throw Error{CICE_EINVAL, "resident EVP loop across ranks: at most 11 wavefronts per workgroup"};
    }
  } else if (r.ftab) {
    if constexpr (W <= 11) {   // (as above)
      if (r.xg) {
        if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, false, true, true>), g, dim3(64 * W), 0, s, r);
        else hipLaunchKernelGGL((k_evp_resident<W, false, false, true, true>), g, dim3(64 * W), 0, s, r);
      } else {
        if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, false, true>), g, dim3(64 * W), 0, s, r);
        else hipLaunchKernelGGL((k_evp_resident<W, false, false, true>), g, dim3(64 * W), 0, s, r);
      }
    } else {
      throw Error{CICE_EINVAL, "resident EVP loop with a tripole fold: at most 11 wavefronts per workgroup"};
    }
  } else if (r.xg) {
    if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, false, false, true>), g, dim3(64 * W), 0, s, r);
    else hipLaunchKernelGGL((k_evp_resident<W, false, false, false, true>), g, dim3(64 * W), 0, s, r);
  } else {
    if (damp) hipLaunchKernelGGL((k_evp_resident<W, true, false>), g, dim3(64 * W), 0, s, r);
    else hipLaunchKernelGGL((k_evp_resident<W, false, false>), g, dim3(64 * W), 0, s, r);
  }
}

template <int W>
static int occ_res(bool damp, bool peer, bool fold = false, bool gran = false) {
  int nb = 0;
  hipError_t e;
  if (gran && !peer && !fold) {
    e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, false, false, true>, 64 * W, 0)
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, false, false, true>, 64 * W, 0);
  } else if (fold && !peer) {
    if constexpr (W <= 11) {
      if (gran)
        e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, false, true, true>, 64 * W, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, false, true, true>, 64 * W, 0);
      else
        e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, false, true>, 64 * W, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, false, true>, 64 * W, 0);
    } else {
      return 0;
    }
  } else if (peer) {
    if constexpr (W <= 11) {
      if (fold)
        e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, true, true>, 64 * W, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, true, true>, 64 * W, 0);
      else
      e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, true>, 64 * W, 0)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, true>, 64 * W, 0);
    } else {
      return 0;
    }
  } else {
    e = damp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, true, false>, 64 * W, 0)
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_evp_resident<W, false, false>, 64 * W, 0);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb;
}

// workgroups of this instantiation one CU holds, as the runtime computes it from the code object (registers, LDS): the
// loop needs every tile resident at once, and a kernel that grew past its budget must not find that out by time-out
int Evp::resident_occupancy(int W, bool damp, bool peer) {
  const int wi = W == 4 ? 0 : W == 6 ? 1 : W == 8 ? 2 : W == 11 ? 3 : 4;
  const bool fold = halo.has_fold();
  const bool gran = !peer && granules_on() && !resident_dense();
  int& c = res_occ[wi][damp][peer ? (fold ? 5 : 1) : (fold ? (gran ? 4 : 2) : (gran ? 3 : 0))];
  if (c == 0) {
    int nb = 0;
    switch (W) {
      case 4: nb = occ_res<4>(damp, peer, fold, gran); break;
      case 6: nb = occ_res<6>(damp, peer, fold, gran); break;
      case 8: nb = occ_res<8>(damp, peer, fold, gran); break;
      case 11: nb = occ_res<11>(damp, peer, fold, gran); break;
      case 12: nb = occ_res<12>(damp, peer, fold, gran); break;
      default: break;
    }
    c = nb > 0 ? nb : -1;
  }
  return c > 0 ? c : 0;
}

// subcycles ksub0 .. ksub0+nsub-1 in one launch; false: not done (time-out), the state is as it was
bool Evp::run_resident(int ksub0, int nsub) {
  const bool peer = halo.multi_rank();
  const int W = resident_waves();
  if (W != res_w || res_deps.n == 0 || peer != res_peer_built) {
    res_map_stale = true;     // another tiling: the tile map is chosen again
    try {
      if (peer) build_resident_peer(W);
      else build_resident(W);
    } catch (const Error& e) {   // e.g. a tile with more producers than RES_MAXDEP: not a domain for this loop
      if (peer) std::fprintf(stderr, "cice4_amd: resident EVP loop across ranks not possible: %s\n", e.msg.c_str());
      resident_failed = true;
      return false;
    }
  }
  // progress words are compared modulo 2^32 over at most 2^31.  (Across ranks every rank runs the same sequence of
  // launches, so the epochs -- and this reset -- stay in step; the neighbours' words about us are theirs to reset.)
  if (res_epoch > 0x70000000u) {
    CICE_REQUIRE(!peer, "resident EVP loop across ranks: progress epoch exhausted (re-create the context)");
    res_prog.zero(stream);
    if (res_prog2.p) res_prog2.zero(stream);
    if (res_xg.p) res_xg.zero(stream);
    if (res_xgr.p) res_xgr.zero(stream);
    res_epoch = 0;
  }
  ResArgs r{};
  r.a = make_args();
  r.a.tiles_x = ((dom.nx_block - 2) + (TX - 1) - 1) / (TX - 1);
  r.a.tiles_y = ((dom.ny_block - 2) + (W - 1) - 1) / (W - 1);
  r.nsub = nsub;
  r.last = (ksub0 + nsub - 1 == sc.ndte) ? 1 : 0;
  r.epoch0 = res_epoch;
  r.prog = res_prog.p;
  r.abort_flag = res_prog.p + (size_t)res_tiles * RES_STRIDE;
  r.deps = res_deps.p;
  r.xu[0] = res_xu[0].p;
  r.xu[1] = res_xu[1].p;
  r.spin_ticks = (long long)res_spin_us * 100;   // wall_clock64() runs at 100 MHz
  if (halo.has_fold()) {   // (one rank, or the rank with the top slab of a tripole grid cut into full-width slabs)
    r.fslot = res_fslot.p;
    r.ffwd = res_ffwd.p;
    r.ftab = res_ftab.p;
    r.xraw[0] = res_xraw[0].p;
    r.xraw[1] = res_xraw[1].p;
    r.prog2 = res_prog2.p;
    r.deps2 = res_deps2.p;
  }
  if (peer) {
    r.rslot = res_rslot.p;
    r.rfwd = res_rfwd.p;
    r.rprog = res_rprog.p;
    r.pub = res_pub.p;
    for (int sd = 0; sd < RES_NPEER; ++sd) {
      r.pxu[sd][0] = peers[sd].xu[0];
      r.pxu[sd][1] = peers[sd].xu[1];
      r.pn[sd] = peers[sd].n;
      // (on the neighbour we are ITS neighbour number peer_back[sd])
      r.prp[sd] = peers[sd].rprog ? peers[sd].rprog + (size_t)peer_back[(size_t)sd] * RP_MAX * RES_STRIDE : nullptr;
    }
  }
  const bool dense = !peer && resident_dense();
  const bool gran = !peer && granules_on() && !dense && res_xg.p && res_src.p && (!halo.has_fold() || res_xgr.p);
  if (gran) {   // data-tagged granules: nothing to initialise (cells nobody publishes are not polled either)
    r.src = res_src.p;
    r.xg = res_xg.p;
    r.xg_half = (unsigned)(n * 32);
    r.xgr = res_xgr.p;
    static const int pd = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_POLL_DELAY"); return e ? std::atoi(e) : 2; }();
    static const int ps = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_POLL_SLEEP"); return e ? std::atoi(e) : 0; }();
    r.poll_delay = pd;
    r.poll_sleep = ps;
    static const int fk = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_FAKE_EW"); return e ? std::atoi(e) : 0; }();
    r.fake_ew = fk;
  } else {
    for (int k = 0; k < 2; ++k)   // cells nobody publishes keep their value: both exchange copies start as (u, v)
      CICE_HIP(hipMemcpyAsync(res_xu[k].p, uv[cur].p, 2 * n * 8, hipMemcpyDeviceToDevice, stream));
  }
  const dim3 g(8 * ((res_tiles + 7) / 8));
  const bool damp = sc.evp_damping != 0;
  {   // every workgroup of the launch has to be resident at once
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const int per_cu = resident_occupancy(W, damp, peer);
    if ((long long)per_cu * ncu < (long long)g.x) {
      std::fprintf(stderr, "cice4_amd: resident EVP loop not used: %u workgroups of %d wavefronts, the device holds %d x %d "
                           "of this kernel\n", g.x, W, per_cu, ncu);
      if (dense) res_level = 1;
      else resident_failed = true;   // a property of the build and the device: not retried
      return false;
    }
  }
  // (the granule loop: a four-step rotation, 196.6 k subcycles/s at gx1 against 194.5 k with the three-step one and 169 k without)
  r.prio_mode = gran ? (res_prio == 2 ? 4 : res_prio) : (dense ? std::min(res_prio, 2) : 0);
  {
    static const bool top = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_PRIO_TOP"); return !(e && e[0] == '0'); }();
    r.prio_top = dense && top ? 1 : 0;
  }
  r.tile_map = nullptr;
  // (not under a fold: there the period is the top-row tiles' second hand-off, not the arithmetic -- gx1 tripole with polar caps
  //  6.37 us per subcycle with the map against 6.31 without)
  if (!peer && !halo.has_fold() && g.x <= 1024) {     // one choice per evp(dt) (prepare() marks it stale: the masks are new)
    if (res_map.n == 0) { res_map.alloc(4); res_map_stale = true; }
    if (res_map_stale) {
      static const int force = [] { const char* e = std::getenv("CICE4_AMD_RESIDENT_MAP"); return e ? std::atoi(e) : -1; }();
      int ncu = 256, dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
      }
      hipLaunchKernelGGL(k_res_choose_map, dim3(1), dim3(1024), 0, stream, r.a.tiles_x * r.a.tiles_y * r.a.nblocks, r.a.tiles_x,
                         r.a.tiles_y, W, dom.nx_block, dom.ny_block, std::max(1, ncu / 8), res_map_opt >= 0 ? res_map_opt : force,
                         (const int32_t*)blk.p, (const int32_t*)icetmask.p, (const int32_t*)iceumask.p, res_map.p, r.abort_flag + 7);
      res_map_stale = false;
    }
    r.tile_map = res_map.p;
  }
  {
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    r.prio_div = std::max(1, ncu / 8);
  }
  r.stamps = stamp_buffer(3 * (size_t)g.x + 600);       // [4 g] stamps, then [8 g] phase sums, then [2400] the trace of a few tiles (GRAN)
  r.phases = r.stamps ? r.stamps + 4 * (size_t)g.x : nullptr;
  if (res_time_it) CICE_HIP(hipEventRecord(res_t0, stream));
  switch (W) {
    case 4: launch_res<4>(r, damp, peer, g, stream); break;
    case 6: launch_res<6>(r, damp, peer, g, stream); break;
    case 8: launch_res<8>(r, damp, peer, g, stream); break;
    case 11: launch_res<11>(r, damp, peer, g, stream); break;
    case 12: launch_res<12>(r, damp, peer, g, stream); break;
    default: throw Error{CICE_EINVAL, "resident_waves must be 4, 6, 8, 11 or 12"};
  }
  if (res_time_it) {
    CICE_HIP(hipEventRecord(res_t1, stream));
    res_timed = true;     // (a loop that gives up is run again by the other loops: subcycles() then reports its own bracket)
  }
  if (const hipError_t le = hipGetLastError(); le != hipSuccess) {   // nothing ran: the other loops take the range
    std::fprintf(stderr, "cice4_amd: resident EVP loop could not be launched (%s); this range runs as one launch per pair "
                         "of subcycles and so do later ones\n", hipGetErrorString(le));
    CICE_REQUIRE(!peer, "resident EVP loop across ranks: launch failed on this rank");   // the others would wait for us
    resident_failed = true;
    return false;
  }
  unsigned aborted = 0;
  if (peer) {
    // This loop ends when the neighbours' loops have run: nothing here may block inside the runtime before they are
    // launched.  Ranks that are threads of ONE process (the one-GPU tests) share the runtime's locks, and a blocking copy
    // or synchronisation of this thread could hold up the launch of the very kernel it is waiting for: poll an event.
    if (!res_done_ev) CICE_HIP(hipEventCreateWithFlags(&res_done_ev, hipEventDisableTiming));
    CICE_HIP(hipEventRecord(res_done_ev, stream));
    hipError_t q;
    while ((q = hipEventQuery(res_done_ev)) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(20));
    CICE_HIP(q);
  }
  if (peer && res_peer_agree) halo.all_max_u32(r.abort_flag);   // every rank falls back, or none does
  // one read-back per loop: the abort word, who gave up on what, and (word 7) the ice cover k_res_choose_map counted --
  // into page-locked memory (a copy into pageable memory is staged and synchronous: ~15 us of a 600-us loop)
  if (!res_why) CICE_HIP(hipHostMalloc((void**)&res_why, 8 * sizeof(unsigned), hipHostMallocDefault));
  unsigned* why = res_why;
  CICE_HIP(hipMemcpyAsync(why, r.abort_flag, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  CICE_HIP(hipStreamSynchronize(stream));
  const int cover[3] = {0, (int)(why[7] >> 16), (int)(why[7] & 0xffffu)};
  if (r.tile_map && cover[2] > 0) {
    // Which shape the NEXT call takes (granules_on): under an ice cover that leaves most tiles empty three barrier-coupled
    // workgroups per CU win -- a CU then holds one tile with ice and two without (gx1 size, polar caps: 231 k subcycles/s
    // against 205 k for the granule loop; fully covered 186 k against 200 k).  The cover of one step is the cover of the
    // next to within a few cells; hysteresis keeps the tables from being rebuilt back and forth.
    const int pct = (int)(100LL * cover[1] / cover[2]);
    if (pct < 50) res_sparse = true;
    else if (pct > 60) res_sparse = false;
  }
  aborted = why[0];
  res_epoch += (unsigned)nsub + (peer ? 3u : 0u);
  if (aborted && why[1]) {   // this rank's own first time-out (none: the word came from another rank)
    std::fprintf(stderr, "cice4_amd: rank %d: tile %u of %d gave up %s (subcycle %u of the launch, word wanted %u), producers "
                         "not heard from: lanes %08x%08x of its dependency list:", dom.rank, why[2], res_tiles,
                 why[1] == 1 ? "waiting for the neighbouring ranks' loops to begin" : why[1] == 4 ? "polling the granules of its halo" : "waiting for the producers of its halo",
                 why[3], why[6], why[5], why[4]);
    std::vector<int32_t> dl(RES_MAXDEP);
    CICE_HIP(hipMemcpy(dl.data(), res_deps.p + (size_t)why[2] * RES_MAXDEP, RES_MAXDEP * 4, hipMemcpyDeviceToHost));
    const unsigned long long ms = ((unsigned long long)why[5] << 32) | why[4];
    for (int l = 0; l < RES_MAXDEP; ++l)
      if ((ms >> l) & 1) {
        unsigned seen = 0;
        const int d = dl[l];
        const unsigned* wp = d <= -2 ? res_rprog.p + (size_t)(-2 - d) * RES_STRIDE : res_prog.p + (size_t)d * RES_STRIDE;
        CICE_HIP(hipMemcpy(&seen, wp, 4, hipMemcpyDeviceToHost));
        if (d <= -2) {
          const std::vector<int> pr = peer_ranks();
          const int sd = (-2 - d) / RP_MAX;
          std::fprintf(stderr, " [%d: rank %d's tile %d, word now %u]", l, sd < (int)pr.size() ? pr[(size_t)sd] : -1, (-2 - d) % RP_MAX, seen);
        } else {
          std::fprintf(stderr, " [%d: own tile %d, word now %u]", l, d, seen);
        }
      }
    std::fprintf(stderr, "\n");
  }
  if (aborted) {
    std::fprintf(stderr, "cice4_amd: resident EVP loop timed out (not every tile was resident%s); this range runs as one "
                         "launch per pair of subcycles%s\n", peer ? ", here or on another rank" : "",
                 dense ? ", later ones with one workgroup per CU" : " and so do later ones");
    if (dense) res_level = 1;   // not every slot of the chip was free: one workgroup per CU from now on
    else resident_failed = true;
    // whoever held the slots may be gone later: look again after res_retry_steps calls (across ranks only where the
    // ranks agreed on the time-out, so that they also agree on the retry)
    res_retry_in = (!peer || res_peer_agree) ? res_retry_steps : 0;
    res_prog.zero(stream);
    if (res_prog2.p) res_prog2.zero(stream);
    if (!peer && res_xg.p) res_xg.zero(stream);
    if (!peer && res_xgr.p) res_xgr.zero(stream);
    if (!peer) res_epoch = 0;   // (across ranks the neighbours hold words about us: the epoch only ever grows)
    return false;
  }
  cur = 1 - cur;   // the result is in the other copy whatever the parity of nsub
  ++flips;
  copies_identical = false;
  return true;
}

// subcycles ksub0 .. ksub0+nsub-1: K per sweep where possible, else pairs, else one by one
void Evp::launch_range(int ksub0, int nsub) {
  const bool fuse = can_fuse();
  const bool skew = can_skew(), skew_fold = !skew && can_skew_fold();
  const int K = skew_levels();
  const bool trim = skew && can_trim(), split = trim && can_split();
  const bool pairs = skew && !trim && split_probe == 0 && pairs_ok();
  const int end = ksub0 + nsub - 1;
  for (int k = ksub0; k <= end;) {
    // a wide-halo refresh falls after subcycles that are multiples of `overlap`: a launch must not straddle one,
    // i.e. none of its subcycles but the last may be such a multiple
    auto clear = [&](int len) {
      if (k + len - 1 > end) return false;
      if (dom.overlap > 0)
        for (int q = k; q < k + len - 1; ++q)
          if (q % dom.overlap == 0) return false;
      return true;
    };
    if (skew && clear(K)) {
      const int kend = k + K - 1;
      if (pairs) to_pairs();
      if (split_probe > 0 && !in_capture && dom.overlap == 0 && dom.nblocks() == 1 && !halo.multi_rank()) {
        if (!stream2) {
          (void)tiles_for(K, 0, true);
          CICE_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
          CICE_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
          CICE_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        launch_subcycle_skew_split(k, K);    // timing only: the extension rows are not computed and nobody refreshes them
      } else if (trim) {
        // subcycles left until the next refresh = extension rows that still have to be right after this sweep
        const int next = std::min(((kend + dom.overlap - 1) / dom.overlap) * dom.overlap, sc.ndte);
        const int ext = std::min(dom.overlap, next - kend);
        if (ext == 0 && split) launch_subcycle_skew_split(k, K);   // a refresh follows: edge segments first
        else launch_subcycle_skew_ext(k, K, ext);
      } else {
        launch_subcycle_skew(k, K);
      }
      k += K;
    } else if (skew_fold && clear(K)) {
      launch_subcycle_skew_fold(k, K);
      k += K;
    } else if (fuse && clear(2)) {
      to_planes();
      launch_subcycle_pair(k);
      k += 2;
    } else {
      to_planes();
      launch_subcycle(k);
      k += 1;
    }
  }
  to_planes();   // whoever comes next (finish, download, another range) finds the state where it always was
}

void Evp::subcycles(int ksub0, int nsub, float* elapsed_ms) {
  CICE_REQUIRE(prepared, "cice_evp_subcycles before cice_evp_prepare");
  CICE_REQUIRE(ksub0 >= 1 && nsub >= 0, "bad subcycle range");
  // elapsed_ms: events that live as long as the object (creating and destroying a pair per call, and four records instead of
  // two, cost a step of the one-launch loop 20 us of its 600: scripts/step_overhead.py).  The one-launch loop is timed around
  // its ONE launch -- what rocprofv3 reports for the kernel -- not around the small launch that chooses the tile map before it
  // and the read-back of the abort word behind it (run_resident); every other form of the range around all its launches.
  hipEvent_t &e0 = sub_t0, &e1 = sub_t1;
  const bool try_resident = nsub >= 2 && (can_reside() || can_reside_peer());
  if (elapsed_ms) {
    for (hipEvent_t* e : {&sub_t0, &sub_t1, &res_t0, &res_t1})
      if (!*e) CICE_HIP(hipEventCreate(e));
    if (!try_resident) CICE_HIP(hipEventRecord(e0, stream));
  }
  res_time_it = elapsed_ms != nullptr;
  res_timed = false;
  // Single-rank domains: the whole loop is captured once and replayed.  Multi-rank domains launch eagerly
  // by default: capturing the grouped ncclSend/ncclRecv calls works on this ROCm (scripts/rccl_graph_probe.cpp,
  // and the 1-rank self-communicator test), but has never run between real peers, where a mis-ordered replay
  // would hang instead of raising an error.  Opt in with cice_evp_set_option("comm_graph", 1) or
  // CICE4_AMD_COMM_GRAPH=1 once a multi-GPU parity run has passed.
  static const bool env_comm_graph = std::getenv("CICE4_AMD_COMM_GRAPH") != nullptr;
  bool graph_ok = use_graph && nsub > 1 && (!halo.multi_rank() || comm_graph || env_comm_graph);
  if (!can_skew() && can_skew_fold()) ensure_band(skew_levels());
  bool tuning = false;
  if ((can_skew() || can_skew_fold()) && skew_rows_on()) {   // the segment table of the sweep kernel, outside any capture
    const int K = skew_levels(), seg = skew_seg_rows(K);
    const int tiles_x = skew_strips(K, nullptr), tiles_y = ((dom.ny_block - 2) + seg - 1) / seg;
    build_skew_rows(K, tiles_x, tiles_y, dom.nblocks(), seg);
    if (balance_on() && !(can_skew() && can_trim()) && nsub >= K && !(nsub >= 2 && (can_reside() || can_reside_peer()))) {
      // this loop's sweeps are measured (eagerly: no graph) while a tuning phase lasts; a new phase every bal_every loops
      if (bal_left == 0 && ++bal_since >= bal_every) {
        bal_left = BAL_AGAIN;
        bal_since = 0;
      }
      const size_t want = 2 * (size_t)(8 * ((std::max(tiles_x * tiles_y * dom.nblocks(), bal_slots) + 7) / 8));
      if (bal_left > 0 && skew_dbg.n < want) skew_dbg.alloc(want);
      tuning = bal_left > 0;
    }
  }
  if (pairs_ok())
    for (int k = 0; k < 2; ++k)
      if (st2[k].n < 14 * n) st2[k].alloc(14 * n);
  if ((can_skew() || can_skew_fold()) && !skew_packed) skew_pack();   // (sweeps switched on after prepare(): allocations outside any capture)
  if (can_skew() && can_trim()) build_split(skew_levels());   // (uploads tables: outside any capture)
  if (tuning) graph_ok = false;
  bool replayed = false;
  loop_launches = 0;
  if (try_resident) {
    replayed = run_resident(ksub0, nsub);
    if (replayed) loop_launches = 1;
    else res_timed = false;
    if (!replayed && elapsed_ms) CICE_HIP(hipEventRecord(e0, stream));   // (the loop gave up: the other loops run the range)
  }
  res_time_it = false;
  if (!replayed && graph_ok) {
    const int key[4] = {cur, ksub0, nsub,
                        ((((waves * 100 + rows_per_wave) * 2 + (derive_on ? 1 : 0)) * 64 + (fuse_on ? 32 : 0) + waves2) * 16 +
                         (can_skew() || can_skew_fold() ? skew_levels() : 0)) * 128 + (halo.generation() & 63) * 2 +
                            (copies_identical ? 1 : 0)};   // (set_option drops the graph anyway)
    const int cur0 = cur;
    if (!graph_exec || std::memcmp(key, graph_key, sizeof(key)) != 0) {
      drop_graph();
      hipGraph_t gph = nullptr;
      bool capturing = false;
      try {
        CICE_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        capturing = true;
        in_capture = true;
        flips = 0;
        loop_launches = 0;
        launch_range(ksub0, nsub);
        graph_flips = flips;
        graph_launches = loop_launches;
        capturing = false;
        in_capture = false;
        CICE_HIP(hipStreamEndCapture(stream, &gph));
        CICE_HIP(hipGraphInstantiate(&graph_exec, gph, nullptr, nullptr, 0));
        CICE_HIP(hipGraphDestroy(gph));
        std::memcpy(graph_key, key, sizeof(key));
      } catch (const Error&) {
        // capture not possible here: close it, forget graphs for this context, run eagerly
        in_capture = false;
        if (capturing) (void)hipStreamEndCapture(stream, &gph);
        if (gph) (void)hipGraphDestroy(gph);
        (void)hipGetLastError();
        drop_graph();
        use_graph = false;
      }
      cur = cur0;
    }
    if (graph_exec) {
      CICE_HIP(hipGraphLaunch(graph_exec, stream));
      if (graph_flips & 1) cur = 1 - cur;
      replayed = true;
      loop_launches = graph_launches;
    }
  }
  if (!replayed) launch_range(ksub0, nsub);
  CICE_HIP(hipGetLastError());
  last_launches = loop_launches;
  {   // CICE4_AMD_STATS=1: how the first ranges of subcycles ran (the whole-model tests read it from the model's log)
    static const bool stats = [] { const char* e = std::getenv("CICE4_AMD_STATS"); return e && e[0] == '1'; }();
    if (stats && stats_left > 0) {
      --stats_left;
      std::fprintf(stderr, "cice4_amd: subcycles %d..%d of evp(dt) on %d block(s): %d kernel launch(es)%s\n", ksub0, ksub0 + nsub - 1,
                   dom.nblocks(), last_launches, last_launches == 1 && nsub > 1 ? " (the whole loop in one launch, state in registers)" : "");
    }
  }
  if (elapsed_ms) {
    if (res_timed) {
      CICE_HIP(hipEventSynchronize(res_t1));
      CICE_HIP(hipEventElapsedTime(elapsed_ms, res_t0, res_t1));
    } else {
      CICE_HIP(hipEventRecord(e1, stream));
      CICE_HIP(hipEventSynchronize(e1));
      CICE_HIP(hipEventElapsedTime(elapsed_ms, e0, e1));
    }
  }
}

void Evp::finish() {
  CICE_REQUIRE(prepared, "cice_evp_finish before cice_evp_prepare");
  PrepArgs a{};
  a.nx = dom.nx_block; a.ny = dom.ny_block; a.nblocks = dom.nblocks(); a.n = n; a.blk = blk.p;
  a.iceumask = iceumask.p; a.uocn = uocn.p; a.vocn = vocn.p; a.u = uv[cur].p; a.v = uv[cur].p + n;
  a.aiu = aiu.p; a.strocnx = strocnx.p; a.strocny = strocny.p; a.strocnxT = strocnxT.p;
  a.strocnyT = strocnyT.p; a.tarea = tarea.p; a.uarea = uarea.p; a.fm = fm.p;
  const dim3 g = grid1(n), blk256(256);
  hipLaunchKernelGGL(k_finish, g, blk256, 0, stream, a);  // :410-425
  // u2tgrid_vector :427-428 = copy, halo update (NE corner), to_tgrid
  CICE_HIP(hipMemcpyAsync(work1.p, strocnxT.p, n * 8, hipMemcpyDeviceToDevice, stream));
  CICE_HIP(hipMemcpyAsync(work1.p + n, strocnyT.p, n * 8, hipMemcpyDeviceToDevice, stream));
  halo.update_r8(work1.p, 2, n, true, LOC_NECORNER, KIND_VECTOR);             // ice_grid.F90:1669
  hipLaunchKernelGGL(k_to_tgrid2, g, blk256, 0, stream, a, (const double*)work1.p,
                     (const double*)(work1.p + n), strocnxT.p, strocnyT.p);
  CICE_HIP(hipGetLastError());
  CICE_HIP(hipStreamSynchronize(stream));
}

// ---- per-routine host-pointer entries ---------------------------------------------------------
void Evp::stress_host(hipStream_t s, double dt, int ndte, int damping, int nx, int ny, int ksub,
                      int icellt, const int32_t* ti, const int32_t* tj, const double* uvel,
                      const double* vvel, const double* const grid10[10], const double* strength,
                      double* const sg[12], double* const diag[5], double* str) {
  const size_t np = (size_t)nx * ny;
  CICE_REQUIRE(icellt >= 0 && (size_t)icellt <= np, "icellt out of range");
  for (int e = 0; e < icellt; ++e)
    CICE_REQUIRE(ti[e] >= 2 && ti[e] <= nx && tj[e] >= 2 && tj[e] <= ny, "stress: index outside block");
  DevBuf<double> d;       // uvel,vvel,10 grid,strength,12 sig,5 diag,8 str = 38 planes
  d.alloc(38 * np);
  DevBuf<int32_t> li;
  li.alloc(2 * np);
  auto up = [&](int plane, const double* h) {
    CICE_HIP(hipMemcpyAsync(d.p + plane * np, h, np * 8, hipMemcpyHostToDevice, s));
  };
  up(0, uvel); up(1, vvel);
  for (int k = 0; k < 10; ++k) up(2 + k, grid10[k]);
  up(12, strength);
  for (int k = 0; k < 12; ++k) up(13 + k, sg[k]);
  for (int k = 0; k < 5; ++k) up(25 + k, diag[k]);
  CICE_HIP(hipMemsetAsync(d.p + 30 * np, 0, 8 * np * 8, s));  // str(:,:,:) = c0, :1051
  if (icellt) {
    CICE_HIP(hipMemcpyAsync(li.p, ti, (size_t)icellt * 4, hipMemcpyHostToDevice, s));
    CICE_HIP(hipMemcpyAsync(li.p + np, tj, (size_t)icellt * 4, hipMemcpyHostToDevice, s));
  }
  StressListArgs a{};
  a.sc.set(dt, ndte, damping);
  a.nx = nx; a.ny = ny; a.ksub = ksub; a.icellt = icellt; a.ti = li.p; a.tj = li.p + np;
  a.uvel = d.p; a.vvel = d.p + np;
  for (int k = 0; k < 10; ++k) a.g[k] = d.p + (2 + k) * np;
  a.strength = d.p + 12 * np;
  for (int k = 0; k < 12; ++k) a.sig[k] = d.p + (13 + k) * np;
  for (int k = 0; k < 5; ++k) a.diag[k] = d.p + (25 + k) * np;
  a.str = d.p + 30 * np;
  if (icellt) {
    const dim3 g = grid1(icellt);
    const bool last = ksub == ndte;
    if (last) {
      if (damping) hipLaunchKernelGGL((k_stress_list<true, true>), g, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_stress_list<true, false>), g, dim3(256), 0, s, a);
    } else {
      if (damping) hipLaunchKernelGGL((k_stress_list<false, true>), g, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_stress_list<false, false>), g, dim3(256), 0, s, a);
    }
    CICE_HIP(hipGetLastError());
  }
  auto down = [&](int plane, double* h, size_t planes = 1) {
    CICE_HIP(hipMemcpyAsync(h, d.p + plane * np, planes * np * 8, hipMemcpyDeviceToHost, s));
  };
  for (int k = 0; k < 12; ++k) down(13 + k, sg[k]);
  for (int k = 0; k < 5; ++k) down(25 + k, diag[k]);
  down(30, str, 8);
  CICE_HIP(hipStreamSynchronize(s));
}

void Evp::stepu_host(hipStream_t s, int nx, int ny, int icellu, const int32_t* ui, const int32_t* uj,
                     const double* const in10[10], const double* str, double* const io6[6]) {
  const size_t np = (size_t)nx * ny;
  CICE_REQUIRE(icellu >= 0 && (size_t)icellu <= np, "icellu out of range");
  for (int e = 0; e < icellu; ++e)
    CICE_REQUIRE(ui[e] >= 1 && ui[e] < nx && uj[e] >= 1 && uj[e] < ny, "stepu: index outside block");
  DevBuf<double> d;  // 10 in, 8 str, 6 io
  d.alloc(24 * np);
  DevBuf<int32_t> li;
  li.alloc(2 * np);
  for (int k = 0; k < 10; ++k)
    CICE_HIP(hipMemcpyAsync(d.p + k * np, in10[k], np * 8, hipMemcpyHostToDevice, s));
  CICE_HIP(hipMemcpyAsync(d.p + 10 * np, str, 8 * np * 8, hipMemcpyHostToDevice, s));
  for (int k = 0; k < 6; ++k)
    CICE_HIP(hipMemcpyAsync(d.p + (18 + k) * np, io6[k], np * 8, hipMemcpyHostToDevice, s));
  if (icellu) {
    CICE_HIP(hipMemcpyAsync(li.p, ui, (size_t)icellu * 4, hipMemcpyHostToDevice, s));
    CICE_HIP(hipMemcpyAsync(li.p + np, uj, (size_t)icellu * 4, hipMemcpyHostToDevice, s));
  }
  StepuListArgs a{};
  a.nx = nx; a.ny = ny; a.icellu = icellu; a.ui = li.p; a.uj = li.p + np;
  for (int k = 0; k < 10; ++k) a.in[k] = d.p + k * np;
  a.str = d.p + 10 * np;
  for (int k = 0; k < 6; ++k) a.io[k] = d.p + (18 + k) * np;
  if (icellu) {
    hipLaunchKernelGGL(k_stepu_list, grid1(icellu), dim3(256), 0, s, a);
    CICE_HIP(hipGetLastError());
  }
  for (int k = 0; k < 6; ++k)
    CICE_HIP(hipMemcpyAsync(io6[k], d.p + (18 + k) * np, np * 8, hipMemcpyDeviceToHost, s));
  CICE_HIP(hipStreamSynchronize(s));
}

}  // namespace cice
