// Column thermodynamics on the device.  Behavioural source: source/ice_therm_vertical.F90
// of the reference (file:line cited).  One lane = one (cell, category) column: the whole
// column (4 ice layers + 1 snow layer, the 6-row tridiagonal system, the interface
// conductances) lives in registers; the reference's per-iteration allocate/deallocate and
// list re-compaction (:1622-1650, :2077-2088) become a per-lane `converged` flag inside the
// iteration loop.  Fields are read and written in the reference's own layout, so lanes of a
// wavefront touch 64 consecutive i of one (j, level) row.
//
// Configuration covered: heat_capacity = T, calc_Tsfc = T or F (input_templates/gx3/ice_in and
// the COSIMA configurations); conduct = 'MU71' or 'bubbly'.
#include "therm.h"

#include <cmath>

namespace cice {

using namespace K;

void ThermoParams::init(const cice_thermo_config& c) {  // init_thermo_vertical :533-584
  constexpr double nsal = 0.407, msal = 0.573, min_salin = 0.1;
  heat_capacity = c.heat_capacity != 0;
  calc_Tsfc = c.calc_Tsfc != 0;
  conduct = c.conduct;
  ustar_min = c.ustar_min;
  tr_iage = c.tr_iage; nt_Tsfc = c.nt_Tsfc; nt_iage = c.nt_iage;
  l_brine = (saltmax > min_salin && heat_capacity) ? 1 : 0;
  for (int k = 1; k <= NILYR; ++k) {
    if (l_brine) {
      const double zn = ((double)k - p5) / (double)NILYR;
      salin[k - 1] = (saltmax / c2) * (c1 - std::cos(pi * std::pow(zn, nsal / (msal + zn))));
    } else {
      salin[k - 1] = c0;
    }
    Tmlt[k - 1] = -salin[k - 1] * depressT;
  }
  salin[NILYR] = l_brine ? saltmax : c0;
  Tmlt[NILYR] = -salin[NILYR] * depressT;
}

namespace {

constexpr int NI = NILYR, NS = NSLYR, NMAT = NI + NS + 1;

// failure stages in the order the reference meets them inside one thermo_vertical call
enum : unsigned {
  ST_TSN_HIGH = 1,        // :1025-1054
  ST_TSN_LOW = 2,         // :1056-1078
  ST_TIN_BASE = 3,        // +2k: Tin > Tmax in layer k (:1144), +2k+1: Tin < Tmin (:1170)
  ST_NOCONV = 3 + 2 * NI, // :2092-2130
  ST_ECONS = 4 + 2 * NI   // :4573-4610
};

struct Col {
  double hin, hsn, hilyr, hslyr, Tsf, einit, efinal, fcondbot, hsn_new;
  double qin[NI], Tin[NI], qsn[NS], Tsn[NS];
};

struct Flx {
  double rhoa, flw, potT, Qa, shcoef, lhcoef, Tbot;
  double fswsfc, fswint, fswthrun, Sswabs[NS], Iswabs[NI];
  double fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn;
};

struct Gro {
  double fbot, fsnow, fhocnn, evapn, meltt, melts, meltb, congel, snoice, mlt_onset, frz_onset;
};

// init_vertical_profile :955-1209
__device__ __forceinline__ unsigned init_profile(const ThermoParams& P, double aicen, double vicen,
                                                 double vsnon, double Tsfcn, const double* ei,
                                                 const double* es, Col& c) {
  constexpr double Tmin = -100.0, rnslyr = (double)NS;
  unsigned stage = 0;
  c.einit = c0;
  c.Tsf = Tsfcn;
  c.hin = vicen / aicen;
  c.hsn = vsnon / aicen;
  c.hilyr = c.hin / (double)NI;
  c.hslyr = c.hsn / rnslyr;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double Tmax;
    if (c.hslyr > hs_min / rnslyr) {
      c.qsn[k] = es[k] * rnslyr / vsnon;
      Tmax = -c.qsn[k] * puny * rnslyr / (rhos * cp_ice * vsnon);
    } else {
      c.qsn[k] = -rhos * Lfresh;
      Tmax = puny;
    }
    c.Tsn[k] = (Lfresh + c.qsn[k] / rhos) / cp_ice;
    if (c.Tsn[k] > Tmax) {
      if (stage == 0 || stage > ST_TSN_HIGH) stage = ST_TSN_HIGH;
    } else if (c.Tsn[k] < Tmin) {
      if (stage == 0) stage = ST_TSN_LOW;
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    if (c.Tsn[k] > c0) {
      c.Tsn[k] = c0;
      c.qsn[k] = -rhos * Lfresh;
    }
    c.einit = c.einit + c.hslyr * c.qsn[k];
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    double Tmax;
    c.qin[k] = ei[k] * (double)NI / vicen;
    if (P.l_brine) {  // calculate_Tin_from_qin :1249-1258
      const double aa1 = cp_ice;
      const double bb1 = (cp_ocn - cp_ice) * P.Tmlt[k] - c.qin[k] / rhoi - Lfresh;
      const double cc1 = Lfresh * P.Tmlt[k];
      c.Tin[k] = (-bb1 - sqrt(bb1 * bb1 - c4 * aa1 * cc1)) / (c2 * aa1);
      Tmax = P.Tmlt[k];
    } else {
      c.Tin[k] = (Lfresh + c.qin[k] / rhoi) / cp_ice;
      Tmax = -c.qin[k] * puny / (rhos * cp_ice * vicen);
    }
    if (c.Tin[k] > Tmax) {
      if (stage == 0) stage = ST_TIN_BASE + 2 * k;
    } else if (c.Tin[k] < Tmin) {
      if (stage == 0) stage = ST_TIN_BASE + 2 * k + 1;
    }
    if (c.Tin[k] > c0) {
      c.Tin[k] = c0;
      c.qin[k] = -rhoi * Lfresh;
    }
    c.einit = c.einit + c.hilyr * c.qin[k];
  }
  return stage;
}

// temperature_changes :1288-2148 (+ conductivity :2169, surface_fluxes :2314,
// get_matrix_elements_calc_Tsfc :2447 or, CALC = false (calc_Tsfc = F: fsurfn, fcondtopn, flatn
// are the caller's, Tsf is not solved for), get_matrix_elements_know_Tsfc :2777;
// tridiag_solver :3069)
template <bool CALC>
__device__ __forceinline__ bool temperature_changes(const ThermoParams& P, double dt, Col& c, Flx& f, int& iters) {
  constexpr int nitermax = 100;
  constexpr double Tsf_errmax = 5.0e-4;
  const double hilyr = c.hilyr, hslyr = c.hslyr;
  bool converged = false, l_snow = false, l_cold = true;
  double dTi1_prev = c0;
  double dTsf_prev = c0, dfsens_dT = c0, dflat_dT = c0, dflwout_dT = c0;
  double Tin_init[NI], Tin_start[NI], Tsn_init[NS], Tsn_start[NS], etas[NS], kh[NMAT];
  const double dt_rhoi_hlyr = dt / (rhoi * hilyr);
  c.fcondbot = c0;
  if (hslyr > hs_min / (double)NS) l_snow = true;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    Tsn_init[k] = Tsn_start[k] = c.Tsn[k];
    etas[k] = l_snow ? dt / (rhos * cp_ice * hslyr) : c0;
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) Tin_init[k] = Tin_start[k] = c.Tin[k];
  {  // conductivity :2221-2293
    double kilyr[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (P.conduct == 0)
        kilyr[k] = kice + betak * P.salin[k] / fmin(-puny, c.Tin[k]);
      else
        kilyr[k] = (2.11 - 0.011 * c.Tin[k] + 0.09 * P.salin[k] / fmin(-puny, c.Tin[k])) * rhoi / 917.0;
      kilyr[k] = fmax(kilyr[k], kimin);
    }
    if (l_snow) {
      kh[0] = c2 * ksno / hslyr;
      kh[NS] = c2 * ksno * kilyr[0] / (ksno * hilyr + kilyr[0] * hslyr);
    } else {
      kh[0] = c0;
      kh[NS] = c2 * kilyr[0] / hilyr;
    }
    kh[NS + NI] = c2 * kilyr[NI - 1] / hilyr;
#pragma unroll
    for (int k = 2; k <= NS; ++k) kh[k - 1] = l_snow ? c2 * ksno * ksno / ((ksno + ksno) * hslyr) : c0;
#pragma unroll
    for (int k = 2; k <= NI; ++k)
      kh[k + NS - 1] = c2 * kilyr[k - 2] * kilyr[k - 1] / ((kilyr[k - 2] + kilyr[k - 1]) * hilyr);
  }
  {  // SW overshoot limiter :1541-1596
    constexpr double frac = 0.9, dTemp = 0.02;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      double Iswabs_tmp = c0;
      if (Tin_init[k] <= P.Tmlt[k] - dTemp) {
        if (P.l_brine) {
          const double ci = cp_ice - Lfresh * P.Tmlt[k] / (Tin_init[k] * Tin_init[k]);
          Iswabs_tmp = fmin(f.Iswabs[k], frac * (P.Tmlt[k] - Tin_init[k]) * ci / dt_rhoi_hlyr);
        } else {
          Iswabs_tmp = fmin(f.Iswabs[k], frac * (-Tin_init[k]) * cp_ice / dt_rhoi_hlyr);
        }
      }
      if (Iswabs_tmp < puny) Iswabs_tmp = c0;
      const double dswabs = fmin(f.Iswabs[k] - Iswabs_tmp, f.fswint);
      f.fswsfc = f.fswsfc + dswabs;
      f.fswint = f.fswint - dswabs;
      f.Iswabs[k] = Iswabs_tmp;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (l_snow) {
        double Sswabs_tmp = c0;
        if (Tsn_init[k] <= -dTemp) Sswabs_tmp = fmin(f.Sswabs[k], -frac * Tsn_init[k] / etas[k]);
        if (f.Sswabs[k] < puny) Sswabs_tmp = c0;
        const double dswabs = fmin(f.Sswabs[k] - Sswabs_tmp, f.fswint);
        f.fswsfc = f.fswsfc + dswabs;
        f.fswint = f.fswint - dswabs;
        f.Sswabs[k] = Sswabs_tmp;
      }
  }
  f.fswabsn = f.fswsfc + f.fswint + f.fswthrun;  // :1605

#pragma unroll 1
  for (int niter = 1; niter <= nitermax && !converged; ++niter) {
    iters = niter;
    double etai[NI], sb[NMAT], dg[NMAT], sp[NMAT], rh[NMAT], Tm[NMAT];
    double dfsurf_dT = c0, avg_Tsi = c0, enew = c0, Tsf_start = c0, dTsf = c0, avg_Tsf = c0;
    double dqmat[NI];
    bool reduce_kh[NI];
    converged = true;
#pragma unroll
    for (int k = 0; k < NI; ++k) {  // :1669-1684
      const double ci = P.l_brine ? cp_ice - Lfresh * P.Tmlt[k] / (c.Tin[k] * Tin_init[k]) : cp_ice;
      etai[k] = dt_rhoi_hlyr / ci;
    }
    if (CALC) {  // surface_fluxes :2389-2421
      const double TsfK = c.Tsf + Tffresh;
      const double tmpvar = c1 / TsfK;
      const double qsat = qqqice * exp_libm(-TTTice * tmpvar);
      const double Qsfc = qsat / f.rhoa;
      const double dQsfcdT = TTTice * tmpvar * tmpvar * Qsfc;
      const double flwdabs = emissivity * f.flw;
      // TsfK**4, TsfK**3 evaluated left to right as the oracle's compiler does
      const double T3 = (TsfK * TsfK) * TsfK;
      f.flwoutn = -emissivity * stefan_boltzmann * (T3 * TsfK);
      f.fsensn = f.shcoef * (f.potT - TsfK);
      f.flatn = f.lhcoef * (f.Qa - Qsfc);
      dflwout_dT = -emissivity * stefan_boltzmann * c4 * T3;
      dfsens_dT = -f.shcoef;
      dflat_dT = -f.lhcoef * dQsfcdT;
      f.fsurfn = f.fswsfc + flwdabs + f.flwoutn + f.fsensn + f.flatn;
      dfsurf_dT = dflwout_dT + dfsens_dT + dflat_dT;
      // :1719-1738
      f.fcondtopn = l_snow ? kh[0] * (c.Tsf - c.Tsn[0]) : kh[NS] * (c.Tsf - c.Tin[0]);
      if (f.fsurfn < f.fcondtopn) c.Tsf = fmin(c.Tsf, -puny);
      Tsf_start = c.Tsf;
      l_cold = (c.Tsf <= -puny);
    }
    // get_matrix_elements_calc_Tsfc :2540-2751 / _know_Tsfc :2871-3048 (0-based rows)
#pragma unroll
    for (int k = 0; k <= NS; ++k) {
      sb[k] = c0; dg[k] = c1; sp[k] = c0; rh[k] = c0;
    }
    if (!CALC) {
      if (l_snow) {  // :2892-2902
        sb[1] = c0;
        sp[1] = -etas[0] * kh[1];
        dg[1] = c1 + etas[0] * kh[1];
        rh[1] = Tsn_init[0] + etas[0] * f.Sswabs[0] + etas[0] * f.fcondtopn;
      }
    } else if (l_cold) {
      if (l_snow) {
        sb[0] = c0; dg[0] = dfsurf_dT - kh[0]; sp[0] = kh[0]; rh[0] = dfsurf_dT * c.Tsf - f.fsurfn;
      } else {
        sb[NS] = c0; dg[NS] = dfsurf_dT - kh[NS]; sp[NS] = kh[NS];
        rh[NS] = dfsurf_dT * c.Tsf - f.fsurfn;
      }
    }
    if (CALC && l_snow) {
      if (l_cold) {
        sb[1] = -etas[0] * kh[0];
        sp[1] = -etas[0] * kh[1];
        dg[1] = c1 + etas[0] * (kh[0] + kh[1]);
        rh[1] = Tsn_init[0] + etas[0] * f.Sswabs[0];
      } else {
        sb[1] = c0;
        sp[1] = -etas[0] * kh[1];
        dg[1] = c1 + etas[0] * (kh[0] + kh[1]);
        rh[1] = Tsn_init[0] + etas[0] * kh[0] * c.Tsf + etas[0] * f.Sswabs[0];
      }
    }
#pragma unroll
    for (int k = 2; k <= NS; ++k)
      if (l_snow) {
        sb[k] = -etas[k - 1] * kh[k - 1];
        sp[k] = -etas[k - 1] * kh[k];
        dg[k] = c1 + etas[k - 1] * (kh[k - 1] + kh[k]);
        rh[k] = Tsn_init[k - 1] + etas[k - 1] * f.Sswabs[k - 1];
      }
    {  // top ice layer
      constexpr int k = NS, kr = NS + 1;
      if (!CALC && !l_snow) {  // :2956-2962
        sb[kr] = c0;
        sp[kr] = -etai[0] * kh[k + 1];
        dg[kr] = c1 + etai[0] * kh[k + 1];
        rh[kr] = Tin_init[0] + etai[0] * f.Iswabs[0] + etai[0] * f.fcondtopn;
      } else if (!CALC || l_snow || l_cold) {
        sb[kr] = -etai[0] * kh[k];
        sp[kr] = -etai[0] * kh[k + 1];
        dg[kr] = c1 + etai[0] * (kh[k] + kh[k + 1]);
        rh[kr] = Tin_init[0] + etai[0] * f.Iswabs[0];
      } else {
        sb[kr] = c0;
        sp[kr] = -etai[0] * kh[k + 1];
        dg[kr] = c1 + etai[0] * (kh[k] + kh[k + 1]);
        rh[kr] = Tin_init[0] + etai[0] * f.Iswabs[0] + etai[0] * kh[k] * c.Tsf;
      }
    }
    {  // bottom ice layer
      constexpr int ki = NI - 1, k = NI - 1 + NS, kr = k + 1;
      sb[kr] = -etai[ki] * kh[k];
      sp[kr] = c0;
      dg[kr] = c1 + etai[ki] * (kh[k] + kh[k + 1]);
      rh[kr] = Tin_init[ki] + etai[ki] * f.Iswabs[ki] + etai[ki] * kh[k + 1] * f.Tbot;
    }
#pragma unroll
    for (int ki = 1; ki < NI - 1; ++ki) {
      const int k = ki + NS, kr = k + 1;
      sb[kr] = -etai[ki] * kh[k];
      sp[kr] = -etai[ki] * kh[k + 1];
      dg[kr] = c1 + etai[ki] * (kh[k] + kh[k + 1]);
      rh[kr] = Tin_init[ki] + etai[ki] * f.Iswabs[ki];
    }
    {  // tridiag_solver :3119-3143
      double wg[NMAT], wbeta = dg[0];
      Tm[0] = rh[0] / wbeta;
#pragma unroll
      for (int k = 1; k < NMAT; ++k) {
        wg[k] = sp[k - 1] / wbeta;
        wbeta = dg[k] - sb[k] * wg[k];
        Tm[k] = (rh[k] - sb[k] * Tm[k - 1]) / wbeta;
      }
#pragma unroll
      for (int k = NMAT - 2; k >= 0; --k) Tm[k] = Tm[k] - wg[k + 1] * Tm[k + 1];
    }
    if (CALC) {  // :1824-1884
      if (l_cold)
        c.Tsf = l_snow ? Tm[0] : Tm[NS];
      else
        c.Tsf = c0;
      dTsf = c.Tsf - Tsf_start;
      if (c.Tsf > puny) {
        c.Tsf = c0;
        dTsf = -Tsf_start;
        if (P.l_brine) avg_Tsi = c1;
        converged = false;
      } else if (niter > 1 && Tsf_start <= -puny && fabs(dTsf) > puny && fabs(dTsf_prev) > puny &&
                 -dTsf / (dTsf_prev + puny * puny) > p5) {
        if (P.l_brine) {
          avg_Tsf = c1;
          avg_Tsi = c1;
        }
        dTsf = p5 * dTsf;
        converged = false;
      }
      c.Tsf = c.Tsf + avg_Tsf * p5 * (Tsf_start - c.Tsf);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {  // :1890-1924
      c.Tsn[k] = l_snow ? Tm[k + 1] : c0;
      if (P.l_brine) c.Tsn[k] = fmin(c.Tsn[k], c0);
      c.Tsn[k] = c.Tsn[k] + avg_Tsi * p5 * (Tsn_start[k] - c.Tsn[k]);
      c.qsn[k] = -rhos * (Lfresh - cp_ice * c.Tsn[k]);
      enew = enew + hslyr * c.qsn[k];
      Tsn_start[k] = c.Tsn[k];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {  // :1926-2001
      dqmat[k] = c0;
      reduce_kh[k] = false;
      c.Tin[k] = Tm[k + 1 + NS];
      if (P.l_brine && c.Tin[k] > P.Tmlt[k] - puny) {
        const double dTmat = c.Tin[k] - P.Tmlt[k];
        dqmat[k] = rhoi * dTmat * (cp_ice - Lfresh * P.Tmlt[k] / (c.Tin[k] * c.Tin[k]));
        c.Tin[k] = P.Tmlt[k];
        reduce_kh[k] = true;
      }
      if (!CALC && k == 0) {  // condition 2b :1961-1975
        double dTi1 = c.Tin[k] - Tin_start[k];
        if (niter > 1 && fabs(dTi1) > puny && fabs(dTi1_prev) > puny &&
            -dTi1 / (dTi1_prev + puny * puny) > p5) {
          if (P.l_brine) avg_Tsi = c1;
          dTi1 = p5 * dTi1;
          converged = false;
        }
        dTi1_prev = dTi1;
      }
      c.Tin[k] = c.Tin[k] + avg_Tsi * p5 * (Tin_start[k] - c.Tin[k]);
      if (P.l_brine)
        c.qin[k] = -rhoi * (cp_ice * (P.Tmlt[k] - c.Tin[k]) + Lfresh * (c1 - P.Tmlt[k] / c.Tin[k]) -
                            cp_ocn * P.Tmlt[k]);
      else
        c.qin[k] = -rhoi * (-cp_ice * c.Tin[k] + Lfresh);
      enew = enew + hilyr * (c.qin[k] - dqmat[k]);
      Tin_start[k] = c.Tin[k];
    }
    if (CALC) {  // :2017-2038
      if (fabs(dTsf) > Tsf_errmax) converged = false;
      f.fsurfn = f.fsurfn + dTsf * dfsurf_dT;
      f.fcondtopn = l_snow ? kh[0] * (c.Tsf - c.Tsn[0]) : kh[NS] * (c.Tsf - c.Tin[0]);
      if (c.Tsf > -puny && f.fsurfn < f.fcondtopn) converged = false;
      dTsf_prev = dTsf;
    }
    // :2053-2073
    c.fcondbot = kh[NS + NI] * (c.Tin[NI - 1] - f.Tbot);
    const double ferr = fabs((enew - c.einit) / dt - (f.fcondtopn - c.fcondbot + f.fswint));
    if (ferr > 0.9 * ferrmax) {
      converged = false;
#pragma unroll
      for (int k = 1; k <= NI; ++k)
        if (reduce_kh[k - 1] && dqmat[k - 1] > c0) {
          const double frac = fmax(0.5 * (c1 - ferr / fabs(f.fcondtopn - c.fcondbot)), p1);
          kh[k + NS] = kh[k + NS] * frac;
          kh[k + NS - 1] = kh[k + NS] * frac;
        }
    }
  }
  if (CALC) {  // :2136-2145
    f.flwoutn = f.flwoutn + dTsf_prev * dflwout_dT;
    f.fsensn = f.fsensn + dTsf_prev * dfsens_dT;
    f.flatn = f.flatn + dTsf_prev * dflat_dT;
  }
  return converged;
}

// thickness_changes :3622-4224, freeboard :4244-4377, adjust_enthalpy :4396-4492
__device__ __forceinline__ void thickness_changes(const ThermoParams& P, double dt, double yday, Col& c,
                                                  const Flx& f, Gro& g) {
  constexpr double qbotmax = -p5 * rhoi * Lfresh;
  double dzi[NI], dzs[NS];
  double esub, econ, etop_mlt, ebot_mlt, ebot_gro, wk1, dhi, dhs, qbot, qsub, hqtot;
  c.hsn_new = c0;
#pragma unroll
  for (int k = 0; k < NI; ++k) dzi[k] = c.hilyr;
#pragma unroll
  for (int k = 0; k < NS; ++k) dzs[k] = c.hslyr;
  if (!P.l_brine) {  // :3780-3812
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double Ts = (Lfresh + c.qsn[k] / rhos) / cp_ice;
      if (Ts > c0) {
        dhs = cp_ice * Ts * dzs[k] / Lfresh;
        dzs[k] = dzs[k] - dhs;
        c.qsn[k] = -rhos * Lfresh;
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const double Ti = (Lfresh + c.qin[k] / rhoi) / cp_ice;
      if (Ti > c0) {
        dhi = cp_ice * Ti * dzi[k] / Lfresh;
        dzi[k] = dzi[k] - dhi;
        c.qin[k] = -rhoi * Lfresh;
      }
    }
  }
  // :3823-3885
  wk1 = -f.flatn * dt;
  esub = fmax(wk1, c0);
  econ = fmin(wk1, c0);
  wk1 = (f.fsurfn - f.fcondtopn) * dt;
  etop_mlt = fmax(wk1, c0);
  wk1 = (c.fcondbot - g.fbot) * dt;
  ebot_mlt = fmax(wk1, c0);
  ebot_gro = fmin(wk1, c0);
  g.evapn = c0;
  if (c.hsn > puny) {
    dhs = econ / (c.qsn[0] - rhos * Lvap);
    dzs[0] = dzs[0] + dhs;
    g.evapn = g.evapn + dhs * rhos;
  } else {
    dhi = econ / (c.qin[0] - rhoi * Lvap);
    dzi[0] = dzi[0] + dhi;
    g.evapn = g.evapn + dhi * rhoi;
  }
  if (P.l_brine) {
    qbot = -rhoi * (cp_ice * (P.Tmlt[NI] - f.Tbot) + Lfresh * (c1 - P.Tmlt[NI] / f.Tbot) -
                    cp_ocn * P.Tmlt[NI]);
    qbot = fmin(qbot, qbotmax);
  } else {
    qbot = -rhoi * (cp_ice * f.Tbot + Lfresh);
  }
  dhi = ebot_gro / qbot;
  hqtot = dzi[NI - 1] * c.qin[NI - 1] + dhi * qbot;
  dzi[NI - 1] = dzi[NI - 1] + dhi;
  if (dzi[NI - 1] > puny) c.qin[NI - 1] = hqtot / dzi[NI - 1];
  g.congel = g.congel + dhi;
  if (dhi > puny && g.frz_onset < puny) g.frz_onset = yday;
#pragma unroll
  for (int k = 0; k < NS; ++k) {  // :3889-3923
    qsub = c.qsn[k] - rhos * Lvap;
    dhs = fmax(-dzs[k], esub / qsub);
    dzs[k] = dzs[k] + dhs;
    esub = esub - dhs * qsub;
    esub = fmax(esub, c0);
    g.evapn = g.evapn + dhs * rhos;
    dhs = fmax(-dzs[k], etop_mlt / c.qsn[k]);
    dzs[k] = dzs[k] + dhs;
    etop_mlt = etop_mlt - dhs * c.qsn[k];
    etop_mlt = fmax(etop_mlt, c0);
    if (dhs < -puny && g.mlt_onset < puny) g.mlt_onset = yday;
    g.melts = g.melts - dhs;
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {  // :3925-3959
    qsub = c.qin[k] - rhoi * Lvap;
    dhi = fmax(-dzi[k], esub / qsub);
    dzi[k] = dzi[k] + dhi;
    esub = esub - dhi * qsub;
    esub = fmax(esub, c0);
    g.evapn = g.evapn + dhi * rhoi;
    dhi = fmax(-dzi[k], etop_mlt / c.qin[k]);
    dzi[k] = dzi[k] + dhi;
    etop_mlt = etop_mlt - dhi * c.qin[k];
    etop_mlt = fmax(etop_mlt, c0);
    if (dhi < -puny && g.mlt_onset < puny) g.mlt_onset = yday;
    g.meltt = g.meltt - dhi;
  }
#pragma unroll
  for (int k = NI - 1; k >= 0; --k) {  // :3961-3982
    dhi = fmax(-dzi[k], ebot_mlt / c.qin[k]);
    dzi[k] = dzi[k] + dhi;
    ebot_mlt = ebot_mlt - dhi * c.qin[k];
    ebot_mlt = fmax(ebot_mlt, c0);
    g.meltb = g.meltb - dhi;
  }
#pragma unroll
  for (int k = NS - 1; k >= 0; --k) {  // :3984-4000
    dhs = fmax(-dzs[k], ebot_mlt / c.qsn[k]);
    dzs[k] = dzs[k] + dhs;
    ebot_mlt = ebot_mlt - dhs * c.qsn[k];
    ebot_mlt = fmax(ebot_mlt, c0);
  }
  g.fhocnn = g.fbot + (esub + etop_mlt + ebot_mlt) / dt;  // :4010
  if (g.fsnow > c0) {                                      // :4031-4045
    c.hsn_new = g.fsnow / rhos * dt;
    const double qsnew = -rhos * Lfresh;
    const double hstot = dzs[0] + c.hsn_new;
    if (hstot > c0) {
      c.qsn[0] = (dzs[0] * c.qsn[0] + c.hsn_new * qsnew) / hstot;
      c.qsn[0] = fmin(c.qsn[0], -rhos * Lfresh);
      dzs[0] = hstot;
    }
  }
  c.hin = c0;
  c.hsn = c0;
#pragma unroll
  for (int k = 0; k < NI; ++k) c.hin = c.hin + dzi[k];
#pragma unroll
  for (int k = 0; k < NS; ++k) c.hsn = c.hsn + dzs[k];
  {  // freeboard :4315-4375
    double dhin = c0, dhsn = c0, hqs = c0;
    wk1 = c.hsn - c.hin * (rhow - rhoi) / rhos;
    if (wk1 > puny && c.hsn > puny) {
      dhsn = fmin(wk1 * rhoi / rhow, c.hsn);
      dhin = dhsn * rhos / rhoi;
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k)
      if (dhin > puny) {
        dhs = fmin(dhsn, dzs[k]);
        c.hsn = c.hsn - dhs;
        dzs[k] = dzs[k] - dhs;
        dhsn = dhsn - dhs;
        dhsn = fmax(dhsn, c0);
        hqs = hqs + dhs * c.qsn[k];
      }
    if (dhin > puny) {
      wk1 = dzi[0] + dhin;
      c.hin = c.hin + dhin;
      c.qin[0] = (dzi[0] * c.qin[0] + hqs) / wk1;
      dzi[0] = wk1;
      g.snoice = g.snoice + dhin;
    }
  }
  // :4096-4145
  if (c.hin > c0) {
    c.hilyr = c.hin / (double)NI;
  } else {
    c.hin = c0;
    c.hilyr = c0;
  }
  if (c.hsn > c0) {
    c.hslyr = c.hsn / (double)NS;
  } else {
    c.hsn = c0;
    c.hslyr = c0;
  }
  {
    double zi1[NI + 1], zi2[NI + 1], hq[NI];
    zi1[0] = c0; zi1[NI] = c.hin; zi2[0] = c0; zi2[NI] = c.hin;
#pragma unroll
    for (int k = 0; k < NI - 1; ++k) {
      zi1[k + 1] = zi1[k] + dzi[k];
      zi2[k + 1] = zi2[k] + c.hilyr;
    }
    double rhlyr = c0;  // adjust_enthalpy :4451-4490
    if (c.hin > puny) rhlyr = c1 / c.hilyr;
#pragma unroll
    for (int k2 = 0; k2 < NI; ++k2) {
      hq[k2] = c0;
#pragma unroll
      for (int k1 = 0; k1 < NI; ++k1) {
        double hovlp = fmin(zi1[k1 + 1], zi2[k2 + 1]) - fmax(zi1[k1], zi2[k2]);
        hovlp = fmax(hovlp, c0);
        hq[k2] = hq[k2] + hovlp * c.qin[k1];
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) c.qin[k] = hq[k] * rhlyr;
  }
  static_assert(NS == 1, "snow re-layering (ice_therm_vertical.F90:4156-4192) needs nslyr > 1 support");
  // :4199-4222
  c.efinal = -g.evapn * Lvap;
  g.evapn = g.evapn / dt;
#pragma unroll
  for (int k = 0; k < NS; ++k) c.efinal = c.efinal + c.hslyr * c.qsn[k];
#pragma unroll
  for (int k = 0; k < NI; ++k) c.efinal = c.efinal + c.hilyr * c.qin[k];
}

template <bool CALC>
__device__ __forceinline__ void zero_outputs(const ThermoArgs& a, size_t c2d) {  // :299-329
  a.fsensn[c2d] = c0; a.fswabsn[c2d] = c0; a.flwoutn[c2d] = c0; a.evapn[c2d] = c0;
  a.freshn[c2d] = c0; a.fsaltn[c2d] = c0; a.fhocnn[c2d] = c0;
  a.meltt[c2d] = c0; a.meltb[c2d] = c0; a.melts[c2d] = c0; a.congel[c2d] = c0; a.snoice[c2d] = c0;
  if (CALC) {  // :321-329; inputs otherwise
    a.flatn[c2d] = c0; a.fsurfn[c2d] = c0; a.fcondtopn[c2d] = c0;
  }
}

// One column of thermo_vertical :108-515.  q: cell offset inside the (nx,ny) plane;
// n, b: category and block (0-based); order: rank of this column in the reference's
// failure-reporting order.
// ZERO: the caller has NOT zeroed the 15 output planes of this column (batched kernel): every exit
// of the routine then leaves them as the reference's initial zeroing (:299-329) plus its own stores would.
template <bool CALC, bool ZERO>
__device__ __forceinline__ void column(const ThermoArgs& a, size_t q, int n, int b,
                                       unsigned long long order) {
  const ThermoParams& P = a.p;
  const size_t np = (size_t)a.nx * a.ny;
  const size_t cb = (size_t)b * a.ncat + n;          // (category, block) plane group
  const size_t c2d = cb * np + q;                    // per-category 2-D
  const size_t f2d = (size_t)b * np + q;             // per-block 2-D
  const size_t tq = (cb * NTRCR + (P.nt_Tsfc - 1)) * np + q;
  const size_t eq = ((size_t)b * a.ncat * NI + (size_t)n * NI) * np + q;
  const size_t sq = ((size_t)b * a.ncat * NS + (size_t)n * NS) * np + q;
  const size_t iq = (cb * NI) * np + q, ssq = (cb * NS) * np + q;

  Col c;
  Flx f;
  Gro g;
  double ei[NI], es[NS];
#pragma unroll
  for (int k = 0; k < NI; ++k) ei[k] = a.eicen[eq + (size_t)k * np];
#pragma unroll
  for (int k = 0; k < NS; ++k) es[k] = a.esnon[sq + (size_t)k * np];
  const double aic = a.aicen[c2d];
  // the forcing and shortwave terms are fetched together with the state, ahead of the branch on the profile's
  // validity (the compiler may not move a load above it): one wait for memory instead of two (no measurable
  // difference: the kernel is bound by its arithmetic)
  f.rhoa = a.rhoa[f2d]; f.flw = a.flw[f2d]; f.potT = a.potT[f2d]; f.Qa = a.Qa[f2d];
  f.Tbot = a.Tbot[f2d];
  f.shcoef = a.shcoef[c2d]; f.lhcoef = a.lhcoef[c2d];
  f.fswsfc = a.fswsfc[c2d]; f.fswint = a.fswint[c2d]; f.fswthrun = a.fswthrun[c2d];
#pragma unroll
  for (int k = 0; k < NS; ++k) f.Sswabs[k] = a.Sswabs[ssq + (size_t)k * np];
#pragma unroll
  for (int k = 0; k < NI; ++k) f.Iswabs[k] = a.Iswabs[iq + (size_t)k * np];
#ifdef THERMO_FLOOR   // timing-only build (scripts/build_ab_therm.sh floor -DTHERMO_FLOOR): the column's loads and stores
  {                    // without its arithmetic -- what the access pattern alone costs (wrong results)
    const double t = aic + a.vicen[c2d] + a.vsnon[c2d] + a.trcrn[tq] + f.rhoa + f.flw + f.potT + f.Qa + f.Tbot + f.shcoef +
                     f.lhcoef + f.fswthrun + a.fbot[f2d] + a.fsnow[f2d] + a.mlt_onset[f2d] + a.frz_onset[f2d];
    a.fswsfc[c2d] = f.fswsfc + t; a.fswint[c2d] = f.fswint + t;
    for (int k = 0; k < NS; ++k) a.Sswabs[ssq + (size_t)k * np] = f.Sswabs[k] + t;
    for (int k = 0; k < NI; ++k) a.Iswabs[iq + (size_t)k * np] = f.Iswabs[k] + t;
    double* const outs[15] = {a.fsurfn, a.fcondtopn, a.fsensn, a.flatn, a.fswabsn, a.flwoutn, a.fhocnn, a.evapn,
                              a.meltt, a.melts, a.meltb, a.congel, a.snoice, a.freshn, a.fsaltn};
    for (int k = 0; k < 15; ++k) outs[k][c2d] = t + k;
    a.vicen[c2d] = t + 15; a.vsnon[c2d] = t + 16; a.trcrn[tq] = t + 17;
    for (int k = 0; k < NI; ++k) a.eicen[eq + (size_t)k * np] = ei[k] + t;
    for (int k = 0; k < NS; ++k) a.esnon[sq + (size_t)k * np] = es[k] + t;
    return;
  }
#endif
  const unsigned stage0 = init_profile(P, aic, a.vicen[c2d], a.vsnon[c2d], a.trcrn[tq], ei, es, c);
  if (stage0) {
    atomicMin(a.errkey, ((unsigned long long)cb << 44) | ((unsigned long long)stage0 << 40) | order);
    if (ZERO) zero_outputs<CALC>(a, c2d);
    return;
  }
  const double worki = c.hin, works = c.hsn;
  f.fsurfn = f.fcondtopn = f.fsensn = f.flatn = f.fswabsn = f.flwoutn = c0;
  if (!CALC) {  // intent(in) when calc_Tsfc = F (:213-217)
    f.fsurfn = a.fsurfn[c2d]; f.fcondtopn = a.fcondtopn[c2d]; f.flatn = a.flatn[c2d];
  }
  int iters = 0;
  const bool conv = temperature_changes<CALC>(P, a.dt, c, f, iters);
  if (a.niter) a.niter[c2d] = (unsigned char)min(iters, 255);   // solver iterations of this column: next step's sort key
  a.fswsfc[c2d] = f.fswsfc; a.fswint[c2d] = f.fswint;
#pragma unroll
  for (int k = 0; k < NS; ++k) a.Sswabs[ssq + (size_t)k * np] = f.Sswabs[k];
#pragma unroll
  for (int k = 0; k < NI; ++k) a.Iswabs[iq + (size_t)k * np] = f.Iswabs[k];
  a.fsurfn[c2d] = f.fsurfn; a.fcondtopn[c2d] = f.fcondtopn; a.fsensn[c2d] = f.fsensn;
  a.flatn[c2d] = f.flatn; a.fswabsn[c2d] = f.fswabsn; a.flwoutn[c2d] = f.flwoutn;
  if (!conv) {
    atomicMin(a.errkey, ((unsigned long long)cb << 44) | ((unsigned long long)ST_NOCONV << 40) | order);
    if (ZERO) {
      a.evapn[c2d] = c0; a.freshn[c2d] = c0; a.fsaltn[c2d] = c0; a.fhocnn[c2d] = c0; a.meltt[c2d] = c0;
      a.melts[c2d] = c0; a.meltb[c2d] = c0; a.congel[c2d] = c0; a.snoice[c2d] = c0;
    }
    return;
  }
  g.fbot = a.fbot[f2d]; g.fsnow = a.fsnow[f2d];
  g.mlt_onset = a.mlt_onset[f2d]; g.frz_onset = a.frz_onset[f2d];
  g.meltt = g.melts = g.meltb = g.congel = g.snoice = c0;
  const double mlt0 = g.mlt_onset, frz0 = g.frz_onset;
  thickness_changes(P, a.dt, a.yday, c, f, g);
  a.fhocnn[c2d] = g.fhocnn; a.evapn[c2d] = g.evapn; a.meltt[c2d] = g.meltt; a.melts[c2d] = g.melts;
  a.meltb[c2d] = g.meltb; a.congel[c2d] = g.congel; a.snoice[c2d] = g.snoice;
  // onsets are shared by the categories of a cell: only ever set to yday (idempotent)
  if (g.mlt_onset != mlt0) a.mlt_onset[f2d] = g.mlt_onset;
  if (g.frz_onset != frz0) a.frz_onset[f2d] = g.frz_onset;
  {  // conservation_check_vthermo :4573-4610
    const double einp = (f.fsurfn - f.flatn + f.fswint - g.fhocnn - g.fsnow * Lfresh) * a.dt;
    const double ferr = fabs(c.efinal - c.einit - einp) / a.dt;
    if (ferr > ferrmax) {
      atomicMin(a.errkey, ((unsigned long long)cb << 44) | ((unsigned long long)ST_ECONS << 40) | order);
      if (ZERO) {
        a.freshn[c2d] = c0; a.fsaltn[c2d] = c0;
      }
      return;
    }
  }
  {  // :474-485
    const double dhi = c.hin - worki, dhs = c.hsn - works;
    a.freshn[c2d] = g.evapn - (rhoi * dhi + rhos * (dhs - c.hsn_new)) / a.dt;
    a.fsaltn[c2d] = -rhoi * dhi * ice_ref_salinity * p001 / a.dt;
  }
  // update_state_vthermo :4699-4745 (Tf dummy = Tbot, :496)
  if (c.hin > c0) {
    const double vi = aic * c.hin, vs = aic * c.hsn;
    a.vicen[c2d] = vi;
    a.vsnon[c2d] = vs;
    a.trcrn[tq] = c.Tsf;
#pragma unroll
    for (int k = 0; k < NI; ++k) a.eicen[eq + (size_t)k * np] = c.qin[k] * vi / (double)NI;
#pragma unroll
    for (int k = 0; k < NS; ++k) a.esnon[sq + (size_t)k * np] = c.qsn[k] * vs / (double)NS;
  } else {
    a.aicen[c2d] = c0; a.vicen[c2d] = c0; a.vsnon[c2d] = c0;
    a.trcrn[tq] = f.Tbot;
#pragma unroll
    for (int k = 0; k < NI; ++k) a.eicen[eq + (size_t)k * np] = c0;
#pragma unroll
    for (int k = 0; k < NS; ++k) a.esnon[sq + (size_t)k * np] = c0;
  }
}


// reference-signature form: one lane per entry of the compressed cell list
template <bool CALC>
__global__ __launch_bounds__(256) void k_thermo_list(const ThermoArgs a) {
  const int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= a.icells) return;
  const size_t q = (size_t)(a.indxj[ij] - 1) * a.nx + (a.indxi[ij] - 1);
  column<CALC, false>(a, q, 0, 0, (unsigned long long)ij);
}

template <bool CALC>
__global__ __launch_bounds__(256) void k_thermo_zero(const ThermoArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)a.nx * a.ny) zero_outputs<CALC>(a, t);
}

// batched form: grid (cell blocks, ncat, nblocks); the aicen > puny test on the physical
// domain replaces the host-side list compaction of step_therm1 (CICE_RunMod.F90:380-389)
// Occupancy: the column state needs ~260 VGPRs unconstrained = ONE wavefront per SIMD; capping at
// 256 (two workgroups of 256 per CU) costs a 20-byte spill and gives two wavefronts per SIMD, which
// hides the latency of the dependent fp64 division chains: +40 % measured (profiles/).
#ifndef CICE_THERMO_MIN_BLOCKS
#define CICE_THERMO_MIN_BLOCKS 2
#endif
template <bool CALC>
__global__ __launch_bounds__(256, CICE_THERMO_MIN_BLOCKS) void k_thermo_dense(const ThermoArgs a) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y, b = blockIdx.z;
  if (q >= np) return;
  const size_t c2d = ((size_t)b * a.ncat + n) * np + q;
  const int j = (int)(q / a.nx) + 1, i = (int)(q - (size_t)(j - 1) * a.nx) + 1;
  const int ilo = a.blk[4 * b], ihi = a.blk[4 * b + 1], jlo = a.blk[4 * b + 2], jhi = a.blk[4 * b + 3];
  const bool active = i >= ilo && i <= ihi && j >= jlo && j <= jhi && a.aicen[c2d] > puny;
  // every point's 15 output planes are written exactly once: by its column, or zeroed here
  if (active) column<CALC, true>(a, q, n, b, (unsigned long long)q);
  else zero_outputs<CALC>(a, c2d);
  unsigned long long cnt = __popcll(__ballot(active));
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(a.nupdates + (size_t)(blockIdx.x % THERMO_COUNT_SLOTS) * THERMO_COUNT_STRIDE, cnt);
}

// ---- homogeneous wavefronts: columns sorted by the work they are expected to take ---------------------------------
//
// A column takes 1 to ~16 iterations of the implicit solve (3.4 on average in the bench workload) and one of four
// branch patterns (snow-covered or bare, cold or melting surface); a wavefront lasts as long as its slowest lane:
// with 64 consecutive cells per wavefront a third of the lane-iterations is idle (profiles/r02_sq_counters_gx1.csv:
// 47 of 64 lanes active).  Which column sits in which lane changes no column's bits, so: k_thermo_sort orders the
// columns of every chunk of SORT_CHUNK consecutive cells of a (category, block) plane by a key -- the iterations the
// column took in the PREVIOUS step (the state of a column changes slowly from step to step; in the first step the
// count is unknown = 0), then snow / no snow and cold / melting surface -- with a counting sort in LDS, and writes the
// permutation; k_thermo_perm is k_thermo_dense with lane l of wavefront w working on column perm[64 w + l].
// The chunk bounds the scatter of a wavefront's memory accesses (512 cells = 32 cache lines of 128 B per field).
struct SortArgs {
  int nx, ny, ncat, nblocks, chunk, group;
  size_t np_pad;                        // cells per plane rounded up to whole chunks
  const int32_t* blk;
  const double *aicen, *vsnon, *tsfc;   // tsfc: the Tsfc tracer plane of (category, block) cb at tsfc + cb * tstride
  size_t tstride;
  const unsigned char* niter;
  int32_t* perm;                        // [ncat * nblocks][np_pad]: cell index q, bit 31 set: not an active column; -1: no cell
};

constexpr int SORT_BINS = 128;          // key = min(iterations, 31) * 4 + snow * 2 + cold; 127 = nothing active

// What is sorted are GROUPS of `group` adjacent cells (1, 8, 16: a power of two), by the largest key among their
// active columns: a wavefront then works on 64 / group runs of adjacent cells and its loads stay (nearly) coalesced --
// with single columns as the unit (group = 1) the scattered accesses cost more than the idle lanes did (measured).
__global__ __launch_bounds__(256) void k_thermo_sort(const SortArgs a) {
  __shared__ int s_cnt[SORT_BINS];
  const size_t np = (size_t)a.nx * a.ny;
  const int n = blockIdx.y, b = blockIdx.z;
  const size_t cb = (size_t)b * a.ncat + n;
  const size_t q0 = (size_t)blockIdx.x * a.chunk;
  const int ilo = a.blk[4 * b], ihi = a.blk[4 * b + 1], jlo = a.blk[4 * b + 2], jhi = a.blk[4 * b + 3];
  const int G = a.group, lane = threadIdx.x & 63;
  if (threadIdx.x < SORT_BINS) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  constexpr int MAXPER = 8;             // chunk <= 2048
  int gkey[MAXPER], rank[MAXPER];
  bool act[MAXPER];
  const int per = a.chunk / 256;        // the chunk is a multiple of 256
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    gkey[e] = -1;
    act[e] = false;
    if (e >= per) continue;
    const size_t q = q0 + (size_t)e * 256 + threadIdx.x;     // consecutive threads, consecutive cells
    int k = -1;
    if (q < np) {
      const size_t c2d = cb * np + q;
      const int j = (int)(q / a.nx) + 1, i = (int)(q - (size_t)(j - 1) * a.nx) + 1;
      const double ai = a.aicen[c2d];
      if (i >= ilo && i <= ihi && j >= jlo && j <= jhi && ai > puny) {
        const bool snow = a.vsnon[c2d] / ai > hs_min;
        const bool cold = a.tsfc[cb * a.tstride + q] <= -puny;
        // heaviest first: bin 0 = most iterations (the wavefronts that run longest start first, see k_thermo_perm)
        k = SORT_BINS - 2 - min(min((int)a.niter[c2d], 31) * 4 + (snow ? 2 : 0) + (cold ? 1 : 0), SORT_BINS - 2);
        act[e] = true;
      }
    }
    int gk = k < 0 ? SORT_BINS - 1 : k;  // heaviest column of the group = smallest bin (SORT_BINS - 1: nothing active in it)
    for (int d = 1; d < G; d <<= 1) gk = min(gk, __shfl_xor(gk, d));
    int r = 0;
    if ((lane & (G - 1)) == 0) r = atomicAdd(&s_cnt[gk], 1);
    rank[e] = __shfl(r, lane & ~(G - 1));
    gkey[e] = gk;
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive prefix sum over the bins: one wavefront, two bins per lane
    const int c0_ = s_cnt[2 * threadIdx.x], c1_ = s_cnt[2 * threadIdx.x + 1];
    int v = c0_ + c1_;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(v, d);
      if ((int)threadIdx.x >= d) v += u;
    }
    const int excl = v - (c0_ + c1_);
    s_cnt[2 * threadIdx.x] = excl;
    s_cnt[2 * threadIdx.x + 1] = excl + c0_;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    if (gkey[e] < 0) continue;
    const size_t q = q0 + (size_t)e * 256 + threadIdx.x;
    const int pos = (s_cnt[gkey[e]] + rank[e]) * G + (lane & (G - 1));
    a.perm[cb * a.np_pad + q0 + pos] = q < np ? ((int32_t)q | (act[e] ? 0 : (int32_t)0x80000000)) : -1;
  }
}

template <bool CALC>
__global__ __launch_bounds__(256, CICE_THERMO_MIN_BLOCKS) void k_thermo_perm(const ThermoArgs a, const int32_t* perm,
                                                                             size_t np_pad, int chunk) {
  const size_t np = (size_t)a.nx * a.ny;
  // workgroup (= wavefront) w of a plane takes the (w / nchunks)-th wavefront of chunk w % nchunks: the heaviest
  // wavefronts of all chunks are dispatched first, the lightest last (a short tail)
  const unsigned wpc = (unsigned)chunk / blockDim.x, nchunks = (unsigned)(np_pad / (size_t)chunk);
  const size_t slot = ((size_t)(blockIdx.x % nchunks) * wpc + blockIdx.x / nchunks) * blockDim.x + threadIdx.x;
  const int n = blockIdx.y, b = blockIdx.z;
  if (slot >= np_pad) return;
  const size_t cb = (size_t)b * a.ncat + n;
  const int32_t e = perm[cb * np_pad + slot];
  if (e == -1) return;                       // padding behind the last cell of the plane
  const size_t q = (size_t)(e & 0x7fffffff);
  const bool active = e >= 0;
  // every point's 15 output planes are written exactly once: by its column, or zeroed here
  if (active) column<CALC, true>(a, q, n, b, (unsigned long long)q);
  else zero_outputs<CALC>(a, cb * np + q);
  unsigned long long cnt = __popcll(__ballot(active));
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(a.nupdates + (size_t)(blockIdx.x % THERMO_COUNT_SLOTS) * THERMO_COUNT_STRIDE, cnt);
}

// merge_fluxes (ice_flux.F90:730-760): one lane per cell accumulates the categories in order
__global__ __launch_bounds__(256) void k_merge(const MergeArgs a) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (q >= np) return;
  const int j = (int)(q / a.nx) + 1, i = (int)(q - (size_t)(j - 1) * a.nx) + 1;
  const int ilo = a.blk[4 * b], ihi = a.blk[4 * b + 1], jlo = a.blk[4 * b + 2], jhi = a.blk[4 * b + 3];
  if (i < ilo || i > ihi || j < jlo || j > jhi) return;
  const size_t f2d = (size_t)b * np + q;
  double acc[20];
  bool any = false;
  for (int n = 0; n < a.ncat; ++n) {
    const size_t c2d = ((size_t)b * a.ncat + n) * np + q;
    const double ai = a.aicen_init[c2d];
    if (!(ai > puny)) continue;
    if (!any) {
#pragma unroll
      for (int k = 0; k < 20; ++k) acc[k] = a.acc[k][f2d];
      any = true;
    }
#pragma unroll
    for (int k = 0; k < 20; ++k) {
      if (k == 7)
        acc[k] = acc[k] + (a.src[k][c2d] - (c1 - emissivity) * a.flw[f2d]) * ai;
      else
        acc[k] = acc[k] + a.src[k][c2d] * ai;
    }
  }
  if (any) {
#pragma unroll
    for (int k = 0; k < 20; ++k) a.acc[k][f2d] = acc[k];
  }
}

// frzmlt_bottom_lateral :605-824 (cpchr: a compile-time constant of the stand-alone build, formed from the namelist's
// chio in the coupled one, :673-694)
__global__ __launch_bounds__(256) void k_frzmlt(const FrzmltArgs a) {
  const size_t np = (size_t)a.nx * a.ny;
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= np) return;
#ifdef CICE4_AMD_AUSCOM
  const double cpchr = -cp_ocn * rhow * a.chio;
#else
  constexpr double cpchr = -cp_ocn * rhow * 0.006;
#endif
  constexpr double floediam = 300.0, alpha = 0.66, m1 = 1.6e-6, m2 = 1.36;
  const int j = (int)(q / a.nx) + 1, i = (int)(q - (size_t)(j - 1) * a.nx) + 1;
  double rside = c0, Tbot = a.Tf[q], fbot = c0;
  if (i >= a.ilo && i <= a.ihi && j >= a.jlo && j <= a.jhi && a.aice[q] > puny && a.frzmlt[q] < c0) {
    double fside = c0;
    const double deltaT = fmax(a.sst[q] - Tbot, c0);
    double ustar = sqrt(sqrt(a.strocnxT[q] * a.strocnxT[q] + a.strocnyT[q] * a.strocnyT[q]) / rhow);
    ustar = fmax(ustar, a.ustar_min);
    fbot = cpchr * deltaT * ustar;
    fbot = fmax(fbot, a.frzmlt[q]);
    const double wlat = m1 * pow_libm(deltaT, m2);
    rside = wlat * a.dt * pi / (alpha * floediam);
    rside = fmax(c0, fmin(rside, c1));
    for (int n = 0; n < NCAT; ++n) {
      double etot = c0;
      for (int k = 0; k < NS; ++k) etot = etot + a.esnon[(size_t)(n * NS + k) * np + q];
      for (int k = 0; k < NI; ++k) etot = etot + a.eicen[(size_t)(n * NI + k) * np + q];
      fside = fside + rside * etot / a.dt;
    }
    double xtmp = a.frzmlt[q] / (fbot + fside + puny);
    xtmp = fmin(xtmp, c1);
    fbot = fbot * xtmp;
    rside = rside * xtmp;
  }
  a.rside[q] = rside;
  a.Tbot[q] = Tbot;
  a.fbot[q] = fbot;
}

}  // namespace

void thermo_launch_list(const ThermoArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  const dim3 gz((unsigned)((np + 255) / 256)), gl((unsigned)((a.icells + 255) / 256));
  if (a.p.calc_Tsfc) {
    hipLaunchKernelGGL(k_thermo_zero<true>, gz, dim3(256), 0, s, a);
    if (a.icells > 0) hipLaunchKernelGGL(k_thermo_list<true>, gl, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_thermo_zero<false>, gz, dim3(256), 0, s, a);
    if (a.icells > 0) hipLaunchKernelGGL(k_thermo_list<false>, gl, dim3(256), 0, s, a);
  }
  CICE_HIP(hipGetLastError());
}

void thermo_launch_dense(const ThermoArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  // one wavefront per workgroup: columns take 1 to 16 solver iterations, wavefronts never talk to each other, and
  // the dispatcher fills a freed slot at once instead of waiting for the slowest of four (+3 % at gx1 and 0.1 degree)
  constexpr unsigned bs = 64;
  const dim3 g((unsigned)((np + bs - 1) / bs), a.ncat, a.nblocks);
  if (a.p.calc_Tsfc)
    hipLaunchKernelGGL(k_thermo_dense<true>, g, dim3(bs), 0, s, a);
  else
    hipLaunchKernelGGL(k_thermo_dense<false>, g, dim3(bs), 0, s, a);
  CICE_HIP(hipGetLastError());
}

size_t thermo_sorted_plane(size_t np, int chunk) { return (np + chunk - 1) / chunk * chunk; }

// dense step with the columns of every chunk of `chunk` cells sorted by expected work (see k_thermo_sort); tsfc / tstride:
// the Tsfc tracer plane of (category, block) cb starts at tsfc + cb * tstride
void thermo_launch_sorted(const ThermoArgs& a, int chunk, int group, int32_t* perm, const double* tsfc, size_t tstride,
                          hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  SortArgs sa{};
  sa.nx = a.nx; sa.ny = a.ny; sa.ncat = a.ncat; sa.nblocks = a.nblocks; sa.chunk = chunk; sa.group = group; sa.blk = a.blk;
  sa.np_pad = thermo_sorted_plane(np, chunk);
  sa.aicen = a.aicen; sa.vsnon = a.vsnon; sa.tsfc = tsfc; sa.tstride = tstride; sa.niter = a.niter; sa.perm = perm;
  hipLaunchKernelGGL(k_thermo_sort, dim3((unsigned)((np + chunk - 1) / chunk), a.ncat, a.nblocks), dim3(256), 0, s, sa);
  constexpr unsigned bs = 64;
  const dim3 g((unsigned)((sa.np_pad + bs - 1) / bs), a.ncat, a.nblocks);
  if (a.p.calc_Tsfc)
    hipLaunchKernelGGL(k_thermo_perm<true>, g, dim3(bs), 0, s, a, (const int32_t*)perm, sa.np_pad, chunk);
  else
    hipLaunchKernelGGL(k_thermo_perm<false>, g, dim3(bs), 0, s, a, (const int32_t*)perm, sa.np_pad, chunk);
  CICE_HIP(hipGetLastError());
}

void merge_launch(const MergeArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  hipLaunchKernelGGL(k_merge, dim3((unsigned)((np + 255) / 256), a.nblocks), dim3(256), 0, s, a);
  CICE_HIP(hipGetLastError());
}

void frzmlt_launch(const FrzmltArgs& a, hipStream_t s) {
  const size_t np = (size_t)a.nx * a.ny;
  hipLaunchKernelGGL(k_frzmlt, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, a);
  CICE_HIP(hipGetLastError());
}

}  // namespace cice
