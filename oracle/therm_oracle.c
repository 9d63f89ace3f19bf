/* TEST INFRASTRUCTURE ONLY (see cice_oracle.h).  CPU restatement of the
 * reference column thermodynamics, source/ice_therm_vertical.F90 (citations
 * are file:line under /root/reference), for heat_capacity = T, calc_Tsfc = T and F
 * (the configuration of input_templates/gx3/ice_in and the COSIMA configs).
 *
 * The reference sweeps compressed cell lists phase by phase; every cell's
 * arithmetic is independent of every other cell's, so this restatement walks
 * one column at a time through the same sequence of operations.  The only
 * cross-cell behaviour -- which failing cell is reported through
 * (l_stop, istop, jstop) when several fail -- is reproduced by ranking
 * failures in the order the reference would meet them (stage, then list
 * position). */
#include "cice_oracle.h"
#include <math.h>
#include <stddef.h>

#define NI ORC_NILYR
#define NS ORC_NSLYR
#define NMAT (NI + NS + 1)

/* drivers/cice4/ice_constants.F90:49-121 */
static const double rhos = 330.0, rhoi = 917.0, rhow = 1026.0;
static const double emissivity = 0.95, cp_ice = 2106.0, depressT = 0.054;
#ifdef ORC_AUSCOM /* drivers/access-om/ice_constants.F90:21,48: the two constants the coupled build takes from MOM */
static const double cp_ocn = 3989.24495292815, ice_ref_salinity = 5.0;
#else
static const double cp_ocn = 4218.0, ice_ref_salinity = 4.0;
#endif
static const double pi = 3.14159265358979323846;
static const double stefan_boltzmann = 567.0e-10, Tffresh = 273.15, Lsub = 2.835e6,
                    Lvap = 2.501e6;
#define Lfresh (Lsub - Lvap)
static const double kice = 2.03, ksno = 0.30;
static const double qqqice = 11637800.0, TTTice = 5897.8;
static const double puny = 1.0e-11;
static const double c0 = 0.0, c1 = 1.0, c2 = 2.0, c4 = 4.0, p5 = 0.5, p1 = 0.1, p001 = 0.001;
/* ice_therm_vertical.F90:45-49,64-65 */
static const double saltmax = 3.2, hs_min = 1.0e-4, betak = 0.13, kimin = 0.10;
static const double ferrmax = 1.0e-3;

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* diagnostics for tests/bench: histogram of temperature-solver iterations per column */
long orc_iter_hist[101];

/* ice_therm_vertical.F90:533-584 */
void orc_init_thermo(int heat_capacity, int calc_Tsfc, int conduct, double ustar_min,
                     orc_thermo_cfg *c) {
  const double nsal = 0.407, msal = 0.573, min_salin = 0.1;
  c->heat_capacity = heat_capacity;
  c->calc_Tsfc = calc_Tsfc;
  c->conduct = conduct;
  c->ustar_min = ustar_min;
  c->tr_iage = 1;
  c->nt_Tsfc = 1;
  c->nt_iage = 2;
  c->l_brine = (saltmax > min_salin && heat_capacity);
  if (c->l_brine) {
    for (int k = 1; k <= NI; k++) {
      double zn = ((double)k - p5) / (double)NI;
      c->salin[k - 1] = (saltmax / c2) * (c1 - cos(pi * pow(zn, nsal / (msal + zn))));
      c->Tmlt[k - 1] = -c->salin[k - 1] * depressT;
    }
    c->salin[NI] = saltmax;
    c->Tmlt[NI] = -c->salin[NI] * depressT;
  } else {
    for (int k = 0; k <= NI; k++) c->salin[k] = c->Tmlt[k] = c0;
  }
}

/* failure stages in the order the reference meets them */
enum {
  ST_OK = 0,
  ST_TSN_HIGH = 1,  /* :1025-1054 */
  ST_TSN_LOW = 2,   /* :1056-1078 */
  ST_TIN_BASE = 3,  /* 3+2(k-1): Tin>Tmax layer k (:1144), 4+2(k-1): Tin<Tmin (:1170) */
  ST_NOCONV = 3 + 2 * NI, /* :2092-2130 */
  ST_ECONS = 4 + 2 * NI   /* :4573-4610 */
};

typedef struct {
  double hin, hsn, hilyr, hslyr, Tsf, einit, efinal, fcondbot, hsn_new;
  double qin[NI], Tin[NI], qsn[NS], Tsn[NS];
} column;

/* :1227-1260 */
static double Tin_from_qin(const orc_thermo_cfg *c, double qin, double Tmltk) {
  if (c->l_brine) {
    double aa1 = cp_ice;
    double bb1 = (cp_ocn - cp_ice) * Tmltk - qin / rhoi - Lfresh;
    double cc1 = Lfresh * Tmltk;
    return (-bb1 - sqrt(bb1 * bb1 - c4 * aa1 * cc1)) / (c2 * aa1);
  }
  return (Lfresh + qin / rhoi) / cp_ice;
}

/* init_vertical_profile :844-1211.  Returns failure stage or 0. */
static int init_vertical_profile(const orc_thermo_cfg *c, double aicen, double vicen,
                                 double vsnon, double Tsfcn, const double *eicen_k,
                                 const double *esnon_k, column *col) {
  const double Tmin = -100.0, rnslyr = (double)NS;
  int stage = ST_OK;
  col->einit = c0;
  col->Tsf = Tsfcn;
  col->hin = vicen / aicen;
  col->hsn = vsnon / aicen;
  col->hilyr = col->hin / (double)NI;
  col->hslyr = col->hsn / rnslyr;
  for (int k = 0; k < NS; k++) {
    double Tmax;
    if (col->hslyr > hs_min / rnslyr && c->heat_capacity) {
      col->qsn[k] = esnon_k[k] * rnslyr / vsnon;
      Tmax = -col->qsn[k] * puny * rnslyr / (rhos * cp_ice * vsnon);
    } else {
      col->qsn[k] = -rhos * Lfresh;
      Tmax = puny;
    }
    col->Tsn[k] = (Lfresh + col->qsn[k] / rhos) / cp_ice;
    if (col->Tsn[k] > Tmax) {
      if (stage == ST_OK || stage > ST_TSN_HIGH) stage = ST_TSN_HIGH;
    } else if (col->Tsn[k] < Tmin) {
      if (stage == ST_OK) stage = ST_TSN_LOW;
    }
  }
  for (int k = 0; k < NS; k++) {
    if (col->Tsn[k] > c0) {
      col->Tsn[k] = c0;
      col->qsn[k] = -rhos * Lfresh;
    }
    col->einit = col->einit + col->hslyr * col->qsn[k];
  }
  for (int k = 0; k < NI; k++) {
    double Tmax;
    col->qin[k] = eicen_k[k] * (double)NI / vicen;
    col->Tin[k] = Tin_from_qin(c, col->qin[k], c->Tmlt[k]);
    if (c->l_brine)
      Tmax = c->Tmlt[k];
    else
      Tmax = -col->qin[k] * puny / (rhos * cp_ice * vicen);
    if (col->Tin[k] > Tmax) {
      if (stage == ST_OK) stage = ST_TIN_BASE + 2 * k;
    } else if (col->Tin[k] < Tmin) {
      if (stage == ST_OK) stage = ST_TIN_BASE + 2 * k + 1;
    }
    if (col->Tin[k] > c0) {
      col->Tin[k] = c0;
      col->qin[k] = -rhoi * Lfresh;
    }
    col->einit = col->einit + col->hilyr * col->qin[k];
  }
  return stage;
}

typedef struct { /* the (i,j) fields one column touches */
  double rhoa, flw, potT, Qa, shcoef, lhcoef, Tbot;
  double fswsfc, fswint, fswthrun, Sswabs[NS], Iswabs[NI];
  double fsurfn, fcondtopn, fsensn, flatn, fswabsn, flwoutn;
} fluxes;

/* temperature_changes :1288-2148 with conductivity :2169, surface_fluxes :2314,
 * get_matrix_elements_calc_Tsfc :2447 (calc_Tsfc = T) or get_matrix_elements_know_Tsfc :2777
 * (calc_Tsfc = F: fsurfn, fcondtopn, flatn are inputs, Tsf is not solved for),
 * tridiag_solver :3069. Returns converged flag. */
static int temperature_changes(const orc_thermo_cfg *c, double dt, column *col, fluxes *f) {
  const int nitermax = 100;
  const double Tsf_errmax = 5.0e-4;
  const double *Tmlt = c->Tmlt, *salin = c->salin;
  double *Tin = col->Tin, *Tsn = col->Tsn, *qin = col->qin, *qsn = col->qsn;
  const double hilyr = col->hilyr, hslyr = col->hslyr;
  const int calc = c->calc_Tsfc;
  int converged = 0, l_snow = 0, l_cold = 1;
  double dTi1_prev = c0;
  double dTsf_prev = c0, dfsens_dT = c0, dflat_dT = c0, dflwout_dT = c0;
  double Tin_init[NI], Tin_start[NI], Tsn_init[NS], Tsn_start[NS], etas[NS], kh[NMAT];
  const double dt_rhoi_hlyr = dt / (rhoi * hilyr); /* :1488 */
  col->fcondbot = c0;
  if (hslyr > hs_min / (double)NS) l_snow = 1;
  for (int k = 0; k < NS; k++) {
    Tsn_init[k] = Tsn_start[k] = Tsn[k];
    etas[k] = l_snow ? dt / (rhos * cp_ice * hslyr) : c0;
  }
  for (int k = 0; k < NI; k++) Tin_init[k] = Tin_start[k] = Tin[k];

  { /* conductivity :2221-2293 */
    double kilyr[NI], kslyr[NS];
    for (int k = 0; k < NS; k++) kslyr[k] = ksno;
    for (int k = 0; k < NI; k++) {
      if (c->conduct == 0)
        kilyr[k] = kice + betak * salin[k] / dmin(-puny, Tin[k]);
      else
        kilyr[k] = (2.11 - 0.011 * Tin[k] + 0.09 * salin[k] / dmin(-puny, Tin[k])) * rhoi / 917.0;
      kilyr[k] = dmax(kilyr[k], kimin);
    }
    if (l_snow) {
      kh[0] = c2 * kslyr[0] / hslyr;
      kh[NS] = c2 * kslyr[NS - 1] * kilyr[0] / (kslyr[NS - 1] * hilyr + kilyr[0] * hslyr);
    } else {
      kh[0] = c0;
      kh[NS] = c2 * kilyr[0] / hilyr;
    }
    kh[NS + NI] = c2 * kilyr[NI - 1] / hilyr;
    for (int k = 2; k <= NS; k++)
      kh[k - 1] = l_snow ? c2 * kslyr[k - 2] * kslyr[k - 1] / ((kslyr[k - 2] + kslyr[k - 1]) * hslyr)
                         : c0;
    for (int k = 2; k <= NI; k++)
      kh[k + NS - 1] = c2 * kilyr[k - 2] * kilyr[k - 1] / ((kilyr[k - 2] + kilyr[k - 1]) * hilyr);
  }

  { /* SW overshoot limiter :1541-1596 */
    double frac = 0.9, dTemp = 0.02;
    for (int k = 0; k < NI; k++) {
      double Iswabs_tmp = c0, ci;
      if (Tin_init[k] <= Tmlt[k] - dTemp) {
        if (c->l_brine) {
          ci = cp_ice - Lfresh * Tmlt[k] / (Tin_init[k] * Tin_init[k]);
          Iswabs_tmp = dmin(f->Iswabs[k], frac * (Tmlt[k] - Tin_init[k]) * ci / dt_rhoi_hlyr);
        } else {
          ci = cp_ice;
          Iswabs_tmp = dmin(f->Iswabs[k], frac * (-Tin_init[k]) * ci / dt_rhoi_hlyr);
        }
      }
      if (Iswabs_tmp < puny) Iswabs_tmp = c0;
      double dswabs = dmin(f->Iswabs[k] - Iswabs_tmp, f->fswint);
      f->fswsfc = f->fswsfc + dswabs;
      f->fswint = f->fswint - dswabs;
      f->Iswabs[k] = Iswabs_tmp;
    }
    for (int k = 0; k < NS; k++)
      if (l_snow) {
        double Sswabs_tmp = c0;
        if (Tsn_init[k] <= -dTemp) Sswabs_tmp = dmin(f->Sswabs[k], -frac * Tsn_init[k] / etas[k]);
        if (f->Sswabs[k] < puny) Sswabs_tmp = c0;
        double dswabs = dmin(f->Sswabs[k] - Sswabs_tmp, f->fswint);
        f->fswsfc = f->fswsfc + dswabs;
        f->fswint = f->fswint - dswabs;
        f->Sswabs[k] = Sswabs_tmp;
      }
  }
  f->fswabsn = f->fswsfc + f->fswint + f->fswthrun; /* :1605 */

  int niter_done = 0;
  for (int niter = 1; niter <= nitermax && !converged; niter++) {
    niter_done = niter;
    double etai[NI], sbdiag[NMAT], diag[NMAT], spdiag[NMAT], rhs[NMAT], Tmat[NMAT];
    double dfsurf_dT = c0, avg_Tsi = c0, enew = c0, Tsf_start = c0, dTsf = c0, avg_Tsf;
    double dTmat[NI], dqmat[NI];
    int reduce_kh[NI];
    converged = 1;
    for (int k = 0; k < NI; k++) { /* :1669-1684 */
      double ci = c->l_brine ? cp_ice - Lfresh * Tmlt[k] / (Tin[k] * Tin_init[k]) : cp_ice;
      etai[k] = dt_rhoi_hlyr / ci;
    }
    if (calc) { /* surface_fluxes :2389-2421 */
      double TsfK = col->Tsf + Tffresh;
      double tmpvar = c1 / TsfK;
      double qsat = qqqice * exp(-TTTice * tmpvar);
      double Qsfc = qsat / f->rhoa;
      double dQsfcdT = TTTice * tmpvar * tmpvar * Qsfc;
      double flwdabs = emissivity * f->flw;
      /* x**4 and x**3 as amdflang 22 (the compiler of oracle/_ref) evaluates integer
       * powers: left to right, ((x*x)*x)*x -- 1 ulp from (x*x)*(x*x) in a third of cases */
      double T3 = (TsfK * TsfK) * TsfK;
      f->flwoutn = -emissivity * stefan_boltzmann * (T3 * TsfK);
      f->fsensn = f->shcoef * (f->potT - TsfK);
      f->flatn = f->lhcoef * (f->Qa - Qsfc);
      dflwout_dT = -emissivity * stefan_boltzmann * c4 * T3;
      dfsens_dT = -f->shcoef;
      dflat_dT = -f->lhcoef * dQsfcdT;
      f->fsurfn = f->fswsfc + flwdabs + f->flwoutn + f->fsensn + f->flatn;
      dfsurf_dT = dflwout_dT + dfsens_dT + dflat_dT;
    }
    if (calc) { /* :1719-1738 */
      f->fcondtopn = l_snow ? kh[0] * (col->Tsf - Tsn[0]) : kh[NS] * (col->Tsf - Tin[0]);
      if (f->fsurfn < f->fcondtopn) col->Tsf = dmin(col->Tsf, -puny);
      Tsf_start = col->Tsf;
      l_cold = (col->Tsf <= -puny);
    }

    /* get_matrix_elements_calc_Tsfc :2540-2751 / _know_Tsfc :2871-3048 (rows 0-based here) */
    for (int k = 0; k <= NS; k++) {
      sbdiag[k] = c0; diag[k] = c1; spdiag[k] = c0; rhs[k] = c0;
    }
    if (!calc) {
      if (l_snow) { /* :2892-2902 */
        sbdiag[1] = c0;
        spdiag[1] = -etas[0] * kh[1];
        diag[1] = c1 + etas[0] * kh[1];
        rhs[1] = Tsn_init[0] + etas[0] * f->Sswabs[0] + etas[0] * f->fcondtopn;
      }
    } else if (l_cold) {
      int kr = l_snow ? 0 : NS;
      sbdiag[kr] = c0;
      diag[kr] = dfsurf_dT - kh[kr];
      spdiag[kr] = kh[kr];
      rhs[kr] = dfsurf_dT * col->Tsf - f->fsurfn;
    }
    if (calc && l_snow) {
      if (l_cold) {
        sbdiag[1] = -etas[0] * kh[0];
        spdiag[1] = -etas[0] * kh[1];
        diag[1] = c1 + etas[0] * (kh[0] + kh[1]);
        rhs[1] = Tsn_init[0] + etas[0] * f->Sswabs[0];
      } else {
        sbdiag[1] = c0;
        spdiag[1] = -etas[0] * kh[1];
        diag[1] = c1 + etas[0] * (kh[0] + kh[1]);
        rhs[1] = Tsn_init[0] + etas[0] * kh[0] * col->Tsf + etas[0] * f->Sswabs[0];
      }
    }
    for (int k = 2; k <= NS; k++)
      if (l_snow) {
        int kr = k; /* 0-based row of snow layer k */
        sbdiag[kr] = -etas[k - 1] * kh[k - 1];
        spdiag[kr] = -etas[k - 1] * kh[k];
        diag[kr] = c1 + etas[k - 1] * (kh[k - 1] + kh[k]);
        rhs[kr] = Tsn_init[k - 1] + etas[k - 1] * f->Sswabs[k - 1];
      }
    { /* top ice layer (nilyr > 1) */
      int k = NS, kr = NS + 1; /* kh[k] above, kh[k+1] below */
      if (!calc && !l_snow) { /* :2956-2962 */
        sbdiag[kr] = c0;
        spdiag[kr] = -etai[0] * kh[k + 1];
        diag[kr] = c1 + etai[0] * kh[k + 1];
        rhs[kr] = Tin_init[0] + etai[0] * f->Iswabs[0] + etai[0] * f->fcondtopn;
      } else if (!calc || l_snow || l_cold) {
        sbdiag[kr] = -etai[0] * kh[k];
        spdiag[kr] = -etai[0] * kh[k + 1];
        diag[kr] = c1 + etai[0] * (kh[k] + kh[k + 1]);
        rhs[kr] = Tin_init[0] + etai[0] * f->Iswabs[0];
      } else {
        sbdiag[kr] = c0;
        spdiag[kr] = -etai[0] * kh[k + 1];
        diag[kr] = c1 + etai[0] * (kh[k] + kh[k + 1]);
        rhs[kr] = Tin_init[0] + etai[0] * f->Iswabs[0] + etai[0] * kh[k] * col->Tsf;
      }
    }
    { /* bottom ice layer */
      int ki = NI - 1, k = NI - 1 + NS, kr = k + 1;
      sbdiag[kr] = -etai[ki] * kh[k];
      spdiag[kr] = c0;
      diag[kr] = c1 + etai[ki] * (kh[k] + kh[k + 1]);
      rhs[kr] = Tin_init[ki] + etai[ki] * f->Iswabs[ki] + etai[ki] * kh[k + 1] * f->Tbot;
    }
    for (int ki = 1; ki < NI - 1; ki++) { /* interior ice layers */
      int k = ki + NS, kr = k + 1;
      sbdiag[kr] = -etai[ki] * kh[k];
      spdiag[kr] = -etai[ki] * kh[k + 1];
      diag[kr] = c1 + etai[ki] * (kh[k] + kh[k + 1]);
      rhs[kr] = Tin_init[ki] + etai[ki] * f->Iswabs[ki];
    }
    { /* tridiag_solver :3119-3143 */
      double wgamma[NMAT], wbeta = diag[0];
      Tmat[0] = rhs[0] / wbeta;
      for (int k = 1; k < NMAT; k++) {
        wgamma[k] = spdiag[k - 1] / wbeta;
        wbeta = diag[k] - sbdiag[k] * wgamma[k];
        Tmat[k] = (rhs[k] - sbdiag[k] * Tmat[k - 1]) / wbeta;
      }
      for (int k = NMAT - 2; k >= 0; k--) Tmat[k] = Tmat[k] - wgamma[k + 1] * Tmat[k + 1];
    }
    avg_Tsf = c0;
    if (calc) { /* :1824-1884 */
      col->Tsf = l_cold ? (l_snow ? Tmat[0] : Tmat[NS]) : c0;
      dTsf = col->Tsf - Tsf_start;
      if (col->Tsf > puny) {
        col->Tsf = c0;
        dTsf = -Tsf_start;
        if (c->l_brine) avg_Tsi = c1;
        converged = 0;
      } else if (niter > 1 && Tsf_start <= -puny && fabs(dTsf) > puny && fabs(dTsf_prev) > puny &&
                 -dTsf / (dTsf_prev + puny * puny) > p5) {
        if (c->l_brine) {
          avg_Tsf = c1;
          avg_Tsi = c1;
        }
        dTsf = p5 * dTsf;
        converged = 0;
      }
      col->Tsf = col->Tsf + avg_Tsf * p5 * (Tsf_start - col->Tsf);
    }
    for (int k = 0; k < NS; k++) { /* :1890-1924 */
      Tsn[k] = l_snow ? Tmat[k + 1] : c0;
      if (c->l_brine) Tsn[k] = dmin(Tsn[k], c0);
      Tsn[k] = Tsn[k] + avg_Tsi * p5 * (Tsn_start[k] - Tsn[k]);
      qsn[k] = -rhos * (Lfresh - cp_ice * Tsn[k]);
      enew = enew + hslyr * qsn[k];
      Tsn_start[k] = Tsn[k];
    }
    for (int k = 0; k < NI; k++) { /* :1926-2001 */
      dTmat[k] = c0; dqmat[k] = c0; reduce_kh[k] = 0;
      Tin[k] = Tmat[k + 1 + NS];
      if (c->l_brine && Tin[k] > Tmlt[k] - puny) {
        dTmat[k] = Tin[k] - Tmlt[k];
        dqmat[k] = rhoi * dTmat[k] * (cp_ice - Lfresh * Tmlt[k] / (Tin[k] * Tin[k]));
        Tin[k] = Tmlt[k];
        reduce_kh[k] = 1;
      }
      if (k == 0 && !calc) { /* condition 2b :1961-1975 */
        double dTi1 = Tin[k] - Tin_start[k];
        if (niter > 1 && fabs(dTi1) > puny && fabs(dTi1_prev) > puny &&
            -dTi1 / (dTi1_prev + puny * puny) > p5) {
          if (c->l_brine) avg_Tsi = c1;
          dTi1 = p5 * dTi1;
          converged = 0;
        }
        dTi1_prev = dTi1;
      }
      Tin[k] = Tin[k] + avg_Tsi * p5 * (Tin_start[k] - Tin[k]);
      if (c->l_brine)
        qin[k] = -rhoi * (cp_ice * (Tmlt[k] - Tin[k]) + Lfresh * (c1 - Tmlt[k] / Tin[k]) -
                          cp_ocn * Tmlt[k]);
      else
        qin[k] = -rhoi * (-cp_ice * Tin[k] + Lfresh);
      enew = enew + hilyr * (qin[k] - dqmat[k]);
      Tin_start[k] = Tin[k];
    }
    if (calc) { /* :2017-2038 */
      if (fabs(dTsf) > Tsf_errmax) converged = 0;
      f->fsurfn = f->fsurfn + dTsf * dfsurf_dT;
      f->fcondtopn = l_snow ? kh[0] * (col->Tsf - Tsn[0]) : kh[NS] * (col->Tsf - Tin[0]);
      if (col->Tsf > -puny && f->fsurfn < f->fcondtopn) converged = 0;
      dTsf_prev = dTsf;
    }
    /* :2053-2073 */
    col->fcondbot = kh[NS + NI] * (Tin[NI - 1] - f->Tbot);
    double ferr = fabs((enew - col->einit) / dt - (f->fcondtopn - col->fcondbot + f->fswint));
    if (ferr > 0.9 * ferrmax) {
      converged = 0;
      for (int k = 1; k <= NI; k++)
        if (reduce_kh[k - 1] && dqmat[k - 1] > c0) {
          double frac = dmax(0.5 * (c1 - ferr / fabs(f->fcondtopn - col->fcondbot)), p1);
          kh[k + NS] = kh[k + NS] * frac;
          kh[k + NS - 1] = kh[k + NS] * frac;
        }
    }
  }
  orc_iter_hist[niter_done]++;
  if (calc) { /* :2136-2145 */
    f->flwoutn = f->flwoutn + dTsf_prev * dflwout_dT;
    f->fsensn = f->fsensn + dTsf_prev * dfsens_dT;
    f->flatn = f->flatn + dTsf_prev * dflat_dT;
  }
  return converged;
}

typedef struct {
  double fbot, fsnow, fhocnn, evapn, meltt, melts, meltb, congel, snoice, mlt_onset, frz_onset;
} growth;

/* thickness_changes :3622-4224, freeboard :4244-4377, adjust_enthalpy :4396-4492 */
static void thickness_changes(const orc_thermo_cfg *c, double dt, double yday, column *col,
                              const fluxes *f, growth *g) {
  const double qbotmax = -p5 * rhoi * Lfresh;
  double dzi[NI], dzs[NS], *qin = col->qin, *qsn = col->qsn;
  double esub, econ, etop_mlt, ebot_mlt, ebot_gro, wk1, dhi, dhs, qbot, qsub, hqtot;
  col->hsn_new = c0;
  for (int k = 0; k < NI; k++) dzi[k] = col->hilyr;
  for (int k = 0; k < NS; k++) dzs[k] = col->hslyr;
  if (!c->l_brine) { /* :3780-3812 */
    for (int k = 0; k < NS; k++) {
      double Ts = (Lfresh + qsn[k] / rhos) / cp_ice;
      if (Ts > c0) {
        dhs = cp_ice * Ts * dzs[k] / Lfresh;
        dzs[k] = dzs[k] - dhs;
        qsn[k] = -rhos * Lfresh;
      }
    }
    for (int k = 0; k < NI; k++) {
      double Ti = (Lfresh + qin[k] / rhoi) / cp_ice;
      if (Ti > c0) {
        dhi = cp_ice * Ti * dzi[k] / Lfresh;
        dzi[k] = dzi[k] - dhi;
        qin[k] = -rhoi * Lfresh;
      }
    }
  }
  /* :3823-3885 */
  wk1 = -f->flatn * dt;
  esub = dmax(wk1, c0);
  econ = dmin(wk1, c0);
  wk1 = (f->fsurfn - f->fcondtopn) * dt;
  etop_mlt = dmax(wk1, c0);
  wk1 = (col->fcondbot - g->fbot) * dt;
  ebot_mlt = dmax(wk1, c0);
  ebot_gro = dmin(wk1, c0);
  g->evapn = c0;
  if (col->hsn > puny) {
    dhs = econ / (qsn[0] - rhos * Lvap);
    dzs[0] = dzs[0] + dhs;
    g->evapn = g->evapn + dhs * rhos;
  } else {
    dhi = econ / (qin[0] - rhoi * Lvap);
    dzi[0] = dzi[0] + dhi;
    g->evapn = g->evapn + dhi * rhoi;
  }
  if (c->heat_capacity) {
    if (c->l_brine) {
      qbot = -rhoi * (cp_ice * (c->Tmlt[NI] - f->Tbot) + Lfresh * (c1 - c->Tmlt[NI] / f->Tbot) -
                      cp_ocn * c->Tmlt[NI]);
      qbot = dmin(qbot, qbotmax);
    } else
      qbot = -rhoi * (cp_ice * f->Tbot + Lfresh);
  } else
    qbot = -rhoi * Lfresh;
  dhi = ebot_gro / qbot;
  hqtot = dzi[NI - 1] * qin[NI - 1] + dhi * qbot;
  dzi[NI - 1] = dzi[NI - 1] + dhi;
  if (dzi[NI - 1] > puny) qin[NI - 1] = hqtot / dzi[NI - 1];
  g->congel = g->congel + dhi;
  if (dhi > puny && g->frz_onset < puny) g->frz_onset = yday;
  for (int k = 0; k < NS; k++) { /* :3889-3923 */
    qsub = qsn[k] - rhos * Lvap;
    dhs = dmax(-dzs[k], esub / qsub);
    dzs[k] = dzs[k] + dhs;
    esub = esub - dhs * qsub;
    esub = dmax(esub, c0);
    g->evapn = g->evapn + dhs * rhos;
    dhs = dmax(-dzs[k], etop_mlt / qsn[k]);
    dzs[k] = dzs[k] + dhs;
    etop_mlt = etop_mlt - dhs * qsn[k];
    etop_mlt = dmax(etop_mlt, c0);
    if (dhs < -puny && g->mlt_onset < puny) g->mlt_onset = yday;
    g->melts = g->melts - dhs;
  }
  for (int k = 0; k < NI; k++) { /* :3925-3959 */
    qsub = qin[k] - rhoi * Lvap;
    dhi = dmax(-dzi[k], esub / qsub);
    dzi[k] = dzi[k] + dhi;
    esub = esub - dhi * qsub;
    esub = dmax(esub, c0);
    g->evapn = g->evapn + dhi * rhoi;
    dhi = dmax(-dzi[k], etop_mlt / qin[k]);
    dzi[k] = dzi[k] + dhi;
    etop_mlt = etop_mlt - dhi * qin[k];
    etop_mlt = dmax(etop_mlt, c0);
    if (dhi < -puny && g->mlt_onset < puny) g->mlt_onset = yday;
    g->meltt = g->meltt - dhi;
  }
  for (int k = NI - 1; k >= 0; k--) { /* :3961-3982 */
    dhi = dmax(-dzi[k], ebot_mlt / qin[k]);
    dzi[k] = dzi[k] + dhi;
    ebot_mlt = ebot_mlt - dhi * qin[k];
    ebot_mlt = dmax(ebot_mlt, c0);
    g->meltb = g->meltb - dhi;
  }
  for (int k = NS - 1; k >= 0; k--) { /* :3984-4000 */
    dhs = dmax(-dzs[k], ebot_mlt / qsn[k]);
    dzs[k] = dzs[k] + dhs;
    ebot_mlt = ebot_mlt - dhs * qsn[k];
    ebot_mlt = dmax(ebot_mlt, c0);
  }
  g->fhocnn = g->fbot + (esub + etop_mlt + ebot_mlt) / dt; /* :4010 */
  if (g->fsnow > c0) { /* :4031-4045 */
    col->hsn_new = g->fsnow / rhos * dt;
    double qsnew = -rhos * Lfresh;
    double hstot = dzs[0] + col->hsn_new;
    if (hstot > c0) {
      qsn[0] = (dzs[0] * qsn[0] + col->hsn_new * qsnew) / hstot;
      qsn[0] = dmin(qsn[0], -rhos * Lfresh);
      dzs[0] = hstot;
    }
  }
  col->hin = c0;
  col->hsn = c0;
  for (int k = 0; k < NI; k++) col->hin = col->hin + dzi[k];
  for (int k = 0; k < NS; k++) col->hsn = col->hsn + dzs[k];
  { /* freeboard :4315-4375 */
    double dhin = c0, dhsn = c0, hqs = c0;
    wk1 = col->hsn - col->hin * (rhow - rhoi) / rhos;
    if (wk1 > puny && col->hsn > puny) {
      dhsn = dmin(wk1 * rhoi / rhow, col->hsn);
      dhin = dhsn * rhos / rhoi;
    }
    for (int k = NS - 1; k >= 0; k--)
      if (dhin > puny) {
        dhs = dmin(dhsn, dzs[k]);
        col->hsn = col->hsn - dhs;
        dzs[k] = dzs[k] - dhs;
        dhsn = dhsn - dhs;
        dhsn = dmax(dhsn, c0);
        hqs = hqs + dhs * qsn[k];
      }
    if (dhin > puny) {
      wk1 = dzi[0] + dhin;
      col->hin = col->hin + dhin;
      qin[0] = (dzi[0] * qin[0] + hqs) / wk1;
      dzi[0] = wk1;
      g->snoice = g->snoice + dhin;
    }
  }
  /* :4096-4192 */
  if (col->hin > c0)
    col->hilyr = col->hin / (double)NI;
  else {
    col->hin = c0;
    col->hilyr = c0;
  }
  if (col->hsn > c0)
    col->hslyr = col->hsn / (double)NS;
  else {
    col->hsn = c0;
    col->hslyr = c0;
  }
  if (c->heat_capacity) {
    double zi1[NI + 1], zi2[NI + 1], hq[NI];
    zi1[0] = c0; zi1[NI] = col->hin; zi2[0] = c0; zi2[NI] = col->hin;
    for (int k = 0; k < NI - 1; k++) {
      zi1[k + 1] = zi1[k] + dzi[k];
      zi2[k + 1] = zi2[k] + col->hilyr;
    }
    double rhlyr = c0; /* adjust_enthalpy :4451-4490 */
    if (col->hin > puny) rhlyr = c1 / col->hilyr;
    for (int k2 = 0; k2 < NI; k2++) {
      hq[k2] = c0;
      for (int k1 = 0; k1 < NI; k1++) {
        double hovlp = dmin(zi1[k1 + 1], zi2[k2 + 1]) - dmax(zi1[k1], zi2[k2]);
        hovlp = dmax(hovlp, c0);
        hq[k2] = hq[k2] + hovlp * qin[k1];
      }
    }
    for (int k = 0; k < NI; k++) qin[k] = hq[k] * rhlyr;
  } else {
    qin[0] = -rhoi * Lfresh;
    qsn[0] = -rhos * Lfresh;
  }
#if ORC_NSLYR > 1
  {
    double zs1[NS + 1], zs2[NS + 1], hq[NS];
    zs1[0] = c0; zs1[NS] = col->hsn; zs2[0] = c0; zs2[NS] = col->hsn;
    for (int k = 0; k < NS - 1; k++) {
      zs1[k + 1] = zs1[k] + dzs[k];
      zs2[k + 1] = zs2[k] + col->hslyr;
    }
    double rhlyr = c0;
    if (col->hsn > puny) rhlyr = c1 / col->hslyr;
    for (int k2 = 0; k2 < NS; k2++) {
      hq[k2] = c0;
      for (int k1 = 0; k1 < NS; k1++) {
        double hovlp = dmin(zs1[k1 + 1], zs2[k2 + 1]) - dmax(zs1[k1], zs2[k2]);
        hovlp = dmax(hovlp, c0);
        hq[k2] = hq[k2] + hovlp * qsn[k1];
      }
    }
    for (int k = 0; k < NS; k++) qsn[k] = hq[k] * rhlyr;
  }
#endif
  /* :4199-4222 */
  col->efinal = -g->evapn * Lvap;
  g->evapn = g->evapn / dt;
  for (int k = 0; k < NS; k++) col->efinal = col->efinal + col->hslyr * qsn[k];
  for (int k = 0; k < NI; k++) col->efinal = col->efinal + col->hilyr * qin[k];
}

/* thermo_vertical :108-515 */
int orc_thermo_vertical(const orc_thermo_cfg *c, int nx, int ny, double dt, int icells,
                        const int32_t *indxi, const int32_t *indxj, double *aicen, double *trcrn,
                        double *vicen, double *vsnon, double *eicen, double *esnon,
                        const double *flw, const double *potT, const double *Qa,
                        const double *rhoa, const double *fsnow, const double *fbot,
                        const double *Tbot, const double *lhcoef, const double *shcoef,
                        double *fswsfc, double *fswint, double *fswthrun, double *Sswabs,
                        double *Iswabs, double *fsurfn, double *fcondtopn, double *fsensn,
                        double *flatn, double *fswabsn, double *flwoutn, double *evapn,
                        double *freshn, double *fsaltn, double *fhocnn, double *meltt,
                        double *melts, double *meltb, double *congel, double *snoice,
                        double *mlt_onset, double *frz_onset, double yday, int *istop,
                        int *jstop) {
  const size_t np = (size_t)nx * ny;
  double *Tsfcn = trcrn + (size_t)(c->nt_Tsfc - 1) * np;
  long best_key = -1; /* stage*icells + ij of the failure the reference reports */
  *istop = 0;
  *jstop = 0;
  if (!c->heat_capacity) return -1; /* zero-layer thermodynamics not restated */
  for (size_t q = 0; q < np; q++) { /* :299-329 */
    fsensn[q] = fswabsn[q] = flwoutn[q] = evapn[q] = c0;
    freshn[q] = fsaltn[q] = fhocnn[q] = c0;
    meltt[q] = meltb[q] = melts[q] = congel[q] = snoice[q] = c0;
    if (c->calc_Tsfc) flatn[q] = fsurfn[q] = fcondtopn[q] = c0; /* :321-329; else inputs */
  }
  for (int ij = 0; ij < icells; ij++) {
    const size_t q = (size_t)(indxj[ij] - 1) * nx + (indxi[ij] - 1);
    column col;
    fluxes f;
    growth g;
    double ei[NI], es[NS];
    for (int k = 0; k < NI; k++) ei[k] = eicen[k * np + q];
    for (int k = 0; k < NS; k++) es[k] = esnon[k * np + q];
    int stage = init_vertical_profile(c, aicen[q], vicen[q], vsnon[q], Tsfcn[q], ei, es, &col);
    if (stage != ST_OK) {
      long key = (long)stage * icells + ij;
      if (best_key < 0 || key < best_key) best_key = key;
      continue;
    }
    const double worki = col.hin, works = col.hsn;
    f.rhoa = rhoa[q]; f.flw = flw[q]; f.potT = potT[q]; f.Qa = Qa[q];
    f.shcoef = shcoef[q]; f.lhcoef = lhcoef[q]; f.Tbot = Tbot[q];
    f.fswsfc = fswsfc[q]; f.fswint = fswint[q]; f.fswthrun = fswthrun[q];
    for (int k = 0; k < NS; k++) f.Sswabs[k] = Sswabs[k * np + q];
    for (int k = 0; k < NI; k++) f.Iswabs[k] = Iswabs[k * np + q];
    f.fsurfn = fsurfn[q]; f.fcondtopn = fcondtopn[q]; f.flatn = flatn[q];
    f.fsensn = f.fswabsn = f.flwoutn = c0;
    int conv = temperature_changes(c, dt, &col, &f);
    /* inout fields are written back even for a failing column (the reference has
     * modified them by the time it stops) */
    fswsfc[q] = f.fswsfc; fswint[q] = f.fswint;
    for (int k = 0; k < NS; k++) Sswabs[k * np + q] = f.Sswabs[k];
    for (int k = 0; k < NI; k++) Iswabs[k * np + q] = f.Iswabs[k];
    fsurfn[q] = f.fsurfn; fcondtopn[q] = f.fcondtopn; fsensn[q] = f.fsensn; flatn[q] = f.flatn;
    fswabsn[q] = f.fswabsn; flwoutn[q] = f.flwoutn;
    if (!conv) {
      long key = (long)ST_NOCONV * icells + ij;
      if (best_key < 0 || key < best_key) best_key = key;
      continue;
    }
    g.fbot = fbot[q]; g.fsnow = fsnow[q];
    g.meltt = g.melts = g.meltb = g.congel = g.snoice = c0;
    g.mlt_onset = mlt_onset[q]; g.frz_onset = frz_onset[q];
    thickness_changes(c, dt, yday, &col, &f, &g);
    fhocnn[q] = g.fhocnn; evapn[q] = g.evapn; meltt[q] = g.meltt; melts[q] = g.melts;
    meltb[q] = g.meltb; congel[q] = g.congel; snoice[q] = g.snoice;
    mlt_onset[q] = g.mlt_onset; frz_onset[q] = g.frz_onset;
    { /* conservation_check_vthermo :4573-4610 */
      double einp = (f.fsurfn - f.flatn + f.fswint - g.fhocnn - g.fsnow * Lfresh) * dt;
      double ferr = fabs(col.efinal - col.einit - einp) / dt;
      if (ferr > ferrmax) {
        long key = (long)ST_ECONS * icells + ij;
        if (best_key < 0 || key < best_key) best_key = key;
        continue;
      }
    }
    { /* :474-485 */
      double dhi = col.hin - worki, dhs = col.hsn - works;
      freshn[q] = g.evapn - (rhoi * dhi + rhos * (dhs - col.hsn_new)) / dt;
      fsaltn[q] = -rhoi * dhi * ice_ref_salinity * p001 / dt;
    }
    /* update_state_vthermo :4699-4745 (Tf dummy = Tbot, :496) */
    if (col.hin > c0) {
      vicen[q] = aicen[q] * col.hin;
      vsnon[q] = aicen[q] * col.hsn;
      Tsfcn[q] = col.Tsf;
      for (int k = 0; k < NI; k++) eicen[k * np + q] = col.qin[k] * vicen[q] / (double)NI;
      for (int k = 0; k < NS; k++) esnon[k * np + q] = col.qsn[k] * vsnon[q] / (double)NS;
    } else {
      aicen[q] = vicen[q] = vsnon[q] = c0;
      Tsfcn[q] = Tbot[q];
      for (int k = 0; k < NI; k++) eicen[k * np + q] = c0;
      for (int k = 0; k < NS; k++) esnon[k * np + q] = c0;
    }
  }
  if (best_key >= 0) {
    int ij = (int)(best_key % icells);
    *istop = indxi[ij];
    *jstop = indxj[ij];
    return 1;
  }
  return 0;
}

/* frzmlt_bottom_lateral :605-824.  cpchr = -cp_ocn*rhow*chio: chio is the constant 0.006 in the stand-alone build and a
 * namelist variable in the AusCOM build (:57-59, :673-694); orc_set_chio(0.006) gives the same double either way. */
static double g_chio = 0.006;
void orc_set_chio(double chio) { g_chio = chio; }

void orc_frzmlt_bottom_lateral(const orc_thermo_cfg *c, int nx, int ny, int ilo, int ihi, int jlo,
                               int jhi, double dt, const double *aice, const double *frzmlt,
                               const double *eicen, const double *esnon, const double *sst,
                               const double *Tf, const double *strocnxT, const double *strocnyT,
                               double *Tbot, double *fbot, double *rside) {
  const size_t np = (size_t)nx * ny;
  const double cpchr = -cp_ocn * rhow * g_chio;
  const double floediam = 300.0, alpha = 0.66, m1 = 1.6e-6, m2 = 1.36;
  for (size_t q = 0; q < np; q++) {
    rside[q] = c0;
    Tbot[q] = Tf[q];
    fbot[q] = c0;
  }
  for (int j = jlo; j <= jhi; j++)
    for (int i = ilo; i <= ihi; i++) {
      const size_t q = (size_t)(j - 1) * nx + (i - 1);
      if (!(aice[q] > puny && frzmlt[q] < c0)) continue;
      double fside = c0;
      double deltaT = dmax(sst[q] - Tbot[q], c0);
      double ustar = sqrt(sqrt(strocnxT[q] * strocnxT[q] + strocnyT[q] * strocnyT[q]) / rhow);
      ustar = dmax(ustar, c->ustar_min);
      fbot[q] = cpchr * deltaT * ustar;
      fbot[q] = dmax(fbot[q], frzmlt[q]);
      double wlat = m1 * pow(deltaT, m2);
      rside[q] = wlat * dt * pi / (alpha * floediam);
      rside[q] = dmax(c0, dmin(rside[q], c1));
      for (int n = 0; n < ORC_NCAT; n++) {
        double etot = c0;
        for (int k = 0; k < NS; k++) etot = etot + esnon[(size_t)(n * NS + k) * np + q];
        for (int k = 0; k < NI; k++) etot = etot + eicen[(size_t)(n * NI + k) * np + q];
        fside = fside + rside[q] * etot / dt;
      }
      double xtmp = frzmlt[q] / (fbot[q] + fside + puny);
      xtmp = dmin(xtmp, c1);
      fbot[q] = fbot[q] * xtmp;
      rside[q] = rside[q] * xtmp;
    }
}

/* merge_fluxes, source/ice_flux.F90:613-762: area-weighted accumulation of one category's
 * fluxes into the cell aggregates, over the compressed list of the category. */
void orc_merge_fluxes(int nx, int ny, int icells, const int32_t *indxi, const int32_t *indxj,
                      const double *aicen, const double *flw, const double *const catn[20],
                      double *const acc[20]) {
  /* order of catn / acc: strairx, strairy, fsurf, fcondtop, fsens, flat, fswabs, flwout, evap,
   * Tref, Qref, fresh, fsalt, fhocn, fswthru, meltt, meltb, melts, congel, snoice */
  (void)ny;
  for (int ij = 0; ij < icells; ij++) {
    const size_t q = (size_t)(indxj[ij] - 1) * nx + (indxi[ij] - 1);
    for (int k = 0; k < 20; k++) {
      if (k == 7)
        acc[k][q] = acc[k][q] + (catn[k][q] - (c1 - emissivity) * flw[q]) * aicen[q];
      else
        acc[k][q] = acc[k][q] + catn[k][q] * aicen[q];
    }
  }
}
