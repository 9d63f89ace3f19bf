/* TEST INFRASTRUCTURE ONLY (see cice_oracle.h).  CPU restatement of the
 * reference EVP dynamics, source/ice_dyn_evp.F90 (citations are file:line
 * under /root/reference).  Build with -O2 -ffp-contract=off (no FMA
 * contraction, no fast-math) so that results are bitwise those of the
 * reference built the same way. */
#include "cice_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* drivers/cice4/ice_constants.F90:49-61,132-179 */
static const double rhos = 330.0, rhoi = 917.0, rhow = 1026.0;
/* The AusCOM / coupled build of the reference (-DAusCOM -Dcoupled, bld/Macros.nci:56-57) makes the ocean turning angle
 * and the ice-ocean drag namelist variables (ice_dyn_evp.F90:91-97, ice_init.F90:258-264), rotates with the hemisphere
 * (:910-913, :1402-1408, :1524-1536) and takes the sea-surface tilt from the ocean model when use_ocnslope is set
 * (:919-933).  g_aus = 0 is the stand-alone build: the compile-time constants and expressions of that build. */
static int g_aus = 0, g_ocnslope = 0;
static double dragio = 0.00536, cosw = 1.0, sinw = 0.0;
void orc_set_auscom(int on, double cosw_, double sinw_, double dragio_, int use_ocnslope) {
  g_aus = on;
  cosw = on ? cosw_ : 1.0;
  sinw = on ? sinw_ : 0.0;
  dragio = on ? dragio_ : 0.00536;
  g_ocnslope = on ? use_ocnslope : 0;
}
static const double gravit = 9.80616;
static const double puny = 1.0e-11;
static const double c0 = 0.0, c1 = 1.0, c2 = 2.0, c4 = 4.0, p5 = 0.5, p25 = 0.25;
#define P166 (1.0 / 6.0)
#define P333 (1.0 / 3.0)
#define P111 (1.0 / 9.0)
#define P222 (2.0 / 9.0)
#define P055 (P111 * 0.5)
#define P027 (P055 * 0.5)
/* ice_dyn_evp.F90:76-88 */
#define DRAGW (dragio * rhow)
static const double eyc = 0.36, a_min = 0.001, m_min = 0.01;

#define IX(i, j) ((size_t)((j)-1) * nx + ((i)-1))
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* ice_dyn_evp.F90:535-577 */
void orc_set_evp_parameters(double dt, int ndte, int evp_damping, orc_evp_params *p) {
  double dte = dt / (double)ndte;
  p->ndte = ndte;
  p->evp_damping = evp_damping;
  p->dtei = c1 / dte;
  double ecc = c4;
  p->ecci = p25;
  double tdamp2 = c2 * eyc * dt;
  p->dte2T = dte / tdamp2;
  p->denom1 = c1 / (c1 + p->dte2T);
  p->denom2 = c1 / (c1 + p->dte2T * ecc);
  p->rcon = 1230.0 * eyc * dt * (p->dtei * p->dtei);
}

/* ice_dyn_evp.F90:586-694 */
void orc_evp_prep1(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *aice,
                   const double *vice, const double *vsno, const int32_t *tmask,
                   const double *strairxT, const double *strairyT, double *strairx,
                   double *strairy, double *tmass, int32_t *icetmask) {
  unsigned char *tmphm = (unsigned char *)malloc((size_t)nx * ny);
  for (int j = 1; j <= ny; j++)
    for (int i = 1; i <= nx; i++) {
      size_t q = IX(i, j);
      tmass[q] = tmask[q] ? (rhoi * vice[q] + rhos * vsno[q]) : c0;
      tmphm[q] = tmask[q] && (aice[q] > a_min) && (tmass[q] > m_min);
      strairx[q] = strairxT[q];
      strairy[q] = strairyT[q];
      icetmask[q] = 0;
    }
  for (int j = jlo; j <= jhi; j++)
    for (int i = ilo; i <= ihi; i++) {
      int any = 0;
      for (int dj = -1; dj <= 1; dj++)
        for (int di = -1; di <= 1; di++) any |= tmphm[IX(i + di, j + dj)];
      if (any) icetmask[IX(i, j)] = 1;
      if (!tmask[IX(i, j)]) icetmask[IX(i, j)] = 0;
    }
  free(tmphm);
}

/* ice_dyn_evp.F90:703-938 (non-coupled, non-AusCOM branches) */
void orc_evp_prep2(const orc_evp_params *p, int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   int *icellt, int *icellu, int32_t *indxti, int32_t *indxtj, int32_t *indxui,
                   int32_t *indxuj, const double *aiu, const double *umass, double *umassdtei,
                   const double *fcor, const int32_t *umask, const double *uocn,
                   const double *vocn, const double *strairx, const double *strairy,
                   const double *ss_tltx, const double *ss_tlty, const int32_t *icetmask,
                   int32_t *iceumask, double *fm, double *strtltx, double *strtlty,
                   double *strocnx, double *strocny, double *strintx, double *strinty,
                   double *waterx, double *watery, double *forcex, double *forcey,
                   double *const sig[12], double *uvel, double *vvel) {

  for (int j = 1; j <= ny; j++)
    for (int i = 1; i <= nx; i++) {
      size_t q = IX(i, j);
      waterx[q] = watery[q] = forcex[q] = forcey[q] = umassdtei[q] = c0;
      if (icetmask[q] == 0)
        for (int k = 0; k < 12; k++) sig[k][q] = c0;
    }
  int nt = 0;
  for (int j = jlo; j <= jhi + 1; j++)
    for (int i = ilo; i <= ihi + 1; i++)
      if (icetmask[IX(i, j)] == 1) {
        indxti[nt] = i;
        indxtj[nt] = j;
        nt++;
      }
  *icellt = nt;
  int nu = 0;
  for (int j = jlo; j <= jhi; j++)
    for (int i = ilo; i <= ihi; i++) {
      size_t q = IX(i, j);
      int old = iceumask[q] != 0;
      int now = umask[q] && (aiu[q] > a_min) && (umass[q] > m_min);
      iceumask[q] = now;
      if (now) {
        indxui[nu] = i;
        indxuj[nu] = j;
        nu++;
        if (!old) {
          uvel[q] = uocn[q];
          vvel[q] = vocn[q];
        }
      } else {
        uvel[q] = vvel[q] = c0;
        strintx[q] = strinty[q] = c0;
        strocnx[q] = strocny[q] = c0;
      }
    }
  *icellu = nu;
  for (int ij = 0; ij < nu; ij++) {
    size_t q = IX(indxui[ij], indxuj[ij]);
    umassdtei[q] = umass[q] * p->dtei;
    fm[q] = fcor[q] * umass[q];
    if (g_aus) { /* :910-913: direction of rotation depends on the hemisphere; sign(1., fm): +1 for +-0 ... */
      const double sg = signbit(fm[q]) ? -1.0 : 1.0;
      waterx[q] = uocn[q] * cosw - vocn[q] * sinw * sg;
      watery[q] = vocn[q] * cosw + uocn[q] * sinw * sg;
    } else {
      waterx[q] = uocn[q] * cosw - vocn[q] * sinw;
      watery[q] = vocn[q] * cosw + uocn[q] * sinw;
    }
    if (g_aus && g_ocnslope) { /* coupled, :923-925 */
      strtltx[q] = -gravit * umass[q] * ss_tltx[q];
      strtlty[q] = -gravit * umass[q] * ss_tlty[q];
    } else { /* :919-922, and AusCOM without use_ocnslope :928-933 */
      strtltx[q] = -fm[q] * vocn[q];
      strtlty[q] = fm[q] * uocn[q];
    }
    forcex[q] = strairx[q] + strtltx[q];
    forcey[q] = strairy[q] + strtlty[q];
  }
}

/* ice_dyn_evp.F90:947-1293 */
void orc_stress(const orc_evp_params *p, int nx, int ny, int ksub, int icellt,
                const int32_t *indxti, const int32_t *indxtj, const double *uvel,
                const double *vvel, const double *dxt, const double *dyt, const double *dxhy,
                const double *dyhx, const double *cxp, const double *cyp, const double *cxm,
                const double *cym, const double *tarear, const double *tinyarea,
                const double *strength, double *const sig[12], double *shear, double *divu,
                double *prs_sig, double *rdg_conv, double *rdg_shear, double *str) {
  const size_t np = (size_t)nx * ny;
  const double ecci = p->ecci, dte2T = p->dte2T, denom1 = p->denom1, denom2 = p->denom2;
  double *sp1 = sig[0], *sp2 = sig[1], *sp3 = sig[2], *sp4 = sig[3];
  double *sm1 = sig[4], *sm2 = sig[5], *sm3 = sig[6], *sm4 = sig[7];
  double *s121 = sig[8], *s122 = sig[9], *s123 = sig[10], *s124 = sig[11];
  /* The two `omp` pragmas of this file are active only in the -fopenmp build (libcice_oracle_omp.so),
   * which bench.py's all-cores cpu_baseline leg uses; iterations are independent (every cell writes
   * only its own entries), so the results do not depend on the thread count. */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
  for (long z = 0; z < (long)(8 * np); z++) str[z] = 0.0;
#else
  memset(str, 0, 8 * np * sizeof(double)); /* :1051 */
#endif
#pragma omp parallel for schedule(static)
  for (int ij = 0; ij < icellt; ij++) {
    const int i = indxti[ij], j = indxtj[ij];
    const size_t q = IX(i, j);
    const double u_ne = uvel[IX(i, j)], u_nw = uvel[IX(i - 1, j)], u_sw = uvel[IX(i - 1, j - 1)],
                 u_se = uvel[IX(i, j - 1)];
    const double v_ne = vvel[IX(i, j)], v_nw = vvel[IX(i - 1, j)], v_sw = vvel[IX(i - 1, j - 1)],
                 v_se = vvel[IX(i, j - 1)];
    const double Cxp = cxp[q], Cyp = cyp[q], Cxm = cxm[q], Cym = cym[q], Dxt = dxt[q],
                 Dyt = dyt[q];
    /* :1065-1092 */
    double divune = Cyp * u_ne - Dyt * u_nw + Cxp * v_ne - Dxt * v_se;
    double divunw = Cym * u_nw + Dyt * u_ne + Cxp * v_nw - Dxt * v_sw;
    double divusw = Cym * u_sw + Dyt * u_se + Cxm * v_sw + Dxt * v_nw;
    double divuse = Cyp * u_se - Dyt * u_sw + Cxm * v_se + Dxt * v_ne;
    double tensionne = -Cym * u_ne - Dyt * u_nw + Cxm * v_ne + Dxt * v_se;
    double tensionnw = -Cyp * u_nw + Dyt * u_ne + Cxm * v_nw + Dxt * v_sw;
    double tensionsw = -Cyp * u_sw + Dyt * u_se + Cxp * v_sw - Dxt * v_nw;
    double tensionse = -Cym * u_se - Dyt * u_sw + Cxp * v_se - Dxt * v_ne;
    double shearne = -Cym * v_ne - Dyt * v_nw - Cxm * u_ne - Dxt * u_se;
    double shearnw = -Cyp * v_nw + Dyt * v_ne - Cxm * u_nw - Dxt * u_sw;
    double shearsw = -Cyp * v_sw + Dyt * v_se - Cxp * u_sw + Dxt * u_nw;
    double shearse = -Cym * v_se - Dyt * v_sw - Cxp * u_se + Dxt * u_ne;
    /* :1095-1098 */
    double Deltane = sqrt(divune * divune + ecci * (tensionne * tensionne + shearne * shearne));
    double Deltanw = sqrt(divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw));
    double Deltase = sqrt(divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse));
    double Deltasw = sqrt(divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw));
    /* :1103-1115 */
    if (ksub == p->ndte) {
      divu[q] = p25 * (divune + divunw + divuse + divusw) * tarear[q];
      double tmp = p25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear[q];
      rdg_conv[q] = -dmin(divu[q], c0);
      rdg_shear[q] = p5 * (tmp - fabs(divu[q]));
      double ts = tensionne + tensionnw + tensionse + tensionsw;
      double ss = shearne + shearnw + shearse + shearsw;
      shear[q] = p25 * tarear[q] * sqrt(ts * ts + ss * ss);
    }
    /* :1121-1141 */
    double c0ne, c0nw, c0sw, c0se;
    const double st = strength[q], ta = tinyarea[q];
    if (p->evp_damping) {
      c0ne = dmin(st / dmax(Deltane, c4 * ta), p->rcon);
      c0nw = dmin(st / dmax(Deltanw, c4 * ta), p->rcon);
      c0sw = dmin(st / dmax(Deltasw, c4 * ta), p->rcon);
      c0se = dmin(st / dmax(Deltase, c4 * ta), p->rcon);
      prs_sig[q] = st * Deltane / dmax(Deltane, c4 * ta);
    } else {
      c0ne = st / dmax(Deltane, ta);
      c0nw = st / dmax(Deltanw, ta);
      c0sw = st / dmax(Deltasw, ta);
      c0se = st / dmax(Deltase, ta);
      prs_sig[q] = c0ne * Deltane;
    }
    double c1ne = c0ne * dte2T, c1nw = c0nw * dte2T, c1sw = c0sw * dte2T, c1se = c0se * dte2T;
    /* :1148-1165 */
    double p1 = sp1[q] = (sp1[q] + c1ne * (divune - Deltane)) * denom1;
    double p2 = sp2[q] = (sp2[q] + c1nw * (divunw - Deltanw)) * denom1;
    double p3 = sp3[q] = (sp3[q] + c1sw * (divusw - Deltasw)) * denom1;
    double p4 = sp4[q] = (sp4[q] + c1se * (divuse - Deltase)) * denom1;
    double m1 = sm1[q] = (sm1[q] + c1ne * tensionne) * denom2;
    double m2 = sm2[q] = (sm2[q] + c1nw * tensionnw) * denom2;
    double m3 = sm3[q] = (sm3[q] + c1sw * tensionsw) * denom2;
    double m4 = sm4[q] = (sm4[q] + c1se * tensionse) * denom2;
    double t1 = s121[q] = (s121[q] + c1ne * shearne * p5) * denom2;
    double t2 = s122[q] = (s122[q] + c1nw * shearnw * p5) * denom2;
    double t3 = s123[q] = (s123[q] + c1sw * shearsw * p5) * denom2;
    double t4 = s124[q] = (s124[q] + c1se * shearse * p5) * denom2;
    /* :1196-1239 */
    double ssigpn = p1 + p2, ssigps = p3 + p4, ssigpe = p1 + p4, ssigpw = p2 + p3;
    double ssigp1 = (p1 + p3) * P055, ssigp2 = (p2 + p4) * P055;
    double ssigmn = m1 + m2, ssigms = m3 + m4, ssigme = m1 + m4, ssigmw = m2 + m3;
    double ssigm1 = (m1 + m3) * P055, ssigm2 = (m2 + m4) * P055;
    double ssig12n = t1 + t2, ssig12s = t3 + t4, ssig12e = t1 + t4, ssig12w = t2 + t3;
    double ssig121 = (t1 + t3) * P111, ssig122 = (t2 + t4) * P111;
    double csigpne = P111 * p1 + ssigp2 + P027 * p3;
    double csigpnw = P111 * p2 + ssigp1 + P027 * p4;
    double csigpsw = P111 * p3 + ssigp2 + P027 * p1;
    double csigpse = P111 * p4 + ssigp1 + P027 * p2;
    double csigmne = P111 * m1 + ssigm2 + P027 * m3;
    double csigmnw = P111 * m2 + ssigm1 + P027 * m4;
    double csigmsw = P111 * m3 + ssigm2 + P027 * m1;
    double csigmse = P111 * m4 + ssigm1 + P027 * m2;
    double csig12ne = P222 * t1 + ssig122 + P055 * t3;
    double csig12nw = P222 * t2 + ssig121 + P055 * t4;
    double csig12sw = P222 * t3 + ssig122 + P055 * t1;
    double csig12se = P222 * t4 + ssig121 + P055 * t2;
    double str12ew = p5 * Dxt * (P333 * ssig12e + P166 * ssig12w);
    double str12we = p5 * Dxt * (P333 * ssig12w + P166 * ssig12e);
    double str12ns = p5 * Dyt * (P333 * ssig12n + P166 * ssig12s);
    double str12sn = p5 * Dyt * (P333 * ssig12s + P166 * ssig12n);
    const double Dxhy = dxhy[q], Dyhx = dyhx[q];
    /* :1244-1289 */
    double strp_tmp = p25 * Dyt * (P333 * ssigpn + P166 * ssigps);
    double strm_tmp = p25 * Dyt * (P333 * ssigmn + P166 * ssigms);
    str[0 * np + q] = -strp_tmp - strm_tmp - str12ew + Dxhy * (-csigpne + csigmne) + Dyhx * csig12ne;
    str[1 * np + q] = strp_tmp + strm_tmp - str12we + Dxhy * (-csigpnw + csigmnw) + Dyhx * csig12nw;
    strp_tmp = p25 * Dyt * (P333 * ssigps + P166 * ssigpn);
    strm_tmp = p25 * Dyt * (P333 * ssigms + P166 * ssigmn);
    str[2 * np + q] = -strp_tmp - strm_tmp + str12ew + Dxhy * (-csigpse + csigmse) + Dyhx * csig12se;
    str[3 * np + q] = strp_tmp + strm_tmp + str12we + Dxhy * (-csigpsw + csigmsw) + Dyhx * csig12sw;
    strp_tmp = p25 * Dxt * (P333 * ssigpe + P166 * ssigpw);
    strm_tmp = p25 * Dxt * (P333 * ssigme + P166 * ssigmw);
    str[4 * np + q] = -strp_tmp + strm_tmp - str12ns - Dyhx * (csigpne + csigmne) + Dxhy * csig12ne;
    str[5 * np + q] = strp_tmp - strm_tmp - str12sn - Dyhx * (csigpse + csigmse) + Dxhy * csig12se;
    strp_tmp = p25 * Dxt * (P333 * ssigpw + P166 * ssigpe);
    strm_tmp = p25 * Dxt * (P333 * ssigmw + P166 * ssigme);
    str[6 * np + q] = -strp_tmp + strm_tmp + str12ns - Dyhx * (csigpnw + csigmnw) + Dxhy * csig12nw;
    str[7 * np + q] = strp_tmp - strm_tmp + str12sn - Dyhx * (csigpsw + csigmsw) + Dxhy * csig12sw;
  }
}

/* ice_dyn_evp.F90:1302-1443 */
void orc_stepu(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
               const double *aiu, const double *str, const double *uocn, const double *vocn,
               const double *waterx, const double *watery, const double *forcex,
               const double *forcey, const double *umassdtei, const double *fm,
               const double *uarear, double *strocnx, double *strocny, double *strintx,
               double *strinty, double *uvel, double *vvel) {
  const size_t np = (size_t)nx * ny;
#pragma omp parallel for schedule(static)
  for (int ij = 0; ij < icellu; ij++) {
    const int i = indxui[ij], j = indxuj[ij];
    const size_t q = IX(i, j);
    double uold = uvel[q], vold = vvel[q];
    double du = uocn[q] - uold, dv = vocn[q] - vold;
    double vrel = aiu[q] * DRAGW * sqrt(du * du + dv * dv);
    double taux = vrel * waterx[q], tauy = vrel * watery[q];
    double cca = umassdtei[q] + vrel * cosw;
    double ccb = (g_aus && fm[q] < 0.) ? fm[q] - vrel * sinw : fm[q] + vrel * sinw; /* :1402-1411 */
    double ab2 = cca * cca + ccb * ccb;
    strintx[q] = uarear[q] * (str[0 * np + IX(i, j)] + str[1 * np + IX(i + 1, j)] +
                              str[2 * np + IX(i, j + 1)] + str[3 * np + IX(i + 1, j + 1)]);
    strinty[q] = uarear[q] * (str[4 * np + IX(i, j)] + str[5 * np + IX(i, j + 1)] +
                              str[6 * np + IX(i + 1, j)] + str[7 * np + IX(i + 1, j + 1)]);
    double cc1 = strintx[q] + forcex[q] + taux + umassdtei[q] * uold;
    double cc2 = strinty[q] + forcey[q] + tauy + umassdtei[q] * vold;
    uvel[q] = (cca * cc1 + ccb * cc2) / ab2;
    vvel[q] = (cca * cc2 - ccb * cc1) / ab2;
    strocnx[q] = taux;
    strocny[q] = tauy;
  }
}

/* ice_dyn_evp.F90:1452-1549 */
void orc_evp_finish(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
                    const double *uvel, const double *vvel, const double *uocn,
                    const double *vocn, const double *aiu, double *strocnx, double *strocny,
                    double *strocnxT, double *strocnyT) {
  orc_evp_finish_fm(nx, ny, icellu, indxui, indxuj, uvel, vvel, uocn, vocn, aiu, NULL, strocnx, strocny, strocnxT,
                    strocnyT);
}

/* the AusCOM build passes fm as well (:1458-1460) */
void orc_evp_finish_fm(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
                       const double *uvel, const double *vvel, const double *uocn, const double *vocn,
                       const double *aiu, const double *fm, double *strocnx, double *strocny,
                       double *strocnxT, double *strocnyT) {
  memset(strocnxT, 0, sizeof(double) * nx * ny);
  memset(strocnyT, 0, sizeof(double) * nx * ny);
  for (int ij = 0; ij < icellu; ij++) {
    const size_t q = IX(indxui[ij], indxuj[ij]);
    double du = uocn[q] - uvel[q], dv = vocn[q] - vvel[q];
    double vrel = DRAGW * sqrt(du * du + dv * dv);
    if (g_aus && fm && fm[q] < 0.) { /* :1524-1531 rotate to the opposite direction in the Southern Hemisphere */
      strocnx[q] = strocnx[q] - vrel * (uvel[q] * cosw + vvel[q] * sinw) * aiu[q];
      strocny[q] = strocny[q] - vrel * (vvel[q] * cosw - uvel[q] * sinw) * aiu[q];
    } else {
      strocnx[q] = strocnx[q] - vrel * (uvel[q] * cosw - vvel[q] * sinw) * aiu[q];
      strocny[q] = strocny[q] - vrel * (vvel[q] * cosw + uvel[q] * sinw) * aiu[q];
    }
    strocnxT[q] = strocnx[q] / aiu[q];
    strocnyT[q] = strocny[q] / aiu[q];
  }
}

/* ice_mechred.F90:1869-2036 with asum_ridging :573 and ridge_itd :773-1098 */
void orc_ice_strength(int kstrength, int krdg_partic, int krdg_redist, double mu_rdg, int nx,
                      int ny, int ilo, int ihi, int jlo, int jhi, int icells,
                      const int32_t *indxi, const int32_t *indxj, const double *aice,
                      const double *vice, const double *aice0, const double *aicen,
                      const double *vicen, double *strength) {
  const size_t np = (size_t)nx * ny;
  const double Cf = 17.0, Cp = p5 * gravit * (rhow - rhoi) * rhoi / rhow;
  const double Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0;
  const double Pstar = 2.75e4, Cstar = 20.0;
  const double Gstari = c1 / Gstar, astari = c1 / astar;
  memset(strength, 0, sizeof(double) * np);
  if (kstrength != 1) { /* Hibler 79, :2026-2030 */
    for (int j = jlo; j <= jhi; j++)
      for (int i = ilo; i <= ihi; i++)
        strength[IX(i, j)] = Pstar * vice[IX(i, j)] * exp(-Cstar * (c1 - aice[IX(i, j)]));
    return;
  }
  for (int ij = 0; ij < icells; ij++) {
    const size_t q = IX(indxi[ij], indxj[ij]);
    double Gsum[ORC_NCAT + 2]; /* index n+1 for n=-1..ncat */
    double apartic[ORC_NCAT + 1], hrmin[ORC_NCAT + 1], hrmax[ORC_NCAT + 1], hrexp[ORC_NCAT + 1],
        krdg[ORC_NCAT + 1];
    Gsum[0] = c0;
    Gsum[1] = (aice0[q] > puny) ? aice0[q] : Gsum[0];
    apartic[0] = c0;
    for (int n = 1; n <= ORC_NCAT; n++) {
      double a = aicen[(size_t)(n - 1) * np + q];
      Gsum[n + 1] = (a > puny) ? Gsum[n] + a : Gsum[n];
      apartic[n] = c0;
      hrmin[n] = hrmax[n] = hrexp[n] = c0;
      krdg[n] = c1;
    }
    double work = c1 / Gsum[ORC_NCAT + 1];
    for (int n = 0; n <= ORC_NCAT; n++) Gsum[n + 1] = Gsum[n + 1] * work;
    if (krdg_partic == 0) {
      for (int n = 0; n <= ORC_NCAT; n++) {
        double g = Gsum[n + 1], gm = Gsum[n];
        if (g < Gstar)
          apartic[n] = Gstari * (g - gm) * (c2 - (gm + g) * Gstari);
        else if (gm < Gstar)
          apartic[n] = Gstari * (Gstar - gm) * (c2 - (gm + Gstar) * Gstari);
      }
    } else {
      double xtmp = c1 / (c1 - exp(-astari));
      for (int n = -1; n <= ORC_NCAT; n++) Gsum[n + 1] = exp(-Gsum[n + 1] * astari) * xtmp;
      for (int n = 0; n <= ORC_NCAT; n++) apartic[n] = Gsum[n] - Gsum[n + 1];
    }
    for (int n = 1; n <= ORC_NCAT; n++) {
      double a = aicen[(size_t)(n - 1) * np + q], v = vicen[(size_t)(n - 1) * np + q];
      if (a > puny) {
        double hi = v / a;
        if (krdg_redist == 0) {
          hrmin[n] = dmin(c2 * hi, hi + maxraft);
          hrmax[n] = c2 * sqrt(Hstar * hi);
          hrmax[n] = dmax(hrmax[n], hrmin[n] + puny);
          double hrmean = p5 * (hrmin[n] + hrmax[n]);
          krdg[n] = hrmean / hi;
        } else {
          hi = dmax(hi, puny);
          hrmin[n] = dmin(c2 * hi, hi + maxraft);
          hrexp[n] = mu_rdg * sqrt(hi);
          krdg[n] = (hrmin[n] + hrexp[n]) / hi;
        }
      }
    }
    double aksum = apartic[0];
    for (int n = 1; n <= ORC_NCAT; n++) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
    double s = c0;
    for (int n = 1; n <= ORC_NCAT; n++) {
      double a = aicen[(size_t)(n - 1) * np + q], v = vicen[(size_t)(n - 1) * np + q];
      if (a > puny && apartic[n] > c0) {
        double hi = v / a, h2rdg;
        if (krdg_redist == 0)
          h2rdg = P333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) /
                  (hrmax[n] - hrmin[n]);
        else
          h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
        double dh2rdg = -hi * hi + h2rdg / krdg[n];
        s = s + apartic[n] * dh2rdg;
      }
    }
    strength[q] = Cf * Cp * s / aksum;
  }
}

/* ice_grid.F90:1580-1633 (per block; caller zeroes nothing -- we zero here) */
void orc_to_ugrid(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *w1,
                  const double *tarea, const double *uarea, double *w2) {
  memset(w2, 0, sizeof(double) * nx * ny);
  for (int j = jlo; j <= jhi; j++)
    for (int i = ilo; i <= ihi; i++)
      w2[IX(i, j)] = p25 *
                     (w1[IX(i, j)] * tarea[IX(i, j)] + w1[IX(i + 1, j)] * tarea[IX(i + 1, j)] +
                      w1[IX(i, j + 1)] * tarea[IX(i, j + 1)] +
                      w1[IX(i + 1, j + 1)] * tarea[IX(i + 1, j + 1)]) /
                     uarea[IX(i, j)];
}

/* ice_grid.F90:1684-1732 (ghost cells of w2 are left as they are) */
void orc_to_tgrid(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *w1,
                  const double *tarea, const double *uarea, double *w2) {
  for (int j = jlo; j <= jhi; j++)
    for (int i = ilo; i <= ihi; i++)
      w2[IX(i, j)] = p25 *
                     (w1[IX(i, j)] * uarea[IX(i, j)] + w1[IX(i - 1, j)] * uarea[IX(i - 1, j)] +
                      w1[IX(i, j - 1)] * uarea[IX(i, j - 1)] +
                      w1[IX(i - 1, j - 1)] * uarea[IX(i - 1, j - 1)]) /
                     tarea[IX(i, j)];
}

/* serial/ice_boundary.F90:682-702: sources are physical cells, destinations ghost cells */
void orc_halo_r8(double *a, int ncopy, const int32_t *src, const int32_t *dst, int nfill,
                 const int32_t *fdst, double fill) {
  for (int n = 0; n < ncopy; n++) a[dst[n]] = a[src[n]];
  for (int n = 0; n < nfill; n++) a[fdst[n]] = fill;
}
void orc_halo_i4(int32_t *a, int ncopy, const int32_t *src, const int32_t *dst, int nfill,
                 const int32_t *fdst, int32_t fill) {
  for (int n = 0; n < ncopy; n++) a[dst[n]] = a[src[n]];
  for (int n = 0; n < nfill; n++) a[fdst[n]] = fill;
}

/* ---- whole evp(dt), ice_dyn_evp.F90:119-432 ------------------------------- */
typedef struct {
  int *icellt, *icellu;
  int32_t *ti, *tj, *ui, *uj;
  double *tmass, *waterx, *watery, *forcex, *forcey, *aiu, *umass, *umassdtei, *str, *work1;
  int32_t *icetmask;
} evp_work;

static void work_alloc(const orc_domain *d, evp_work *w) {
  size_t np = (size_t)d->nx * d->ny, n = np * d->nblocks;
  w->icellt = calloc(d->nblocks, sizeof(int));
  w->icellu = calloc(d->nblocks, sizeof(int));
  w->ti = malloc(n * 4); w->tj = malloc(n * 4); w->ui = malloc(n * 4); w->uj = malloc(n * 4);
  w->tmass = calloc(n, 8); w->waterx = calloc(n, 8); w->watery = calloc(n, 8);
  w->forcex = calloc(n, 8); w->forcey = calloc(n, 8); w->aiu = calloc(n, 8);
  w->umass = calloc(n, 8); w->umassdtei = calloc(n, 8); w->work1 = calloc(n, 8);
  w->str = calloc(8 * np, 8);
  w->icetmask = calloc(n, 4);
}
static void work_free(evp_work *w) {
  free(w->icellt); free(w->icellu); free(w->ti); free(w->tj); free(w->ui); free(w->uj);
  free(w->tmass); free(w->waterx); free(w->watery); free(w->forcex); free(w->forcey);
  free(w->aiu); free(w->umass); free(w->umassdtei); free(w->work1); free(w->str);
  free(w->icetmask);
}

static void halo8(const orc_domain *d, double *a) {
  orc_halo_r8(a, d->ncopy, d->hsrc, d->hdst, d->nfill, d->hfill, 0.0);
}

static void subcycle(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s, evp_work *w,
                     int ksub) {
  const size_t np = (size_t)d->nx * d->ny;
  for (int b = 0; b < d->nblocks; b++) {
    size_t o = b * np;
    double *sg[12];
    for (int k = 0; k < 12; k++) sg[k] = s->sig[k] + o;
    orc_stress(p, d->nx, d->ny, ksub, w->icellt[b], w->ti + o, w->tj + o, s->uvel + o,
               s->vvel + o, d->dxt + o, d->dyt + o, d->dxhy + o, d->dyhx + o, d->cxp + o,
               d->cyp + o, d->cxm + o, d->cym + o, d->tarear + o, d->tinyarea + o,
               s->strength + o, sg, s->shear + o, s->divu + o, s->prs_sig + o, s->rdg_conv + o,
               s->rdg_shear + o, w->str);
    orc_stepu(d->nx, d->ny, w->icellu[b], w->ui + o, w->uj + o, w->aiu + o, w->str, s->uocn + o,
              s->vocn + o, w->waterx + o, w->watery + o, w->forcex + o, w->forcey + o,
              w->umassdtei + o, s->fm + o, d->uarear + o, s->strocnx + o, s->strocny + o,
              s->strintx + o, s->strinty + o, s->uvel + o, s->vvel + o);
  }
  halo8(d, s->uvel);
  halo8(d, s->vvel);
}

static void evp_prepare(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s,
                        evp_work *w) {
  const int nx = d->nx, ny = d->ny;
  const size_t np = (size_t)nx * ny, n = np * d->nblocks;
  /* :214-244 */
  memset(s->rdg_conv, 0, n * 8); memset(s->rdg_shear, 0, n * 8); memset(s->divu, 0, n * 8);
  memset(s->shear, 0, n * 8); memset(s->prs_sig, 0, n * 8);
  for (int b = 0; b < d->nblocks; b++) {
    size_t o = b * np;
    orc_evp_prep1(nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], s->aice + o, s->vice + o,
                  s->vsno + o, d->tmask + o, s->strairxT + o, s->strairyT + o, s->strairx + o,
                  s->strairy + o, w->tmass + o, w->icetmask + o);
  }
  orc_halo_i4(w->icetmask, d->ncopy, d->hsrc, d->hdst, d->nfill, d->hfill, 0); /* :250 */
  for (int b = 0; b < d->nblocks; b++) { /* :259-260 */
    size_t o = b * np;
    orc_to_ugrid(nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], w->tmass + o, d->tarea + o,
                 d->uarea + o, w->umass + o);
    orc_to_ugrid(nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], s->aice + o, d->tarea + o,
                 d->uarea + o, w->aiu + o);
  }
  /* t2ugrid_vector(strairx), (strairy): :276-277, ice_grid.F90:1540 */
  for (int c = 0; c < 2; c++) {
    double *f = c ? s->strairy : s->strairx;
    memcpy(w->work1, f, n * 8);
    halo8(d, w->work1);
    for (int b = 0; b < d->nblocks; b++) {
      size_t o = b * np;
      orc_to_ugrid(nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], w->work1 + o,
                   d->tarea + o, d->uarea + o, f + o);
    }
  }
  for (int b = 0; b < d->nblocks; b++) { /* :280-334 */
    size_t o = b * np;
    double *sg[12];
    for (int k = 0; k < 12; k++) sg[k] = s->sig[k] + o;
    orc_evp_prep2(p, nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], &w->icellt[b],
                  &w->icellu[b], w->ti + o, w->tj + o, w->ui + o, w->uj + o, w->aiu + o,
                  w->umass + o, w->umassdtei + o, d->fcor + o, d->umask + o, s->uocn + o,
                  s->vocn + o, s->strairx + o, s->strairy + o, s->ss_tltx + o, s->ss_tlty + o,
                  w->icetmask + o, s->iceumask + o, s->fm + o, s->strtltx + o, s->strtlty + o,
                  s->strocnx + o, s->strocny + o, s->strintx + o, s->strinty + o, w->waterx + o,
                  w->watery + o, w->forcex + o, w->forcey + o, sg, s->uvel + o, s->vvel + o);
    orc_ice_strength(d->kstrength, d->krdg_partic, d->krdg_redist, d->mu_rdg, nx, ny, d->ilo[b],
                     d->ihi[b], d->jlo[b], d->jhi[b], w->icellt[b], w->ti + o, w->tj + o,
                     s->aice + o, s->vice + o, s->aice0 + o, s->aicen + (size_t)b * ORC_NCAT * np,
                     s->vicen + (size_t)b * ORC_NCAT * np, s->strength + o);
  }
  if (d->perturb_strength_ulp)
    for (size_t q = 0; q < n; q++)
      if (s->strength[q] != 0.0)
        s->strength[q] = nextafter(s->strength[q], ((q * 2654435761u) >> 7) & 1 ? INFINITY : -INFINITY);
  halo8(d, s->strength); /* :336-344 */
  halo8(d, s->uvel);
  halo8(d, s->vvel);
}

void orc_evp(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s) {
  const int nx = d->nx, ny = d->ny;
  const size_t np = (size_t)nx * ny, n = np * d->nblocks;
  evp_work w;
  work_alloc(d, &w);
  evp_prepare(d, p, s, &w);
  for (int ksub = 1; ksub <= p->ndte; ksub++) subcycle(d, p, s, &w, ksub); /* :347-404 */
  for (int b = 0; b < d->nblocks; b++) { /* :410-425 */
    size_t o = b * np;
    orc_evp_finish_fm(nx, ny, w.icellu[b], w.ui + o, w.uj + o, s->uvel + o, s->vvel + o, s->uocn + o,
                      s->vocn + o, w.aiu + o, s->fm + o, s->strocnx + o, s->strocny + o, s->strocnxT + o,
                      s->strocnyT + o);
  }
  for (int c = 0; c < 2; c++) { /* u2tgrid_vector :427-428, ice_grid.F90:1642 */
    double *f = c ? s->strocnyT : s->strocnxT;
    memcpy(w.work1, f, n * 8);
    halo8(d, w.work1);
    for (int b = 0; b < d->nblocks; b++) {
      size_t o = b * np;
      orc_to_tgrid(nx, ny, d->ilo[b], d->ihi[b], d->jlo[b], d->jhi[b], w.work1 + o, d->tarea + o,
                   d->uarea + o, f + o);
    }
  }
  if (s->aiu) memcpy(s->aiu, w.aiu, n * 8);
  if (s->umass) memcpy(s->umass, w.umass, n * 8);
  if (s->icetmask) memcpy(s->icetmask, w.icetmask, n * 4);
  work_free(&w);
}

/* cpu_baseline helper: prepare once, then time nsub subcycles (ksub never equals ndte
 * unless nsub >= ndte, exactly as inside evp).  Returns seconds. */
double orc_evp_subcycles_only(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s,
                              int nsub) {
  evp_work w;
  work_alloc(d, &w);
  evp_prepare(d, p, s, &w);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int ksub = 1; ksub <= nsub; ksub++) subcycle(d, p, s, &w, ksub);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  work_free(&w);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
