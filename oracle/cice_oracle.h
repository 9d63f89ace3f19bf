/* TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the
 * product (cice4_amd/).  Plain-C fp64 CPU restatement of the reference's hot
 * path, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_vs_ref.py)
 * against the compiled reference itself (oracle/_ref, built by oracle/build_ref.sh
 * from /root/reference) and against golden vectors minted from it
 * (tests/golden/, generator tests/golden/make_golden.py).
 *
 * Layout: a Fortran (nx,ny[,k]) array is addressed as a[(k*ny + (j-1))*nx + (i-1)]
 * with the reference's 1-based (i,j).  Index lists hold 1-based values.
 */
#ifndef CICE_ORACLE_H
#define CICE_ORACLE_H
#include <stdint.h>

#define ORC_NCAT 5
#define ORC_NILYR 4
#define ORC_NSLYR 1
#define ORC_MAX_NTRCR 5

typedef struct {
  double dtei, dte2T, denom1, denom2, rcon, ecci;
  int ndte, evp_damping;
} orc_evp_params;

typedef struct {
  double salin[ORC_NILYR + 1], Tmlt[ORC_NILYR + 1];
  double ustar_min;
  int l_brine, heat_capacity, calc_Tsfc, conduct; /* conduct: 0 MU71, 1 bubbly */
  int tr_iage, nt_Tsfc, nt_iage;                  /* 1-based tracer slots */
} orc_thermo_cfg;

void orc_set_evp_parameters(double dt, int ndte, int evp_damping, orc_evp_params *p);
/* on = 1: the -DAusCOM -Dcoupled build of the reference (turning angle and drag from the namelist, rotation by
 * hemisphere, sea-surface tilt from the ocean model when use_ocnslope); on = 0: the stand-alone build */
void orc_set_auscom(int on, double cosw, double sinw, double dragio, int use_ocnslope);
void orc_set_chio(double chio);

void orc_evp_prep1(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *aice,
                   const double *vice, const double *vsno, const int32_t *tmask,
                   const double *strairxT, const double *strairyT, double *strairx,
                   double *strairy, double *tmass, int32_t *icetmask);

void orc_evp_prep2(const orc_evp_params *p, int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   int *icellt, int *icellu, int32_t *indxti, int32_t *indxtj, int32_t *indxui,
                   int32_t *indxuj, const double *aiu, const double *umass, double *umassdtei,
                   const double *fcor, const int32_t *umask, const double *uocn,
                   const double *vocn, const double *strairx, const double *strairy,
                   const double *ss_tltx, const double *ss_tlty, const int32_t *icetmask,
                   int32_t *iceumask, double *fm, double *strtltx, double *strtlty,
                   double *strocnx, double *strocny, double *strintx, double *strinty,
                   double *waterx, double *watery, double *forcex, double *forcey,
                   double *const sig[12], double *uvel, double *vvel);

void orc_stress(const orc_evp_params *p, int nx, int ny, int ksub, int icellt,
                const int32_t *indxti, const int32_t *indxtj, const double *uvel,
                const double *vvel, const double *dxt, const double *dyt, const double *dxhy,
                const double *dyhx, const double *cxp, const double *cyp, const double *cxm,
                const double *cym, const double *tarear, const double *tinyarea,
                const double *strength, double *const sig[12], double *shear, double *divu,
                double *prs_sig, double *rdg_conv, double *rdg_shear, double *str);

void orc_evp_finish_fm(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
                       const double *uvel, const double *vvel, const double *uocn, const double *vocn,
                       const double *aiu, const double *fm, double *strocnx, double *strocny,
                       double *strocnxT, double *strocnyT);
void orc_stepu(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
               const double *aiu, const double *str, const double *uocn, const double *vocn,
               const double *waterx, const double *watery, const double *forcex,
               const double *forcey, const double *umassdtei, const double *fm,
               const double *uarear, double *strocnx, double *strocny, double *strintx,
               double *strinty, double *uvel, double *vvel);

void orc_evp_finish(int nx, int ny, int icellu, const int32_t *indxui, const int32_t *indxuj,
                    const double *uvel, const double *vvel, const double *uocn,
                    const double *vocn, const double *aiu, double *strocnx, double *strocny,
                    double *strocnxT, double *strocnyT);

void orc_ice_strength(int kstrength, int krdg_partic, int krdg_redist, double mu_rdg, int nx,
                      int ny, int ilo, int ihi, int jlo, int jhi, int icells,
                      const int32_t *indxi, const int32_t *indxj, const double *aice,
                      const double *vice, const double *aice0, const double *aicen,
                      const double *vicen, double *strength);

void orc_to_ugrid(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *work1,
                  const double *tarea, const double *uarea, double *work2);
void orc_to_tgrid(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *work1,
                  const double *tarea, const double *uarea, double *work2);

/* halo: ncopy local copies dst[n] <- src[n] (linear addresses into the
 * (nblocks,ny,nx) array); nfill cells set to `fill`. */
void orc_halo_r8(double *a, int ncopy, const int32_t *src, const int32_t *dst, int nfill,
                 const int32_t *fdst, double fill);
void orc_halo_i4(int32_t *a, int ncopy, const int32_t *src, const int32_t *dst, int nfill,
                 const int32_t *fdst, int32_t fill);

/* whole evp(dt) over nblocks local blocks: ice_dyn_evp.F90:119-432 */
typedef struct {
  int nx, ny, nblocks;
  const int32_t *ilo, *ihi, *jlo, *jhi;
  int ncopy;
  const int32_t *hsrc, *hdst;
  int nfill;
  const int32_t *hfill;
  /* grid, each (nblocks,ny,nx) */
  const double *dxt, *dyt, *dxhy, *dyhx, *cxp, *cyp, *cxm, *cym, *tarea, *uarea, *tarear,
      *uarear, *tinyarea, *fcor;
  const int32_t *tmask, *umask;
  int kstrength, krdg_partic, krdg_redist;
  double mu_rdg;
  /* sensitivity probe (tests only): move every computed strength value by one ulp, up or down
   * in a fixed pseudo-random pattern, before the subcycling -- the size of the libm (exp)
   * difference between any two hosts -- so that a test can measure how much the reference's own
   * result moves under it. 0 = off. */
  int perturb_strength_ulp;
} orc_domain;

typedef struct {
  /* in */
  const double *aice, *vice, *vsno, *aice0, *aicen, *vicen; /* aicen: (nblocks,ncat,ny,nx) */
  const double *strairxT, *strairyT, *uocn, *vocn, *ss_tltx, *ss_tlty;
  /* inout */
  double *uvel, *vvel, *sig[12];
  int32_t *iceumask;
  double *fm, *strtltx, *strtlty, *strocnx, *strocny, *strintx, *strinty;
  /* out */
  double *strairx, *strairy, *strength, *divu, *shear, *rdg_conv, *rdg_shear, *prs_sig,
      *strocnxT, *strocnyT;
  /* optional work exposure for tests (may be NULL) */
  double *aiu, *umass;
  int32_t *icetmask;
} orc_evp_state;

void orc_evp(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s);
/* only the ndte-subcycle loop (stress+stepu+2 halos), for timing: lists must be prepared */
double orc_evp_subcycles_only(const orc_domain *d, const orc_evp_params *p, orc_evp_state *s,
                              int nsub);

/* thermodynamics */
void orc_init_thermo(int heat_capacity, int calc_Tsfc, int conduct, double ustar_min,
                     orc_thermo_cfg *c);

/* thermo_vertical, ice_therm_vertical.F90:108-515; same argument order. Returns l_stop. */
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
                        int *jstop);

void orc_frzmlt_bottom_lateral(const orc_thermo_cfg *c, int nx, int ny, int ilo, int ihi, int jlo,
                               int jhi, double dt, const double *aice, const double *frzmlt,
                               const double *eicen, const double *esnon, const double *sst,
                               const double *Tf, const double *strocnxT, const double *strocnyT,
                               double *Tbot, double *fbot, double *rside);

/* merge_fluxes, ice_flux.F90:613-762; catn/acc order: strairx, strairy, fsurf, fcondtop, fsens,
 * flat, fswabs, flwout, evap, Tref, Qref, fresh, fsalt, fhocn, fswthru, meltt, meltb, melts,
 * congel, snoice */
void orc_merge_fluxes(int nx, int ny, int icells, const int32_t *indxi, const int32_t *indxj,
                      const double *aicen, const double *flw, const double *const catn[20],
                      double *const acc[20]);

#endif
