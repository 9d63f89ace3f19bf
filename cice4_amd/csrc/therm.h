// Column thermodynamics on the device (source/ice_therm_vertical.F90).
#pragma once
#include "common.h"

namespace cice {

struct ThermoParams {  // module state of ice_therm_vertical (:45-79) after init_thermo_vertical
  double salin[NILYR + 1], Tmlt[NILYR + 1];
  double ustar_min;
  int l_brine, heat_capacity, calc_Tsfc, conduct;
  int tr_iage, nt_Tsfc, nt_iage;
  void init(const cice_thermo_config& c);
};

// Pointers of one launch.  Index of category n, block b, plane k, cell q (0-based):
//   per-category 2-D:   ((b*ncat + n)*np + q)
//   trcrn:              (((b*ncat + n)*NTRCR + it)*np + q)
//   eicen / esnon:      ((b*ncat*NILYR + n*NILYR + k)*np + q)   (NSLYR for esnon)
//   Sswabs / Iswabs:    (((b*ncat + n)*NSLYR + k)*np + q)       (NILYR for Iswabs)
//   forcing, onsets:    (b*np + q)
// The single-call form uses ncat = 1, nblocks = 1 and an index list.

// The batched kernels count the columns they update.  One counter for the whole launch costs one atomic per
// wavefront on ONE address, and the memory side takes those one at a time: 9,715 of them at gx1 size = 132 us, the
// whole kernel's duration with nothing else in it (measured on a build that only loads and stores: 132 us with the
// counter, 56 us without).  So the counter is THERMO_COUNT_SLOTS words on different cache lines, picked by workgroup
// index; the host adds them up after the download of the status words.
constexpr int THERMO_COUNT_SLOTS = 64;
constexpr int THERMO_COUNT_STRIDE = 16;                                            // words (128 B)
constexpr int THERMO_STATUS_WORDS = THERMO_COUNT_STRIDE * (1 + THERMO_COUNT_SLOTS);  // [0] error key, [16 + 16 i] counters

struct ThermoArgs {
  ThermoParams p;
  int nx, ny, ncat, nblocks;
  double dt, yday;
  // list mode (reference signature): icells entries; dense mode: list == nullptr
  int icells;
  const int32_t *indxi, *indxj;
  const int32_t* blk;  // dense mode: ilo,ihi,jlo,jhi per block
  double *aicen, *trcrn, *vicen, *vsnon, *eicen, *esnon;
  const double *flw, *potT, *Qa, *rhoa, *fsnow, *fbot, *Tbot, *lhcoef, *shcoef;
  double *fswsfc, *fswint, *fswthrun, *Sswabs, *Iswabs;
  double *fsurfn, *fcondtopn, *fsensn, *flatn, *fswabsn, *flwoutn, *evapn, *freshn, *fsaltn, *fhocnn,
      *meltt, *melts, *meltb, *congel, *snoice, *mlt_onset, *frz_onset;
  unsigned long long* errkey;   // atomicMin target, initialised to ~0
  unsigned long long* nupdates; // dense mode: THERMO_COUNT_SLOTS counters, THERMO_COUNT_STRIDE words apart; their sum =
                                // the number of columns updated
  unsigned char* niter;         // dense mode (may be NULL): iterations the implicit solve of every column took
};

void thermo_launch_list(const ThermoArgs& a, hipStream_t s);
void thermo_launch_dense(const ThermoArgs& a, hipStream_t s);
size_t thermo_sorted_plane(size_t np, int chunk);   // entries of the permutation per (category, block) plane
void thermo_launch_sorted(const ThermoArgs& a, int chunk, int group, int32_t* perm, const double* tsfc, size_t tstride,
                          hipStream_t s);

struct MergeArgs {  // merge_fluxes, ice_flux.F90:613-762
  int nx, ny, ncat, nblocks;
  const int32_t* blk;
  const double *aicen_init, *flw;
  const double* src[20];  // per-category sources, (nx,ny,ncat,nb)
  double* acc[20];        // cumulative, (nx,ny,nb)
};
void merge_launch(const MergeArgs& a, hipStream_t s);

struct FrzmltArgs {
  int nx, ny, ilo, ihi, jlo, jhi;
  double dt, ustar_min;
  double chio;   // coupled flavour only (ice_therm_vertical.F90:57-60,692-694); the stand-alone build's constant 0.006
  const double *aice, *frzmlt, *eicen, *esnon, *sst, *Tf, *strocnxT, *strocnyT;
  double *Tbot, *fbot, *rside;
};
void frzmlt_launch(const FrzmltArgs& a, hipStream_t s);

}  // namespace cice
