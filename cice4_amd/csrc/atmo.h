// Atmosphere-ice (or -ocean) boundary layer on the device (source/ice_atmo.F90:56-384 atmo_boundary_layer):
// turbulent exchange coefficients, wind stress and the 2 m reference temperature / humidity -- the routine
// step_therm1 calls per category in front of thermo_vertical (drivers/cice4/CICE_RunMod.F90:402-425).
#pragma once
#include "common.h"

namespace cice {

struct AtmoParams {  // ice_constants.F90:53-121,178 and the two logarithms of constants the routine evaluates
  double vonkar, gravit, zvir, cp_air, cpvir, Tffresh, pih, zTrf, umin;
  double qqq[2], TTT[2], Lheat[2];  // [0] ice, [1] ocean: qsat coefficients, Lsub / Lvap
  double rdn_ice;                   // vonkar / log(zref / iceruf)   (:202)
  double al2;                       // log(zref / zTrf)              (:173)
  double zref;
  void init();                      // host libm for the two constant logarithms, as the reference's host does
};

// List form = the reference's signature (one block, one surface type); dense form = every category of every
// block in one launch: the cells of category n with aicen > puny on the physical domain (CICE_RunMod.F90:380-389),
// Tsf = trcrn(:,:,nt_Tsfc,n,iblk), outputs (nx,ny,ncat,nblocks) zero elsewhere.
struct AtmoArgs {
  AtmoParams p;
  int nx, ny, ncat, nblocks, ocn, calc_strair;
  int icells;
  const int32_t *indxi, *indxj;  // list form
  const int32_t* blk;            // dense form: ilo, ihi, jlo, jhi per block
  const double* aicen;           // dense form: (nx,ny,ncat,nb)
  const double* Tsf;             // list: (nx,ny); dense: trcrn (nx,ny,NTRCR,ncat,nb), plane it_Tsfc
  int it_Tsfc;
  const double *potT, *uatm, *vatm, *wind, *zlvl, *Qa, *rhoa;  // (nx,ny[,nb])
  const double *strax, *stray;   // dense form, calc_strair = F: data stresses copied to every category (:435-439)
  double *strx, *stry, *Tref, *Qref, *delt, *delq, *lhcoef, *shcoef;  // delt, delq may be NULL in dense form
};
void atmo_launch_list(const AtmoArgs& a, hipStream_t s);
void atmo_launch_dense(const AtmoArgs& a, hipStream_t s);

}  // namespace cice
