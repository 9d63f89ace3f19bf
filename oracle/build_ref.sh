#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the checker, never the product.
#
# Compile the reference's own Fortran hot-path modules (EVP dynamics + vertical
# thermodynamics, the horizontal transport and the modules they `use`) WHERE THEY LIE under
# /root/reference, plus our C-ABI capture wrapper oracle/ref_capi.F90, into
#   oracle/_ref/libcice_ref_<cfg>.so
# No reference source is copied, patched or stubbed: every file below compiles
# unmodified with amdflang and no external library (netCDF use in
# ice_read_write.F90 is guarded by `#ifdef ncdf`, which we leave undefined;
# ice_forcing.F90 -- the only file with unguarded netCDF calls -- is not in the
# closure of the hot path).  The reference's build system (comp_ice, bld/) is
# not run.
#
# usage: oracle/build_ref.sh <cfg> <NXGLOB> <NYGLOB> <BLCKX> <BLCKY> <MXBLCKS>
# Flags follow bld/Macros.ubuntu:20-22 (-fdefault-real-8, big-endian unformatted
# I/O) -- the hot path has un-kinded literals that depend on the promotion
# (ice_therm_vertical.F90:1541,2067).
set -euo pipefail
REF=${CICE_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
CFG=$1; NXG=$2; NYG=$3; BX=$4; BY=$5; MXB=$6
FC=${FC:-/opt/rocm/bin/amdflang}
OUT=$HERE/_ref
OBJ=$OUT/obj_$CFG
mkdir -p "$OBJ"
if [ ! -d "$REF/source" ]; then
  echo "build_ref: $REF not present (GPU box?) -- using prebuilt files in $OUT" >&2
  exit 0
fi
FFLAGS="-O2 -fPIC -w -cpp -fdefault-real-8 -fconvert=big-endian -ffp-contract=off \
 -DLINUX -DNXGLOB=$NXG -DNYGLOB=$NYG -DBLCKX=$BX -DBLCKY=$BY -DMXBLCKS=$MXB \
 -J $OBJ -I $OBJ"
# dependency order (module/use topological sort of the hot-path closure)
SRCS="source/ice_kinds_mod.F90 serial/ice_communicate.F90 source/ice_domain_size.F90
 source/ice_fileunits.F90 serial/ice_exit.F90 source/ice_blocks.F90
 drivers/cice4/ice_constants.F90 source/ice_spacecurve.F90 source/ice_distribution.F90
 serial/ice_global_reductions.F90 serial/ice_boundary.F90 serial/ice_broadcast.F90
 source/ice_domain.F90 source/ice_state.F90 source/ice_flux.F90
 serial/ice_gather_scatter.F90 source/ice_work.F90 source/ice_read_write.F90
 serial/ice_timers.F90 source/ice_grid.F90 source/ice_itd.F90 source/ice_mechred.F90
 source/ice_dyn_evp.F90 source/ice_calendar.F90 source/ice_atmo.F90 source/ice_ocean.F90
 source/ice_restart.F90 source/ice_age.F90 source/ice_therm_vertical.F90
 source/ice_transport_remap.F90 source/ice_transport_driver.F90"
# DROPIN=1: the same closure, but with OUR drop-in modules (cice4_amd/fortran/rccl/ice_boundary.F90,
# ice_dyn_evp.F90 and ice_therm_vertical.F90, which forward ice_HaloUpdate / evp(dt) /
# thermo_vertical(...) to the GPU library through the ISO_C_BINDING shim) in place of the reference's
# files of the same name -> libcice_dropin_<cfg>.so.  Every caller in the closure (ice_domain,
# ice_grid, ice_state, ...) and the capture wrapper are the reference's / the same: this is the
# drop-in test.
DROPIN=${DROPIN:-0}
# MPI=1: the reference's mpi/ modules (ice_communicate, ice_exit, ice_global_reductions, ice_broadcast,
# ice_gather_scatter, ice_timers and -- unless DROPIN=1 replaces it -- ice_boundary) instead of serial/,
# compiled against the image's MPICH through `include 'mpif.h'` -> libcice_<kind>mpi_<cfg>.so.
# A process that loads it is a 1-rank MPI job (singleton MPI_Init); with DROPIN=1 this is the build an
# MPI user of the reference would make (our ice_boundary then also sets up the RCCL communicator).
MPI=${MPI:-0}
MPIROOT=${MPIROOT:-/opt/conda}
# AUS=1: the hot-path modules compiled the way COSIMA's production builds compile them (-DAusCOM -Dcoupled,
# bld/Macros.nci:56-57) with the constants of drivers/access-om/ice_constants.F90: namelist turning angle / drag / chio,
# rotation by hemisphere, sea-surface tilt from the ocean model, mom4's cp_ocn and ice salinity.  The macros go to
# ice_constants, ice_dyn_evp, ice_therm_vertical and ice_atmo only; the set-up and I/O modules of the closure keep the
# stand-alone form (their AusCOM branches need netCDF and the coupler) -> libcice_<kind>aus_<cfg>.so
AUS=${AUS:-0}
KIND=ref
if [ "$DROPIN" = "1" ]; then
  KIND=dropin
  OBJ=$OUT/obj_${CFG}_dropin
  rm -rf "$OBJ"            # our modules change: never reuse dependents' objects / .mod files
  mkdir -p "$OBJ"
  FFLAGS="${FFLAGS//obj_$CFG/obj_${CFG}_dropin}"
fi
if [ "$AUS" = "1" ]; then
  OBJ0=$OBJ
  KIND=${KIND}aus
  OBJ=${OBJ}_aus
  rm -rf "$OBJ"; mkdir -p "$OBJ"
  FFLAGS="${FFLAGS//$OBJ0/$OBJ}"
  # constants: drivers/access-om/ice_constants.F90 (cp_ocn, ice_ref_salinity, ... as COSIMA runs them); the two data
  # modules of the coupler right behind it
  SRCS="${SRCS/drivers\/cice4\/ice_constants.F90/drivers/access-om/ice_constants.F90 drivers/access-om/cpl_parameters.F90 drivers/access-om/cpl_arrays_setup.F90}"
  # (the macro goes to the files named below only: with it ice_read_write.F90, ice_grid.F90 and ice_calendar.F90 turn to
  # netCDF / the coupler, which this image does not have -- and the hot path does not need)
fi
if [ "$MPI" = "1" ]; then
  OBJ0=$OBJ
  KIND=${KIND}mpi
  OBJ=${OBJ}_mpi
  rm -rf "$OBJ"; mkdir -p "$OBJ"
  FFLAGS="${FFLAGS//$OBJ0/$OBJ} -I$MPIROOT/include -DCICE4_AMD_MPI"
  SRCS="${SRCS//serial\//mpi/}"
fi
# our own sources unchanged since the last build of this variant: nothing to do (the variants that swap
# in our modules are otherwise rebuilt from scratch, see above)
TARGET="$OUT/libcice_${KIND}_$CFG.so"
if [ "$DROPIN" = "1" ] || [ "$MPI" = "1" ] || [ "$AUS" = "1" ]; then
  if [ -f "$TARGET" ] && [ -z "$(find "$HERE/ref_capi.F90" "$HERE/build_ref.sh" "$HERE/../cice4_amd/fortran" \
        -newer "$TARGET" -name '*.F90' -o -newer "$TARGET" -name '*.sh' 2>/dev/null | head -1)" ]; then
    echo "up to date $TARGET"
    exit 0
  fi
fi
OBJS=""
for s in $SRCS; do
  src="$REF/$s"
  o=$OBJ/$(basename "${s%.F90}").o
  FX=""
  if [ "$AUS" = "1" ]; then
    case "$(basename $s)" in
      ice_constants.F90|ice_therm_vertical.F90|ice_atmo.F90) FX="-DAusCOM" ;;
      ice_dyn_evp.F90) FX="-DAusCOM -Dcoupled" ;;   # `coupled` only where the hot path branches on it (:919-926)
    esac
  fi
  if [ "$DROPIN" = "1" ] && [ "$(basename $s)" = "ice_boundary.F90" ]; then
    $FC $FFLAGS -c "$HERE/../cice4_amd/fortran/cice4_amd_c.F90" -o "$OBJ/cice4_amd_c.o"
    OBJS="$OBJS $OBJ/cice4_amd_c.o"
    src="$HERE/../cice4_amd/fortran/rccl/ice_boundary.F90"
    $FC $FFLAGS -c "$src" -o "$o"
  elif [ "$DROPIN" = "1" ] && [ "$s" = "source/ice_dyn_evp.F90" ]; then
    src="$HERE/../cice4_amd/fortran/ice_dyn_evp.F90"
    $FC $FFLAGS $FX -c "$src" -o "$o"
  elif [ "$DROPIN" = "1" ] && [ "$s" = "source/ice_therm_vertical.F90" ]; then
    src="$HERE/../cice4_amd/fortran/ice_therm_vertical.F90"
    $FC $FFLAGS $FX -c "$src" -o "$o"
  elif [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then
    $FC $FFLAGS $FX -c "$src" -o "$o"
  fi
  OBJS="$OBJS $o"
done
EXTRA=""
if [ "$DROPIN" = "1" ]; then EXTRA="-DDROPIN"; fi
if [ "$AUS" = "1" ]; then EXTRA="$EXTRA -DREF_AUSCOM"; fi
$FC $FFLAGS $EXTRA -c "$HERE/ref_capi.F90" -o "$OBJ/ref_capi.o"
LINK=""
if [ "$DROPIN" = "1" ]; then
  LIBNAME=cice4_amd
  if [ "$AUS" = "1" ]; then LIBNAME=cice4_amd_auscom; fi
  LINK="-L$HERE/../cice4_amd -l$LIBNAME -Wl,-rpath,\$ORIGIN/../../cice4_amd"
fi
if [ "$MPI" = "1" ]; then
  LINK="$LINK -L$MPIROOT/lib -lmpifort -lmpi -Wl,-rpath,$MPIROOT/lib"
fi
$FC -shared -Wl,-Bsymbolic -o "$OUT/libcice_${KIND}_$CFG.so" $OBJS "$OBJ/ref_capi.o" $LINK
echo "built $OUT/libcice_${KIND}_$CFG.so"
