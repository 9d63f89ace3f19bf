#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the checker, never the product.
#
# Compile the reference's own Fortran hot-path modules (EVP dynamics + vertical
# thermodynamics and the 27 modules they `use`) WHERE THEY LIE under
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
 source/ice_restart.F90 source/ice_age.F90 source/ice_therm_vertical.F90"
OBJS=""
for s in $SRCS; do
  o=$OBJ/$(basename "${s%.F90}").o
  if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then
    $FC $FFLAGS -c "$REF/$s" -o "$o"
  fi
  OBJS="$OBJS $o"
done
$FC $FFLAGS -c "$HERE/ref_capi.F90" -o "$OBJ/ref_capi.o"
$FC -shared -Wl,-Bsymbolic -o "$OUT/libcice_ref_$CFG.so" $OBJS "$OBJ/ref_capi.o"
echo "built $OUT/libcice_ref_$CFG.so"
