#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- integration harness, not the pinned oracle (that is build_ref.sh).
#
# Build the reference's WHOLE stand-alone model (program icemodel, drivers/cice4/CICE.F90:64-94:
# CICE_Initialize -> CICE_Run (ice_step: prep_radiation, step_therm1, step_therm2, step_dynamics,
# step_radiation, ...) -> CICE_Finalize) from the sources where they lie under /root/reference, twice:
#
#   oracle/_ref/cice_ref_<cfg>      every module the reference's own (serial/ backend)
#   oracle/_ref/cice_dropin_<cfg>   the same, except that source/ice_dyn_evp.F90,
#                                   source/ice_therm_vertical.F90, source/ice_transport_driver.F90 and
#                                   serial/ice_boundary.F90 are replaced by cice4_amd/fortran/{ice_dyn_evp,
#                                   ice_therm_vertical,ice_transport_driver,rccl/ice_boundary}.F90
#                                   (+ cice4_amd_c.F90) and the program is
#                                   linked with libcice4_amd.so -- what a user of the reference who
#                                   swaps the four modules builds.
#
# No reference source is copied into the repository and none is stubbed.  The image has no netCDF;
# all netCDF use in the model is behind `#ifdef ncdf` (left undefined) except three
# `status = nf90_close(fid)` lines in source/ice_forcing.F90:2171,2182,2210 (inside rct_data, never
# executed with atm_data_type='default').  That one file is compiled from a pipe
# (sed ... | amdflang -x f95-cpp-input -) which wraps exactly those three lines in `#ifdef ncdf`; nothing
# is written to disk.  source/dump_field.F90 (unconditional `use netcdf`, not used by this driver) is
# left out.  The reference's build system (comp_ice, bld/) is not run.
#
# MPI=1: the reference's mpi/ backend (MPICH of the image) instead of serial/ -> cice_refmpi_<cfg> / cice_dropinmpi_<cfg>:
# the model as an MPI user builds it; the drop-in boundary module then also creates the device communicator
# (-DCICE4_AMD_MPI).
#
# usage: oracle/build_driver.sh <cfg> <NXGLOB> <NYGLOB> <BLCKX> <BLCKY> <MXBLCKS>     (env: DROPIN=0|1, MPI=0|1)
set -euo pipefail
REF=${CICE_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
CFG=$1; NXG=$2; NYG=$3; BX=$4; BY=$5; MXB=$6
FC=${FC:-/opt/rocm/bin/amdflang}
DROPIN=${DROPIN:-0}
OUT=$HERE/_ref
if [ ! -d "$REF/source" ]; then
  echo "build_driver: $REF not present (GPU box?) -- using prebuilt files in $OUT" >&2
  exit 0
fi
MPI=${MPI:-0}
MPIROOT=${MPIROOT:-/opt/conda}
KIND=ref
[ "$DROPIN" = "1" ] && KIND=dropin
BACKEND=serial
if [ "$MPI" = "1" ]; then
  KIND=${KIND}mpi
  BACKEND=mpi
fi
OBJ=$OUT/drv_${CFG}_$KIND
TARGET=$OUT/cice_${KIND}_$CFG
OURS=$HERE/../cice4_amd/fortran
if [ -f "$TARGET" ] && [ -z "$(find "$HERE/build_driver.sh" "$OURS" -newer "$TARGET" \( -name '*.F90' -o -name '*.sh' \) 2>/dev/null | head -1)" ]; then
  echo "up to date $TARGET"; exit 0
fi
rm -rf "$OBJ"; mkdir -p "$OBJ"
FFLAGS="-O2 -w -cpp -fdefault-real-8 -fconvert=big-endian -ffp-contract=off \
 -DLINUX -DNXGLOB=$NXG -DNYGLOB=$NYG -DBLCKX=$BX -DBLCKY=$BY -DMXBLCKS=$MXB -J $OBJ -I $OBJ"
[ "$MPI" = "1" ] && FFLAGS="$FFLAGS -I$MPIROOT/include -DCICE4_AMD_MPI"

# the source list: the reference's files, with ours substituted for the drop-in build
LIST=$(ls $REF/drivers/cice4/*.F90 $REF/source/*.F90 $REF/$BACKEND/*.F90 $REF/csm_share/*.F90 | grep -v dump_field.F90)
if [ "$DROPIN" = "1" ]; then
  LIST=$(echo "$LIST" | grep -v -e source/ice_dyn_evp.F90 -e source/ice_therm_vertical.F90 -e $BACKEND/ice_boundary.F90 \
         -e source/ice_transport_driver.F90)
  LIST="$LIST $OURS/cice4_amd_c.F90 $OURS/ice_dyn_evp.F90 $OURS/ice_therm_vertical.F90 $OURS/rccl/ice_boundary.F90 \
        $OURS/ice_transport_driver.F90"
fi
# module/use topological order
ORDER=$(python3 - $LIST <<'EOF'
import re, sys
files = sys.argv[1:]
prov, uses = {}, {}
for f in files:
    txt = open(f, errors="replace").read().lower()
    for m in re.finditer(r"^\s*module\s+(\w+)\s*$", txt, re.M):
        if m.group(1) != "procedure":
            prov[m.group(1)] = f
    uses[f] = set(re.findall(r"^\s*use\s+(\w+)", txt, re.M))
done, out = set(), []
def visit(f, stack=()):
    if f in done:
        return
    assert f not in stack, ("cycle", f)
    for u in sorted(uses[f]):
        g = prov.get(u)
        if g and g != f:
            visit(g, stack + (f,))
    done.add(f); out.append(f)
for f in sorted(files):
    visit(f)
print("\n".join(out))
EOF
)
OBJS=""
for src in $ORDER; do
  o=$OBJ/$(basename "${src%.F90}").o
  if [ "$src" = "$REF/source/ice_forcing.F90" ]; then
    sed -e '2171s/^\(.*nf90_close.*\)$/#ifdef ncdf\n\1\n#endif/' \
        -e '2182s/^\(.*nf90_close.*\)$/#ifdef ncdf\n\1\n#endif/' \
        -e '2210s/^\(.*nf90_close.*\)$/#ifdef ncdf\n\1\n#endif/' "$src" \
      | $FC $FFLAGS -x f95-cpp-input -c - -o "$o"
  else
    $FC $FFLAGS -c "$src" -o "$o"
  fi
  OBJS="$OBJS $o"
done
LINK=""
if [ "$DROPIN" = "1" ]; then
  LINK="-L$HERE/../cice4_amd -lcice4_amd -Wl,-rpath,\$ORIGIN/../../cice4_amd"
fi
[ "$MPI" = "1" ] && LINK="$LINK -L$MPIROOT/lib -lmpifort -lmpi -Wl,-rpath,$MPIROOT/lib"
$FC -o "$TARGET" $OBJS $LINK
echo "built $TARGET"
