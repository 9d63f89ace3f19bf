#!/bin/bash
# build/ab/lib_<name>.so = the library with therm.hip compiled with extra flags
set -e
cd "$(dirname "$0")/../cice4_amd/csrc"
name=$1; shift
mkdir -p ../../build/ab
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 "$@" -c therm.hip -o ../../build/ab/therm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 ../../build/obj/capi.hip.o ../../build/obj/evp.hip.o ../../build/ab/therm_$name.o ../../build/obj/atmo.hip.o ../../build/obj/transport.hip.o ../../build/obj/halo.hip.o ../../build/obj/domain.cpp.o -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../../build/ab/lib_$name.so
