#!/bin/bash
# The library's HOST code under AddressSanitizer (CPU only; GPU ASAN is not available on the pool): capi / evp / halo / domain
# compiled -fsanitize=address into build/asan/, the no-device C-ABI tests run against it (domain and message lists, the sweep's
# strip layout, the balancer's step, argument checks).  usage: scripts/asan_host_tests.sh
set -e
cd "$(dirname "$0")/../cice4_amd/csrc"
make -s
mkdir -p ../../build/asan
for f in capi.hip evp.hip halo.hip domain.cpp; do
  x=""; [ $f = domain.cpp ] && x="-x hip"
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fsanitize=address -fno-omit-frame-pointer $x -c $f -o ../../build/asan/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -shared-libsan ../../build/asan/capi.hip.o ../../build/asan/evp.hip.o \
  ../../build/obj/therm.hip.o ../../build/obj/atmo.hip.o ../../build/obj/transport.hip.o ../../build/asan/halo.hip.o ../../build/asan/domain.cpp.o \
  -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o ../../build/asan/libcice4_amd_asan.so
cd ../..
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cat > build/asan/run.py <<'PY'
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from cice4_amd import lib
lib.LIBPATH = os.path.join(root, "build", "asan", "libcice4_amd_asan.so")
import pytest
raise SystemExit(pytest.main(["-q", "-m", "not gpu", "tests/test_capi.py", "tests/test_domain.py", "-k", "not auscom", "-p", "no:cacheprovider"]))
PY
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python build/asan/run.py
