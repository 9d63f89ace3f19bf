#!/bin/bash
# build variants of the batched thermo kernel (workgroups per CU via OCC_LIST, extra flags via OCC_EXTRA),
# parity tests unless SKIP_TESTS is set, then the gx1 thermo rate of each build
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
FL="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result"
for nb in ${OCC_LIST:-3 4 2}; do
  touch cice4_amd/csrc/therm.hip
  make -s -C cice4_amd/csrc CXXFLAGS="$FL -DCICE_THERMO_MIN_BLOCKS=$nb $OCC_EXTRA" > gpurun_out/r02/occ_build_$nb.log 2>&1
  if [ -z "$SKIP_TESTS" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_thermo.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r02/occ_tests_$nb.log 2>&1 || { tail -20 gpurun_out/r02/occ_tests_$nb.log; exit 1; }
    grep -E "passed|failed" gpurun_out/r02/occ_tests_$nb.log
  fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing --no-tenth > gpurun_out/r02/occ_gx1_$nb.json 2> gpurun_out/r02/occ_gx1_$nb.err
  python - $nb <<'PY'
import json, sys
nb = sys.argv[1]
d = json.loads(open(f"gpurun_out/r02/occ_gx1_{nb}.json").read().strip().splitlines()[-1])
print("min_blocks", nb, "gx1 thermo", "%.3e" % d["thermo"]["value"], "ms/pass %.4f" % d["thermo"]["ms_per_pass"], "evp", "%.0f" % d["value"])
PY
done
