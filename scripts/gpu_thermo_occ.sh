#!/bin/bash
# occupancy sweep of the batched thermo kernel: rebuild therm.hip with 3 and 4 workgroups per CU
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for nb in 3 4 2; do
  touch cice4_amd/csrc/therm.hip
  make -s -C cice4_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result -DCICE_THERMO_MIN_BLOCKS=$nb" > gpurun_out/occ_build_$nb.log 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing > gpurun_out/occ_gx1_$nb.json 2> gpurun_out/occ_gx1_$nb.err
  timeout -k 10 300 python bench.py --workload tenth --steps 1 --warmup 1 --no-cpu-baseline --no-dropin-timing > gpurun_out/occ_tenth_$nb.json 2> gpurun_out/occ_tenth_$nb.err
  python - $nb <<'PY'
import json, sys
nb = sys.argv[1]
for w in ("gx1", "tenth"):
    d = json.loads(open(f"gpurun_out/occ_{w}_{nb}.json").read().strip().splitlines()[-1])
    print("min_blocks", nb, w, "thermo", d["thermo"]["value"])
PY
done
