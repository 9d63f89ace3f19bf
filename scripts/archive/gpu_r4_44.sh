#!/bin/bash
# Round 4, call 44: what a rank's slab of the 0.1-degree grid costs with the kernel as it stands at the end of the round
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing --no-thermo --no-tenth --steps 4 --warmup 1"
: > gpurun_out/r4_44_slabs.txt
for wl in 3600x2400x240 3600x316x240 3600x616x240 3600x1216x240; do
  timeout -k 10 200 python bench.py --workload $wl $B > gpurun_out/r4_44.json 2> gpurun_out/r4_44.err || { tail -3 gpurun_out/r4_44.err; exit 1; }
  python - $wl <<'PY' | tee -a gpurun_out/r4_44_slabs.txt
import json, sys
d = json.load(open("gpurun_out/r4_44.json"))
print(sys.argv[1], "| us per subcycle", round(1e6 / d["value"], 2), flush=True)
PY
done
