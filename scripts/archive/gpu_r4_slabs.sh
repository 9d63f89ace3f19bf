#!/bin/bash
# round 4: what one rank's slab of the 0.1-degree grid costs with sweeps (k_subcycle_skew), by overlap H, K and rows per workgroup
# usage: gpu_r4_slabs.sh [lib.so]   (default: the product library)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
LIB=${1:-cice4_amd/libcice4_amd.so}
B="--no-cpu-baseline --no-dropin-timing --no-thermo --no-tenth --steps 4 --warmup 1"
out=gpurun_out/r4_slabs.jsonl
: > $out
run() {   # label, workload, extra args
  local label=$1 wl=$2; shift 2
  timeout -k 10 200 python scripts/bench_with_lib.py $LIB --workload $wl $B "$@" > gpurun_out/r4_slab_one.json 2>> gpurun_out/r4_slabs.err || { echo "$label FAILED"; tail -3 gpurun_out/r4_slabs.err; return 0; }
  python - "$label" <<'PY' | tee -a gpurun_out/r4_slabs.txt
import json, sys
d = json.load(open("gpurun_out/r4_slab_one.json"))
r = d["roofline"]
print(sys.argv[1], d["config"]["nx_global"], d["config"]["ny_global"], "| us/subcycle", round(r["us_per_launch"] / r["subcycles_per_launch"], 2),
      "| us/launch", round(r["us_per_launch"], 1), "|", d["config"]["tile"][:120], flush=True)
PY
  cat gpurun_out/r4_slab_one.json >> $out
}
: > gpurun_out/r4_slabs.txt
run full 3600x2400x240
for rows in 300 600 1200; do
  for H in 4 8 12; do
    run "rows${rows}_H${H}_auto" 3600x$((rows + 2 * H))x240
  done
done
# the 8-rank slab, H = 8: rows per workgroup and K
for seg in 24 32 40 46 53 64 79 106 158; do
  run "rows300_H8_K4_seg$seg" 3600x316x240 --skew-levels 4 --skew-seg-rows $seg
done
for seg in 0 16 21 27 32 40 53 79; do
  run "rows300_H8_K3_seg$seg" 3600x316x240 --skew-levels 3 --skew-seg-rows $seg
done
run "rows300_H8_pairs" 3600x316x240 --no-skew
run "rows300_H8_K2" 3600x316x240 --skew-levels 2
