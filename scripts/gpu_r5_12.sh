#!/bin/bash
# Round 5, call 12: shape by cover re-checked on one box; per-rank costs of the 8-rank decompositions through the slab code path
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_12.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_12.json 2>gpurun_out/r5_12.err || { tail -20 gpurun_out/r5_12.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_12.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][60:130])" "$extra" "$@" | tee -a gpurun_out/r5_12.txt
}
run "--cover caps" A=1
run "--cover caps" CICE4_AMD_RESIDENT_GRANULES=0
run "--cover caps" CICE4_AMD_RESIDENT_GRANULES=1
run "" A=1
run "" CICE4_AMD_RESIDENT_GRANULES=0
timeout -k 10 900 python scripts/rank_costs.py --workload tenth --ranks 8 --out gpurun_out/r5_12_rank_costs_tenth8.jsonl 2>gpurun_out/r5_12_rc.err | tee -a gpurun_out/r5_12.txt || { tail -20 gpurun_out/r5_12_rc.err; exit 1; }
