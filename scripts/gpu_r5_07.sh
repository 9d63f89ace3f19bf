#!/bin/bash
# Round 5, call 7: priority rules of the free-running loop
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_07.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_07.json 2>gpurun_out/r5_07.err || { tail -20 gpurun_out/r5_07.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_07.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_07.txt
}
run "" CICE4_AMD_RESIDENT_GRANULES=0
for w in 11 12 8 0; do
  for p in 2 3 4; do
    run "--resident-waves $w --resident-prio $p" CICE4_AMD_RESIDENT_GRANULES=1
  done
done
for d in 1 2 3; do
  run "--resident-waves 11 --resident-prio 2" CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=$d
  run "--resident-waves 11 --resident-prio 3" CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=$d
done
