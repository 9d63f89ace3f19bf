#!/bin/bash
# Round 5, call 10: issue priority by what a wavefront is doing (0 while it waits)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_10.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_10.json 2>gpurun_out/r5_10.err || { tail -20 gpurun_out/r5_10.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_10.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_10.txt
}
for p in 4 5 6; do
  for d in 0 2; do
    run "--resident-prio $p" CICE4_AMD_RESIDENT_POLL_DELAY=$d
  done
  run "--resident-prio $p --resident-waves 12" A=1
  run "--resident-prio $p" CICE4_AMD_RESIDENT_FAKE_EW=1
done
