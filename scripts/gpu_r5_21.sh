#!/bin/bash
# Round 5, call 21: how often a waiting wavefront looks (poll delay / sleep between passes / sleep between looks at an LDS flag)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_21.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_21.json 2>gpurun_out/r5_21.err || { tail -20 gpurun_out/r5_21.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_21.json')); print(d['config']['nx_global'], 'x', d['config']['ny_global'], ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_21.txt
}
run "" A=1
for ls in 1 2 4 8; do run "" CICE4_AMD_RESIDENT_LDS_SLEEP=$ls; done
for ps in 1 2; do for pd in 1 2 3; do run "" CICE4_AMD_RESIDENT_POLL_DELAY=$pd CICE4_AMD_RESIDENT_POLL_SLEEP=$ps; done; done
run "" CICE4_AMD_RESIDENT_POLL_DELAY=3
run "" A=1
