#!/bin/bash
# Round 5, call 2: where a subcycle of the granule loop goes (phase clock, diagnostic build), and the poll's delay / sleep knobs
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python scripts/resident_phases.py build/ab/lib_stamps.so gpurun_out/r5_02_phases.csv > gpurun_out/r5_02_phases.txt 2>&1 || { tail -20 gpurun_out/r5_02_phases.txt; exit 1; }
cat gpurun_out/r5_02_phases.txt
: > gpurun_out/r5_02.txt
run() {
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing > gpurun_out/r5_02.json 2>gpurun_out/r5_02.err || { tail -20 gpurun_out/r5_02.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_02.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$@" | tee -a gpurun_out/r5_02.txt
}
run CICE4_AMD_RESIDENT_GRANULES=0
for d in 0 1 2 3 4 6; do run CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=$d; done
for s in 1 2 4; do run CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=2 CICE4_AMD_RESIDENT_POLL_SLEEP=$s; done
run CICE4_AMD_RESIDENT_GRANULES=0
