#!/bin/bash
# Round 5, call 3: the free-running granule loop (no workgroup barrier): parity, phase clock, rates by shape / priority / poll delay
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "whole_loop_in_one_launch or resident" > gpurun_out/r5_03_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_03_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_03_tests.log | cut -c1-400 | tail -90; exit 1; }
timeout -k 10 300 python scripts/resident_phases.py build/ab/lib_stamps.so gpurun_out/r5_03_phases.csv > gpurun_out/r5_03_phases.txt 2>&1 || { tail -20 gpurun_out/r5_03_phases.txt; exit 1; }
cat gpurun_out/r5_03_phases.txt
: > gpurun_out/r5_03.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_03.json 2>gpurun_out/r5_03.err || { tail -20 gpurun_out/r5_03.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_03.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_03.txt
}
run "" CICE4_AMD_RESIDENT_GRANULES=0
for w in 0 11 12; do
  for p in 0 1 2; do
    run "--resident-waves $w --resident-prio $p" CICE4_AMD_RESIDENT_GRANULES=1
  done
done
for d in 1 2 4; do
  run "--resident-waves 0 --resident-prio 1" CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=$d
  run "--resident-waves 12 --resident-prio 1" CICE4_AMD_RESIDENT_GRANULES=1 CICE4_AMD_RESIDENT_POLL_DELAY=$d
done
run "" CICE4_AMD_RESIDENT_GRANULES=0
