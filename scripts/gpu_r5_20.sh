#!/bin/bash
# Round 5, call 20: which row a wavefront keeps (the chain rows on the oldest wavefront of every SIMD)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CICE4_AMD_RESIDENT_ROWS=1 timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "whole_loop_in_one_launch" > gpurun_out/r5_20_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_20_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_20_tests.log | cut -c1-500 | tail -100; exit 1; }
: > gpurun_out/r5_20.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_20.json 2>gpurun_out/r5_20.err || { tail -20 gpurun_out/r5_20.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_20.json')); print(d['config']['nx_global'], 'x', d['config']['ny_global'], ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_20.txt
}
for rep in 1 2; do
  run "" CICE4_AMD_RESIDENT_ROWS=0
  run "" CICE4_AMD_RESIDENT_ROWS=1
done
run "--resident-prio 0" CICE4_AMD_RESIDENT_ROWS=1
run "--resident-prio 1" CICE4_AMD_RESIDENT_ROWS=1
run "--resident-prio 3" CICE4_AMD_RESIDENT_ROWS=1
run "--resident-waves 12" CICE4_AMD_RESIDENT_ROWS=1
run "" CICE4_AMD_RESIDENT_ROWS=1 CICE4_AMD_RESIDENT_POLL_DELAY=0
run "" CICE4_AMD_RESIDENT_ROWS=1 CICE4_AMD_RESIDENT_POLL_DELAY=4
run "--workload gx3" CICE4_AMD_RESIDENT_ROWS=0
run "--workload gx3" CICE4_AMD_RESIDENT_ROWS=1
