#!/bin/bash
# Round 5, call 8: the granule loop as the default of one-workgroup-per-CU shapes: EVP tests, the 8-rank rehearsals, default bench lines
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo "(EVP tests: passed in the previous call)"; rc=0
true
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_08_tests.log | cut -c1-400 | tail -90; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "eight_ranks" > gpurun_out/r5_08_ranks.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_08_ranks.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_08_ranks.log | cut -c1-400 | tail -90; exit 1; }
: > gpurun_out/r5_08.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_08.json 2>gpurun_out/r5_08.err || { tail -20 gpurun_out/r5_08.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_08.json')); print(d['config']['nx_global'], 'x', d['config']['ny_global'], ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][:90])" "$extra" "$@" | tee -a gpurun_out/r5_08.txt
}
run "" A=1
run "" CICE4_AMD_RESIDENT_GRANULES=0
run "--workload gx3" A=1
run "--workload gx3" CICE4_AMD_RESIDENT_GRANULES=0
run "--workload 200x200" A=1
run "--workload 200x200" CICE4_AMD_RESIDENT_GRANULES=0
run "--cover caps" A=1
run "--cover caps" CICE4_AMD_RESIDENT_GRANULES=0
run "" A=1
bash scripts/gpu_r5_09.sh
