#!/bin/bash
# Round 4, call 18: which time levels share a SIMD, and the deal that mixes them (CICE4_AMD_SKEW_DEAL)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_18_placement.txt
for d in 0 1; do
  CICE4_AMD_SKEW_DEAL=$d timeout -k 10 200 python scripts/sweep_placement.py build/ab/lib_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_18_placement.txt || exit 1
done
CICE4_AMD_SKEW_DEAL=1 timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "sweep" > gpurun_out/r4_18_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_18_tests.log | tail -2
[ $rc -eq 0 ] || { tail -30 gpurun_out/r4_18_tests.log; exit 1; }
: > gpurun_out/r4_18_ab.txt
for i in 1 2; do
  for d in 0 1 3 2; do
    CICE4_AMD_SKEW_DEAL=$d timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_18_$d.json 2> gpurun_out/r4_18_$d.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_18_$d.json'))
print('deal $d', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_18_ab.txt
  done
done
