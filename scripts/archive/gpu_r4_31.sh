#!/bin/bash
# Round 4, call 31: the tiles of the strips that hold 11 tiles instead of 12 issue first in 0 / 1 / 2 / 3 of three steps
set -o pipefail
mkdir -p gpurun_out
CICE4_AMD_SKEW_BOOST=2 timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "sweep" > gpurun_out/r4_31_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_31_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_31_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_31_ab.txt
for i in 1 2; do
  for v in 0 1 2 3; do
    CICE4_AMD_SKEW_BOOST=$v timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_31.json 2> gpurun_out/r4_31.err || { tail -5 gpurun_out/r4_31.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r4_31.json'))
print('cover full, boost $v:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_31_ab.txt
  done
done
CICE4_AMD_SKEW_BOOST=2 timeout -k 10 300 python scripts/sweep_balance_trace.py full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_31_trace_full.txt | grep -v "^   strip " | cut -c1-400 | tail -2
