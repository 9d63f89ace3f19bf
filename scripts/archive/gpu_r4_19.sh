#!/bin/bash
# Round 4, call 19: longer segments for the workgroups on CUs that hold two instead of three (CICE4_AMD_SKEW_FILL): parity, A/B
set -o pipefail
mkdir -p gpurun_out
CICE4_AMD_SKEW_FILL=1 timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "sweep" > gpurun_out/r4_19_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_19_tests.log | tail -2
[ $rc -eq 0 ] || { tail -30 gpurun_out/r4_19_tests.log; exit 1; }
: > gpurun_out/r4_19_ab.txt
for i in 1 2 3; do
  for d in 0 1; do
    CICE4_AMD_SKEW_FILL=$d timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_19_$d.json 2> gpurun_out/r4_19_$d.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_19_$d.json'))
print('fill $d', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_19_ab.txt
  done
done
