#!/bin/bash
# Round 4, call 48: the one-launch loop under a tripole fold in the dense shape (three 4-wavefront workgroups per CU): parity, gx1 rates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py tests/test_gpu_step.py -x -q -m gpu -k "tripole or fold" > gpurun_out/r4_48_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_48_tests.log | tail -2 | cut -c1-200
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_48_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_48.txt
for i in 1 2; do
  for n in tripole tripoleT; do
    for d in 0 1; do
      CICE4_AMD_RESIDENT_FOLD_DENSE=$d timeout -k 10 200 python bench.py --north $n --no-thermo --no-tenth > gpurun_out/r4_48.json 2>/dev/null || exit 1
      python -c "
import json
d=json.load(open('gpurun_out/r4_48.json')); print('gx1 $n, dense shape under the fold $d:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle |', d['config']['tile'][:70])" | tee -a gpurun_out/r4_48.txt
    done
  done
done
