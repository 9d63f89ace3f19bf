#!/bin/bash
# Round 4, call 52: under a fold (dense shape) the tiles of the top row always issue first on their CU: parity, gx1 tripole A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "tripole or fold" > gpurun_out/r4_52_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_52_tests.log | tail -2 | cut -c1-200
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_52_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_52.txt
for i in 1 2; do
  for t in 0 1; do
    CICE4_AMD_RESIDENT_PRIO_TOP=$t timeout -k 10 200 python bench.py --north tripole --no-thermo --no-tenth > gpurun_out/r4_52.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_52.json')); print('gx1 tripole, top-row tiles first $t:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle')" | tee -a gpurun_out/r4_52.txt
  done
done
