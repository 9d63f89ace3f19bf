#!/bin/bash
# Round 4, call 40: the one-launch loop with tile = blockIdx (a CU holds three tiles a third of the grid apart): parity, gx1 by cover
set -o pipefail
mkdir -p gpurun_out
CICE4_AMD_RESIDENT_MAP=1 timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "one_launch or resident or whole_loop" > gpurun_out/r4_40_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_40_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_40_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_40.txt
for i in 1 2; do
for c in full caps patchy; do
  for m in 0 1; do
    CICE4_AMD_RESIDENT_MAP=$m timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --cover $c > gpurun_out/r4_40.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_40.json')); print('gx1 cover $c map $m:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle')" | tee -a gpurun_out/r4_40.txt
  done
done
done
