#!/bin/bash
# Round 4, call 23: sweep segments by measured cost -- parity, then the bench at 0.1 degree: full cover and polar caps, balance off / on
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "sweep or tripole" > gpurun_out/r4_23_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_23_tests.log | tail -2
[ $rc -eq 0 ] || { tail -40 gpurun_out/r4_23_tests.log | cut -c1-300; exit 1; }
: > gpurun_out/r4_23_ab.txt
for i in 1 2; do
  for c in full caps; do
    for b in 0 1; do
      timeout -k 10 300 python bench.py --no-thermo --workload tenth --cover $c --skew-balance $b > gpurun_out/r4_23.json 2> gpurun_out/r4_23.err || { tail -5 gpurun_out/r4_23.err; exit 1; }
      python -c "
import json
d=json.load(open('gpurun_out/r4_23.json'))
print('cover $c balance $b:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_23_ab.txt
    done
  done
done
