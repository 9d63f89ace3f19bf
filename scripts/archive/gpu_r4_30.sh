#!/bin/bash
# Round 4, call 30: HTE west of ilo by a wave shift instead of a load (the seam strips): parity, trace, bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu > gpurun_out/r4_30_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_30_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_30_tests.log | cut -c1-300 | tail -60; exit 1; }
timeout -k 10 300 python scripts/sweep_balance_trace.py full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_30_trace_full.txt | grep -v "^   strip " | cut -c1-400 | tail -4
: > gpurun_out/r4_30_ab.txt
for i in 1 2; do
  for v in "0 0" "1 1"; do
    set -- $v
    CICE4_AMD_SKEW_ROWACT=$1 timeout -k 10 300 python bench.py --no-thermo --workload tenth --skew-balance $2 > gpurun_out/r4_30.json 2> gpurun_out/r4_30.err || { tail -5 gpurun_out/r4_30.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r4_30.json'))
print('cover full, rows-with-ice $1 balance $2:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_30_ab.txt
  done
done
