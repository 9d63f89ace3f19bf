#!/bin/bash
# Round 4, call 26: full cover at 0.1 degree: rows-with-ice / balance off and on, alternating
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_26_ab.txt
for i in 1 2 3; do
  for v in "0 0" "1 0" "1 1" "0 1"; do
    set -- $v
    CICE4_AMD_SKEW_ROWACT=$1 timeout -k 10 300 python bench.py --no-thermo --workload tenth --skew-balance $2 > gpurun_out/r4_26.json 2> gpurun_out/r4_26.err || { tail -5 gpurun_out/r4_26.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r4_26.json'))
print('full cover, rows-with-ice $1 balance $2:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_26_ab.txt
  done
done
