#!/bin/bash
# Round 4, call 21/22: segments by CU fill and by dispatch order, A/B through the bench at 0.1 degree
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_21_ab.txt
for i in 1 2; do
  for d in "0 0" "26 20" "26 15" "26 25" "20 20" "32 20" "40 20"; do
    set -- $d
    CICE4_AMD_SKEW_FILL=$1 timeout -k 10 300 python bench.py --no-thermo --workload tenth --skew-gen-pct $2 > gpurun_out/r4_21.json 2> gpurun_out/r4_21.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_21.json'))
print('fill $1 gen_pct $2:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_21_ab.txt
  done
done
