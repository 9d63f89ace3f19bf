#!/bin/bash
# Round 4, call 24: workgroups of the sweep shrink to the rows that hold ice + segments by measured cost: parity (whole EVP file), bench
set -o pipefail
mkdir -p gpurun_out
true


: > gpurun_out/r4_24_ab.txt
for c in caps full patchy; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    CICE4_AMD_SKEW_ROWACT=$1 timeout -k 10 300 python bench.py --no-thermo --workload tenth --cover $c --skew-balance $2 > gpurun_out/r4_24.json 2> gpurun_out/r4_24.err || { tail -5 gpurun_out/r4_24.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r4_24.json'))
print('cover $c rows-with-ice $1 balance $2:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_24_ab.txt
  done
done
