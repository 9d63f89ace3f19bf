#!/bin/bash
# Round 4, call 16: sweeps on a tripole grid with the sweep BESIDE the band (second stream) -- parity, then A/B at 0.1 degree.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "tripole" > gpurun_out/r4_16_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_16_tests.log | tail -2
[ $rc -eq 0 ] || { tail -30 gpurun_out/r4_16_tests.log; exit 1; }
for b in 1 0; do
  for n in tripole tripoleT; do
    CICE4_AMD_SKEW_FOLD_BESIDE=$b timeout -k 10 300 python bench.py --north $n --no-thermo --workload tenth > gpurun_out/r4_16_${n}_$b.json 2> gpurun_out/r4_16_${n}_$b.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_16_${n}_$b.json'))
print('beside=$b', '$n', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_16_ab.txt
  done
done
