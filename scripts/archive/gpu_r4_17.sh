#!/bin/bash
# Round 4, call 17: the T-cell inputs along the hand-off (SKEW_TPASS) on top of the interleaved inputs: parity, then A/B at 0.1 degree
set -o pipefail
mkdir -p gpurun_out
L=build/ab/lib_tp.so
timeout -k 10 600 python scripts/test_with_lib.py $L tests/test_gpu_evp.py -x -q -m gpu -k "sweep or slabs or tripole" > gpurun_out/r4_17_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_17_tests.log | tail -2
[ $rc -eq 0 ] || { tail -30 gpurun_out/r4_17_tests.log; exit 1; }
: > gpurun_out/r4_17_ab.txt
for i in 1 2 3; do
  for v in default tp; do
    if [ $v = default ]; then X="bench.py"; else X="scripts/bench_with_lib.py $L"; fi
    timeout -k 10 300 python $X --no-thermo --workload tenth > gpurun_out/r4_17_$v.json 2> gpurun_out/r4_17_$v.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_17_$v.json'))
print('$v', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_17_ab.txt
  done
done
