#!/bin/bash
# Round 4, call 41: the one-launch loop chooses its tile map by the ice cover (k_res_choose_map): parity, gx1 by cover
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu > gpurun_out/r4_41_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_41_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_41_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_41.txt
for c in full caps patchy; do
    timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --cover $c > gpurun_out/r4_41.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_41.json')); print('gx1 cover $c, map chosen by the library:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle')" | tee -a gpurun_out/r4_41.txt
done
