#!/bin/bash
# Round 5, call 25: the granule loop under the fold after the store-hazard fix; gx1 open / tripole rates
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "one_launch or whole_loop or fold or shape or tile" > gpurun_out/r5_25_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_25_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_25_tests.log | cut -c1-400 | tail -90; exit 1; }
: > gpurun_out/r5_25.txt
for north in open tripole; do for g in 1 0; do
  CICE4_AMD_RESIDENT_GRANULES=$g timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --north $north > gpurun_out/r5_25.json 2>gpurun_out/r5_25.err || { tail -20 gpurun_out/r5_25.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r5_25.json')); print('gx1 $north granules=$g:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][:110])" | tee -a gpurun_out/r5_25.txt
done; done
