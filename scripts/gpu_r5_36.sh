#!/bin/bash
# Round 5, call 36: the sweep kernel with the loop over pieces (nothing uses it yet): same rate?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "k_subcycles_per_sweep or sweep_segments" > gpurun_out/r5_36_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_36_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_36_tests.log | cut -c1-400 | tail -60; exit 1; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --workload tenth --no-thermo --no-cpu-baseline --no-dropin-timing ${EXTRA} > gpurun_out/r5_36.json 2>gpurun_out/r5_36.err || { tail -20 gpurun_out/r5_36.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r5_36.json')); print('tenth:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; kernel', round(d['roofline']['us_per_launch'],1), 'us per launch')"
done
