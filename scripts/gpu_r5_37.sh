#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py tests/test_gpu_fullsize.py -x -q -m gpu -k "k_subcycles_per_sweep or sweep_segments or tenth_degree_24 or tenth_degree_whole or sweeps_on_a_tripole" > gpurun_out/r5_37_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_37_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_37_tests.log | cut -c1-500 | tail -70; exit 1; }
: > gpurun_out/r5_37.txt
for cover in full caps; do for st in 0 1 0 1; do
CICE4_AMD_SKEW_PLACES=$st timeout -k 10 300 python bench.py --workload tenth --cover $cover --no-thermo --no-cpu-baseline --no-dropin-timing > gpurun_out/r5_37.json 2>gpurun_out/r5_37.err || { tail -20 gpurun_out/r5_37.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r5_37.json')); print('tenth cover $cover places=$st:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; kernel', round(d['roofline']['us_per_launch'],1), 'us per launch')" | tee -a gpurun_out/r5_37.txt
done; done
