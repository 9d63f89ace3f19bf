#!/bin/bash
# Round 5, call 39: places by strip once (1) / and a second time from measured strip times (2) / not at all (0): same box
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_39.txt
for rep in 1 2 3; do for pl in 0 1 2; do
CICE4_AMD_SKEW_PLACES=$pl timeout -k 10 300 python bench.py --workload tenth --no-thermo --no-cpu-baseline --no-dropin-timing > gpurun_out/r5_39.json 2>gpurun_out/r5_39.err || { tail -20 gpurun_out/r5_39.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r5_39.json')); print('tenth full cover places=$pl:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; kernel', round(d['roofline']['us_per_launch'],1), 'us per launch')" | tee -a gpurun_out/r5_39.txt
done; done
