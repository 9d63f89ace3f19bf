#!/bin/bash
# Round 5, call 38: places by strip (CICE4_AMD_SKEW_PLACES, default on) with other static weights for the dispatch order
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_38.txt
for rep in 1 2 3; do for pct in 15 12 10 8; do
timeout -k 10 300 python bench.py --workload tenth --no-thermo --no-cpu-baseline --no-dropin-timing --skew-gen-pct $pct > gpurun_out/r5_38.json 2>gpurun_out/r5_38.err || { tail -20 gpurun_out/r5_38.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r5_38.json')); print('tenth full cover, places by strip, skew_gen_pct $pct:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; kernel', round(d['roofline']['us_per_launch'],1), 'us per launch')" | tee -a gpurun_out/r5_38.txt
done; done
