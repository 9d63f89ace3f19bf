#!/bin/bash
# round 4, call 15: K of the sweep kernel again, with this round's kernel (pairs off for all, so that the levels compare)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_k.txt
for rep in 1 2; do
  for K in 4 3 5 6 2; do
    for wl in tenth 3600x316x240; do
      CICE4_AMD_SKEW_PAIRS=0 timeout -k 10 300 python bench.py --workload $wl --skew-levels $K $B > gpurun_out/k_one.json 2> gpurun_out/k_one.err || { echo "K=$K $wl FAILED" | tee -a gpurun_out/r4_k.txt; tail -3 gpurun_out/k_one.err; continue; }
      echo "rep$rep K=$K $wl $(python -c "import json;d=json.load(open('gpurun_out/k_one.json'));r=d['roofline'];print(round(1e6/d['value'],2), 'us per subcycle; launch', round(r['us_per_launch'],1), d['config']['tile'][:100])")" | tee -a gpurun_out/r4_k.txt
    done
  done
done
