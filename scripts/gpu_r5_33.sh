#!/bin/bash
# Round 5, call 33: issue priority by a wavefront's place on its SIMD (modes 7-10) against the four-step rotation (2 -> 4)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_33.txt
for rep in 1 2; do for m in 2 15 16; do
  timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --resident-prio $m > gpurun_out/r5_33.json 2>gpurun_out/r5_33.err || { tail -20 gpurun_out/r5_33.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r5_33.json')); print('gx1 --resident-prio $m:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" | tee -a gpurun_out/r5_33.txt
done; done
