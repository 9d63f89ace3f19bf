#!/bin/bash
# Round 5, call 1: granule hand-off in the one-launch loop: parity of the loop tests, then gx1 A/B (granules against progress words), interleaved
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "whole_loop_in_one_launch or resident or smallest" > gpurun_out/r5_01_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_01_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_01_tests.log | cut -c1-400 | tail -90; exit 1; }
: > gpurun_out/r5_01.txt
for rep in 1 2 3; do
  for g in 1 0; do
    CICE4_AMD_RESIDENT_GRANULES=$g timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing > gpurun_out/r5_01.json 2>gpurun_out/r5_01.err || { tail -20 gpurun_out/r5_01.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r5_01.json')); print('gx1 granules=$g:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" | tee -a gpurun_out/r5_01.txt
  done
done
