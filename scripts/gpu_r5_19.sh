#!/bin/bash
# Round 5, call 19: one pinned read-back per loop: step time against kernel time; loop tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "whole_loop_in_one_launch or resident or shape or tile_map" > gpurun_out/r5_19_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_19_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_19_tests.log | cut -c1-500 | tail -100; exit 1; }
for c in full caps; do
  timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --cover $c > gpurun_out/r5_19.json 2>gpurun_out/r5_19.err || { tail -20 gpurun_out/r5_19.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r5_19.json')); print('gx1 cover $c:', round(d['value']), 'subcycles/s; step', round(d['ms_per_step']*1e3,1), 'us, kernel by HIP events', round(d['roofline']['us_per_launch'],1), 'us;', d['config']['tile'][60:150])" | tee -a gpurun_out/r5_19.txt
done
