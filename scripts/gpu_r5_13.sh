#!/bin/bash
# Round 5, call 13: the cross-rank one-launch loop on cartesian layouts (E-W and diagonal neighbours)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "cartesian" > gpurun_out/r5_13_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_13_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_13_tests.log | cut -c1-500 | tail -100; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cartesian" > gpurun_out/r5_13_full.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_13_full.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_13_full.log | cut -c1-500 | tail -100; exit 1; }
for c in caps full; do
  timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --cover $c > gpurun_out/r5_13.json 2>gpurun_out/r5_13.err || { tail -20 gpurun_out/r5_13.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r5_13.json')); print('gx1 cover $c:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][60:160])" | tee -a gpurun_out/r5_13.txt
done
