#!/bin/bash
# Round 5, call 11: shape by ice cover; the whole GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_11_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_11_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_11_tests.log | cut -c1-400 | tail -90; exit 1; }
for c in full caps patchy; do
  timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --cover $c > gpurun_out/r5_11.json 2>gpurun_out/r5_11.err || { tail -20 gpurun_out/r5_11.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r5_11.json')); print('gx1 cover $c:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][:100])" | tee -a gpurun_out/r5_11.txt
done
