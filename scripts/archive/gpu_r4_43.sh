#!/bin/bash
# Round 4, call 43: workgroups of the per-subcycle kernels leave at once where their tile holds no ice: parity (whole GPU suite), rates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_43_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_43_tests.log | tail -2 | cut -c1-200
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_43_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_43.txt
for c in full caps; do
  for f in "" "--no-fuse"; do
    timeout -k 10 200 python bench.py --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing --no-resident $f --cover $c > gpurun_out/r4_43.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_43.json')); print('gx1 cover $c, no one-launch loop $f:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle')" | tee -a gpurun_out/r4_43.txt
  done
  timeout -k 10 300 python bench.py --no-thermo --workload tenth --no-skew --cover $c > gpurun_out/r4_43.json 2>/dev/null || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/r4_43.json')); print('0.1 degree cover $c, pairs of subcycles (no sweeps):', round(1e6/d['value'],1), 'us per subcycle')" | tee -a gpurun_out/r4_43.txt
done
