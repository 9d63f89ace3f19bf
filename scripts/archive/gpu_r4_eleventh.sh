#!/bin/bash
# round 4, eleventh call: the sweep's pair layout of the state (16-byte loads / stores of u | v and the stresses) -- parity, rates with and without
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_evp.py tests/test_gpu_fullsize.py tests/test_gpu_auscom.py -x -q -k "k_subcycles_per_sweep or sweeps_on_a_tripole or tenth or sweep or auscom" > gpurun_out/r4_tests11.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4_tests11.log | tail -3 | cut -c1-300 | tee gpurun_out/r4_tests11.txt
grep -q "passed" gpurun_out/r4_tests11.txt && ! grep -q "failed" gpurun_out/r4_tests11.txt || { grep -B40 "short test summary" gpurun_out/r4_tests11.log | tail -60 | cut -c1-250; exit 1; }
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_ab11.txt
for rep in 1 2 3; do
  for pairs in 0 1; do
    for wl in tenth 1440x1080x240 3600x600x240; do
      CICE4_AMD_SKEW_PAIRS=$pairs timeout -k 10 300 python bench.py --workload $wl $B > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "pairs=$pairs $wl FAILED" | tee -a gpurun_out/r4_ab11.txt; tail -3 gpurun_out/ab_one.err; continue; }
      echo "rep$rep pairs=$pairs $wl $(python -c "import json;d=json.load(open('gpurun_out/ab_one.json'));r=d['roofline'];print(round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; launch', round(r['us_per_launch'],1))")" | tee -a gpurun_out/r4_ab11.txt
    done
  done
done
