#!/bin/bash
# round 4, call 13: twelve-wavefront workgroups of the sweep kernel (three wavefronts per level) -- parity on all strip widths, slabs, 0.1 degree; rates against one wavefront per level
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_evp.py tests/test_gpu_fullsize.py tests/test_gpu_auscom.py tests/test_gpu_multiproc.py -x -q -k "k_subcycles_per_sweep or wide_halo or sweeps_on_a_tripole or tenth or ranks_in_one_process or sweep or slabs or auscom or rank_processes" > gpurun_out/r4_tests13.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4_tests13.log | tail -3 | cut -c1-300 | tee gpurun_out/r4_tests13.txt
grep -q "passed" gpurun_out/r4_tests13.txt && ! grep -q "failed" gpurun_out/r4_tests13.txt || { grep -B45 "short test summary" gpurun_out/r4_tests13.log | tail -70 | cut -c1-250; exit 1; }
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_ab13.txt
for rep in 1 2 3; do
  for subs in 1 3; do
    for wl in tenth 1440x1080x240 3600x316x240; do
      CICE4_AMD_SKEW_SUBS=$subs timeout -k 10 300 python bench.py --workload $wl $B > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "subs=$subs $wl FAILED" | tee -a gpurun_out/r4_ab13.txt; tail -3 gpurun_out/ab_one.err; continue; }
      echo "rep$rep subs=$subs $wl $(python -c "import json;d=json.load(open('gpurun_out/ab_one.json'));r=d['roofline'];print(round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle; launch', round(r['us_per_launch'],1), d['config']['tile'][:40])")" | tee -a gpurun_out/r4_ab13.txt
    done
  done
done
