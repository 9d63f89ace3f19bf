#!/bin/bash
# round 4, sixth call: wider column strips of the sweep kernel -- parity (all strip-stride widths, tripole, slabs, 0.1-degree tests), then rates against the narrow-strip build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_evp.py tests/test_gpu_fullsize.py tests/test_gpu_auscom.py tests/test_gpu_multiproc.py -x -q -k "k_subcycles_per_sweep or wide_halo or sweeps_on_a_tripole or tenth or ranks_in_one_process or sweep or auscom or slabs or rank_processes or bench" > gpurun_out/r4_tests6.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4_tests6.log | tail -3 | cut -c1-300 | tee gpurun_out/r4_tests6.txt
grep -q "passed" gpurun_out/r4_tests6.txt && ! grep -q "failed" gpurun_out/r4_tests6.txt || { grep -B30 "short test summary" gpurun_out/r4_tests6.log | tail -45 | cut -c1-250; exit 1; }
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_ab6.txt
for rep in 1 2 3; do
  for lib in build/ab/lib_prio.so cice4_amd/libcice4_amd.so; do
    for wl in tenth 3600x316x240 1440x1080x240; do
      timeout -k 10 300 python scripts/bench_with_lib.py $lib --workload $wl $B > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "$lib $wl FAILED" | tee -a gpurun_out/r4_ab6.txt; tail -3 gpurun_out/ab_one.err; continue; }
      echo "rep$rep $lib $wl $(python -c "import json;d=json.load(open('gpurun_out/ab_one.json'));r=d['roofline'];print(round(d['value'],1), 'us/launch', round(r['us_per_launch'],1), 'us/subcycle', round(r['us_per_launch']/r['subcycles_per_launch'],2), d['config']['tile'][:60])")" | tee -a gpurun_out/r4_ab6.txt
    done
  done
done
