#!/bin/bash
# round 4: A/B of sweep-kernel builds (build/ab/lib_<name>.so): parity of each on the sweep tests, then 0.1-degree and 300-row-slab rates, interleaved
# usage: gpu_r4_ab.sh name1 name2 ...
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r4_ab.txt
for lib in "$@"; do
  timeout -k 10 400 python scripts/test_with_lib.py build/ab/lib_$lib.so tests/test_gpu_evp.py -x -q -k "k_subcycles_per_sweep or wide_halo or sweeps_on_a_tripole" > gpurun_out/r4_ab_test_$lib.log 2>&1
  echo "parity $lib rc=$? $(tail -1 gpurun_out/r4_ab_test_$lib.log)" | tee -a gpurun_out/r4_ab.txt
done
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
for rep in 1 2; do
  for lib in "$@"; do
    for wl in tenth 3600x316x240; do
      timeout -k 10 300 python scripts/bench_with_lib.py build/ab/lib_$lib.so --workload $wl $B > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "$lib $wl FAILED" | tee -a gpurun_out/r4_ab.txt; tail -3 gpurun_out/ab_one.err; continue; }
      echo "rep$rep $lib $wl $(python -c "import json;d=json.load(open('gpurun_out/ab_one.json'));r=d['roofline'];print(round(d['value'],1), 'us/launch', round(r['us_per_launch'],1), 'us/subcycle', round(r['us_per_launch']/r['subcycles_per_launch'],2))")" | tee -a gpurun_out/r4_ab.txt
    done
  done
done
