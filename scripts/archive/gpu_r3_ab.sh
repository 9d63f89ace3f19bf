#!/bin/bash
# A/B of library builds under build/ab/*.so at 0.1 degree: usage gpu_r3_ab.sh "<bench args>" lib1 lib2 ...  (each lib run REPS times, interleaved)
set -o pipefail
mkdir -p gpurun_out
args="$1"; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    timeout -k 10 300 python scripts/bench_with_lib.py build/ab/$lib.so --workload tenth --steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing $args > gpurun_out/ab_$lib.json 2> gpurun_out/ab_$lib.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab_$lib.err; continue; }
    echo "rep$rep $lib $(python -c "import json;d=json.load(open('gpurun_out/ab_$lib.json'));print(round(d['value'],1), round(d['roofline']['us_per_launch'],1))")"
  done
done
