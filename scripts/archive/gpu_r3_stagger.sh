#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
K=$1; shift
for ns in "$@"; do
  timeout -k 10 300 python bench.py --workload tenth --steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --skew-levels $K --skew-stagger-ns $ns $SKEW_EXTRA > gpurun_out/stg.json 2> gpurun_out/stg.err || { echo "ns $ns FAILED"; tail -3 gpurun_out/stg.err; continue; }
  echo "K=$K stagger=$ns $(python -c "import json;d=json.load(open('gpurun_out/stg.json'));print('value',round(d['value'],1),'us/launch',round(d['roofline']['us_per_launch'],1))")"
done
