#!/bin/bash
# timing at 0.1 degree for a list of bench option strings (one per argument), repeated twice
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for o in "$@"; do
  timeout -k 10 300 python bench.py --workload tenth --steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing $o > gpurun_out/opt.json 2> gpurun_out/opt.err || { echo "[$o] FAILED"; tail -3 gpurun_out/opt.err; continue; }
  echo "[$o] $(python -c "import json;d=json.load(open('gpurun_out/opt.json'));print('value',round(d['value'],1),'us/launch',round(d['roofline']['us_per_launch'],1))")"
done
done
