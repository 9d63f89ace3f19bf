#!/bin/bash
# parity of the K-subcycle sweep + 0.1-degree timing per K (args: list of K; 0 = pair kernel)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -k "k_subcycles_per_sweep" > gpurun_out/skew_tests.log 2>&1
rc=$?; echo "skew tests rc=$rc"; tail -3 gpurun_out/skew_tests.log
[ $rc = 0 ] || exit 1
for K in ${@:-0 3 4 6}; do
  if [ $K = 0 ]; then opt="--no-skew"; else opt="--skew-levels $K"; fi
  timeout -k 10 300 python bench.py --workload tenth --steps 3 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing $opt $SKEW_EXTRA > gpurun_out/tenth_K$K.json 2> gpurun_out/tenth_K$K.err
  echo "K=$K rc=$? $(python -c "import json;d=json.load(open('gpurun_out/tenth_K$K.json'));print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['us_per_launch'],1), d['config']['tile'][:90])")"
done
