#!/bin/bash
# round 3, first GPU call: parity of the K-subcycle sweep, the self-launcher on a one-GPU box, 0.1-degree timing per K
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -k "k_subcycles_per_sweep" > gpurun_out/skew_tests.log 2>&1
echo "skew tests rc=$?" | tee -a gpurun_out/skew_tests.log
tail -5 gpurun_out/skew_tests.log
( time python bench.py --gpus 2 --workload gx3 ) > gpurun_out/launch2.log 2>&1; echo "launcher --gpus 2 rc=$?" | tee -a gpurun_out/launch2.log
( time CICE4_AMD_BENCH_DEVICE=0 python bench.py --gpus 2 --workload gx3 --comm-timeout 40 --no-cpu-baseline --no-thermo --no-tenth --no-dropin-timing ) > gpurun_out/launch2_samedev.log 2>&1; echo "two ranks on one device rc=$?" | tee -a gpurun_out/launch2_samedev.log
tail -3 gpurun_out/launch2.log gpurun_out/launch2_samedev.log
for K in 0 2 3 4 6 8; do
  if [ $K = 0 ]; then opt="--no-skew"; else opt="--skew-levels $K"; fi
  timeout -k 10 300 python bench.py --workload tenth --steps 3 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing $opt > gpurun_out/tenth_K$K.json 2> gpurun_out/tenth_K$K.err
  echo "K=$K rc=$? $(python -c "import json;d=json.load(open('gpurun_out/tenth_K$K.json'));print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['config']['tile'])")"
done
