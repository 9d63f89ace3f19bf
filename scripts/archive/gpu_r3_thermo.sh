#!/bin/bash
# thermo parity + rate against the sorting chunk
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/thermo_tests.log 2>&1
rc=$?; echo "thermo tests rc=$rc"; grep -E "passed|failed" gpurun_out/thermo_tests.log | tail -2
[ $rc = 0 ] || { grep -B5 -A25 "Error\|assert" gpurun_out/thermo_tests.log | head -60; exit 1; }
for rep in 1 2; do
for c in ${@:-0 512 2048}; do for g in ${GROUPS_:-8 16}; do
  CICE4_AMD_THERMO_GROUP=$g CICE4_AMD_THERMO_SORT=$c timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-tenth --no-cpu-baseline --no-dropin-timing > gpurun_out/th.json 2> gpurun_out/th.err || { echo "chunk $c FAILED"; tail -3 gpurun_out/th.err; continue; }
  echo "chunk=$c group=$g $(python -c "import json;d=json.load(open('gpurun_out/th.json'))['thermo'];print('G updates/s',round(d['value']/1e9,3),'ms/pass',round(d['ms_per_pass'],4))")"
done
done
done
