#!/bin/bash
# Round 4, call 49: the tile map by ice cover under a fold as well: parity, gx1 tripole by cover; then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "tile_map" > gpurun_out/r4_49_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_49_tests.log | tail -2 | cut -c1-200
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_49_tests.log | cut -c1-300 | tail -60; exit 1; }
: > gpurun_out/r4_49.txt
for c in full caps; do
  for m in 0 -1; do
    if [ $m = 0 ]; then export CICE4_AMD_RESIDENT_MAP=0; else unset CICE4_AMD_RESIDENT_MAP; fi
    timeout -k 10 200 python bench.py --north tripole --no-thermo --no-tenth --cover $c > gpurun_out/r4_49.json 2>/dev/null || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_49.json')); print('gx1 tripole cover $c map $m:', round(d['value']), 'subcycles/s =', round(1e6/d['value'],2), 'us per subcycle')" | tee -a gpurun_out/r4_49.txt
  done
done
unset CICE4_AMD_RESIDENT_MAP
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_49_full.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_49_full.log | tail -2 | cut -c1-200
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_49_full.log | cut -c1-300 | tail -60; exit 1; }
