#!/bin/bash
# Round 4, call 20: how long the sweep's workgroups run on CUs that hold two / three of them, with and without longer segments for the former
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_20_placement.txt
for d in "0 0" "1 0" "1 10" "1 20"; do
  set -- $d
  CICE4_AMD_SKEW_FILL=$1 GEN_PCT=$2 timeout -k 10 200 python scripts/sweep_placement.py build/ab/lib_stamps.so 2>&1 | grep -v "amdgpu.ids\|^   (\|levels sharing\|example CU" | tee -a gpurun_out/r4_20_placement.txt || exit 1
done
