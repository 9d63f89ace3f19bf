#!/bin/bash
# Round 4, call 22: per-workgroup durations of several sweeps in a row (is the spread systematic?)
set -o pipefail
mkdir -p gpurun_out
CICE4_AMD_SKEW_FILL=0 timeout -k 10 200 python scripts/sweep_wg_times.py build/ab/lib_stamps.so gpurun_out/r4_22_wg_f0.npz 2>&1 | grep -v amdgpu.ids || exit 1
CICE4_AMD_SKEW_FILL=26 GEN_PCT=15 timeout -k 10 200 python scripts/sweep_wg_times.py build/ab/lib_stamps.so gpurun_out/r4_22_wg_f26g15.npz 2>&1 | grep -v amdgpu.ids || exit 1
