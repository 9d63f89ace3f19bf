#!/bin/bash
# Round 5, call 6: who runs when in the free-running loop (trace of one tile, diagnostic build)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_06_trace.txt
for cfg in "11 0" "11 2" "11 1" "4 2"; do
  timeout -k 10 200 python scripts/resident_trace.py build/ab/lib_stamps.so $cfg >> gpurun_out/r5_06_trace.txt 2>&1 || { tail -20 gpurun_out/r5_06_trace.txt; exit 1; }
done
cat gpurun_out/r5_06_trace.txt | grep -v amdgpu.ids
