#!/bin/bash
# Round 5, call 5: the hand-off latency itself (store of a granule -> its arrival at the polling lane), diagnostic build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for pd in 0 3; do
  echo "== poll delay $pd"
  CICE4_AMD_RESIDENT_POLL_DELAY=$pd timeout -k 10 300 python scripts/resident_phases.py build/ab/lib_stamps.so gpurun_out/r5_05_phases_$pd.csv > gpurun_out/r5_05_phases_$pd.txt 2>&1 || { tail -20 gpurun_out/r5_05_phases_$pd.txt; exit 1; }
  grep -v "^   [7]:" gpurun_out/r5_05_phases_$pd.txt | grep -A9 "granules 1"
done
