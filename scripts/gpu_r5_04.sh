#!/bin/bash
# Round 5, call 4: passes of the granule poll, first-pass time (diagnostic build)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for pd in 0 2 4; do
  echo "== poll delay $pd, prio 2"
  CICE4_AMD_RESIDENT_POLL_DELAY=$pd timeout -k 10 300 python scripts/resident_phases.py build/ab/lib_stamps.so gpurun_out/r5_04_phases_$pd.csv > gpurun_out/r5_04_phases_$pd.txt 2>&1 || { tail -20 gpurun_out/r5_04_phases_$pd.txt; exit 1; }
  grep -v "^   [57]:" gpurun_out/r5_04_phases_$pd.txt
done
