#!/bin/bash
# Round 4, call 42: the reference's own timers with the drop-in modules, gx1 size, 48 steps: tile map fixed (0) against chosen by the cover
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_42_driver_timers.txt
for m in 0 auto 0 auto; do
  if [ $m = auto ]; then unset CICE4_AMD_RESIDENT_MAP; else export CICE4_AMD_RESIDENT_MAP=$m; fi
  CICE4_AMD_CHAIN=1 timeout -k 10 400 python scripts/driver_timers.py 48 dropin 2>&1 | grep -E "dropin|Step|Dynamics|Advection" | sed "s/^/map=$m /" | tee -a gpurun_out/r4_42_driver_timers.txt
done
