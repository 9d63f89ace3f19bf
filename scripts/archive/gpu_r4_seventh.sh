#!/bin/bash
# round 4, seventh call: split-sweep cost on one GPU; soaks with fine-grained peer buffers / the new sweep; whole-model timers with and without the evp -> transport chain
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash scripts/gpu_r4_split.sh
timeout -k 10 400 python scripts/soak_peer.py 2 500 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee gpurun_out/r4_soak.txt
timeout -k 10 400 python scripts/soak_peer.py 3 300 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee -a gpurun_out/r4_soak.txt
timeout -k 10 400 python scripts/soak_sweep.py 2>&1 | tail -3 | cut -c1-300 | tee -a gpurun_out/r4_soak.txt
for chain in 0 1; do
  CICE4_AMD_CHAIN=$chain timeout -k 10 400 python scripts/driver_timers.py 48 dropin 2>&1 | grep -E "dropin|Step|Dynamics|Advection|Column|Thermo|Bound" | sed "s/^/chain=$chain /" | tee -a gpurun_out/r4_driver_timers.txt
done
timeout -k 10 300 python scripts/driver_timers.py 24 dropin gx3 2>&1 | grep -E "dropin|Step|Dynamics|Advection|Thermo|Bound" | tee -a gpurun_out/r4_driver_timers.txt
