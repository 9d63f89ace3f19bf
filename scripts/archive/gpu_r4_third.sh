#!/bin/bash
# round 4, third call: several blocks per rank in the one-launch loop, evp -> transport chain, whole-model runs; sweep-kernel A/B (early loads)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py tests/test_gpu_transport.py -x -q -k "several_blocks or whole_loop_in_one_launch or pcie_round_trip or remap_equals or resident_loop_is_repeatable" 2>&1 | tail -5 | tee gpurun_out/r4_tests3.txt &&
timeout -k 10 1200 python -m pytest tests/test_gpu_step.py -x -q 2>&1 | tail -5 | tee gpurun_out/r4_tests3b.txt &&
for a in "320 384 160 192" "320 384 320 96" "320 384 80 384" "100 116 50 58" "100 116 10 10"; do timeout -k 10 200 python scripts/blocks_rate.py $a 2>&1 | grep "per subcycle" | tee -a gpurun_out/r4_blocks_rate.txt; done &&
bash scripts/gpu_r4_ab.sh prio tprio early eprio
