#!/bin/bash
# Round 5, call 29: gx1 on eight ranks under a tripole fold (PEER && FOLD on rank 7)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python tests/ranks_peer_case.py 1 8 320 384 120 3 > gpurun_out/r5_29.txt 2> gpurun_out/r5_29.err
echo rc=$?
grep -v amdgpu.ids gpurun_out/r5_29.err | cut -c1-600 | head -30; tail -3 gpurun_out/r5_29.txt
