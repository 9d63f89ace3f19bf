#!/bin/bash
# Round 5, call 15: the cross-rank one-launch loop with several blocks per rank
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "cartesian or ranks_in_one_process or eliminated" > gpurun_out/r5_15_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_15_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_15_tests.log | cut -c1-500 | tail -100; exit 1; }
