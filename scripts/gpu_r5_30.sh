#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -q -m gpu -k "tripole_grid_cut_into_slabs and peer" > gpurun_out/r5_30_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_30_tests.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/r5_30_tests.log | cut -c1-300
grep -E "^cice4_amd:" gpurun_out/r5_30_tests.log | cut -c1-400 | head -20
exit $rc
