#!/bin/bash
# Round 5, call 28: the cross-rank one-launch loop with the tripole fold inside (PEER && FOLD)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_evp.py tests/test_gpu_fullsize.py -q -m gpu -k "tripole_grid_cut_into_slabs or cartesian_layouts or ranks_in_one_process or eliminated or eight_ranks or cartesian_layout_of_one" > gpurun_out/r5_28_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_28_tests.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/r5_28_tests.log | cut -c1-300
grep -E "^E  " gpurun_out/r5_28_tests.log | cut -c1-400 | head -30
grep -E "^cice4_amd:" gpurun_out/r5_28_tests.log | cut -c1-300 | head -20
exit $rc
