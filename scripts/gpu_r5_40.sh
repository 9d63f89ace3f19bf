#!/bin/bash
# Round 5, call 40: the eight-rank rehearsals of the cross-rank loop, several times over (they were flaky while copy streams
# could take a hardware queue between two ranks' main streams)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2 3 4 5; do
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_evp.py -q -m gpu -k "(eight_ranks and peer) or cartesian_layout_of_one or (cartesian_layouts and peer) or (tripole_grid_cut_into_slabs and peer) or eliminated" > gpurun_out/r5_40_tests.log 2>&1
rc=$?; echo "rep $rep: $(grep -E 'passed|failed' gpurun_out/r5_40_tests.log | tail -1)"
[ $rc -eq 0 ] || { grep -E "^FAILED" gpurun_out/r5_40_tests.log | head; exit 1; }
done
