#!/bin/bash
# Round 5, call 31: MPI jobs of the whole model (tripole slabs: PEER && FOLD); soak of the granule loop under both folds
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_step.py -q -m gpu -k "across_a_tripole_fold or mpi_job" > gpurun_out/r5_31_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_31_tests.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/r5_31_tests.log | cut -c1-300
grep -E "^E  " gpurun_out/r5_31_tests.log | cut -c1-600 | head -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/soak_fold.py 300 2>&1 | grep -v amdgpu.ids | tail -3
