#!/bin/bash
# Round 5, call 16: whole-model MPI jobs with several blocks per task in the cross-rank loop; Fortran drop-in tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_step.py tests/test_gpu_multiproc.py tests/test_boundary.py tests/test_gpu_evp.py -x -q -m gpu -k "mpi or multiproc or fortran or boundary or Fortran or job" > gpurun_out/r5_16_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_16_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_16_tests.log | cut -c1-600 | tail -100; exit 1; }
