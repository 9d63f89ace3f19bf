#!/bin/bash
# Round 5, call 27: whole model with CICE4_AMD_KEEP_STATE; MPI jobs; Fortran drop-in tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_step.py tests/test_gpu_evp.py -x -q -m gpu -k "dropin or restart or fortran or standalone or pcie or mpi" > gpurun_out/r5_27_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_27_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_27_tests.log | cut -c1-400 | tail -90; exit 1; }
