#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_step.py -q -m gpu -k "mpi_job_on_one_gpu" > gpurun_out/r5_35_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_35_tests.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/r5_35_tests.log | cut -c1-300
grep -E "^E  " gpurun_out/r5_35_tests.log | cut -c1-600 | head -12
exit $rc
