#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu > gpurun_out/r4_25_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_25_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B30 "short test summary" gpurun_out/r4_25_tests.log | cut -c1-300 | tail -50; exit 1; }
timeout -k 10 300 python scripts/sweep_balance_trace.py caps 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_25_trace.txt
