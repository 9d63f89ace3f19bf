#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -x -q -k "ranks_in_one_process or sorted_columns" tests/test_gpu_thermo.py > gpurun_out/ranks_tests.log 2>&1
rc=$?; echo "rc=$rc"; grep -E "passed|failed" gpurun_out/ranks_tests.log | tail -2
[ $rc = 0 ] || { grep -v "^ \|Domain\|^$" gpurun_out/ranks_tests.log | tail -60; }
