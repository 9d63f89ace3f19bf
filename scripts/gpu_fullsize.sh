#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=5 > gpurun_out/fullsize_tests.log 2>&1 || { grep -v "^ " gpurun_out/fullsize_tests.log | tail -40; exit 1; }
grep -v "^ " gpurun_out/fullsize_tests.log | tail -12
