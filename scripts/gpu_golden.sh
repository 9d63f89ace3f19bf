#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -s > gpurun_out/golden_tests.log 2>&1 || { grep -v "^ " gpurun_out/golden_tests.log | tail -40; exit 1; }
grep -v "^ " gpurun_out/golden_tests.log | tail -6
