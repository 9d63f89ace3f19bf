#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/all_gpu_tests.log 2>&1 || { grep -v "^ " gpurun_out/all_gpu_tests.log | tail -40; exit 1; }
grep -v "^ " gpurun_out/all_gpu_tests.log | tail -8
