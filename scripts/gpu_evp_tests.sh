#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -m gpu -x -q > gpurun_out/evp_tests.log 2>&1 || { grep -v "^ " gpurun_out/evp_tests.log | tail -40; exit 1; }
grep -v "^ " gpurun_out/evp_tests.log | tail -5
