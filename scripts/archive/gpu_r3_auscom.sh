#!/bin/bash
# the coupled flavour on the GPU: its own tests, then the stand-alone EVP / thermo suites (shared sources)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_auscom.py -m gpu -x -q -s > gpurun_out/auscom_tests.log 2>&1 || { grep -v "^ " gpurun_out/auscom_tests.log | tail -40; exit 1; }
grep -v "^ " gpurun_out/auscom_tests.log | tail -5
