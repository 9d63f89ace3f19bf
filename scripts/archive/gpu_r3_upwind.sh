#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_transport.py "tests/test_gpu_step.py::test_whole_model_with_upwind_advection" -m gpu -q > gpurun_out/upwind_tests.log 2>&1 || { grep -a -v "^ " gpurun_out/upwind_tests.log | tail -40; exit 1; }
grep -a "passed\|failed" gpurun_out/upwind_tests.log | tail -2
