#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_boundary.py tests/test_gpu_transport.py -m gpu -q > gpurun_out/tfold_tests.log 2>&1 || { grep -a -v "^ \|^E  *[a-zA-Z(].*[=:] " gpurun_out/tfold_tests.log | tail -40; exit 1; }
grep -a "passed\|failed" gpurun_out/tfold_tests.log | tail -2
timeout -k 10 300 python scripts/tripole_rate.py 2>&1 | grep -a "us per"
