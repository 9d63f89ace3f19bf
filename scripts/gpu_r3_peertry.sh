#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py -m gpu -q -s -k "bench" > gpurun_out/peertry.log 2>&1
grep -a "passed\|failed\|two rank processes\|Error\|assert" gpurun_out/peertry.log | tail -12
