#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_evp.py -m gpu -q -k "ranks_in_one_process" > gpurun_out/nscyc.log 2>&1
grep -a "passed\|failed\|Error\|assert" gpurun_out/nscyc.log | cut -c1-500 | tail -12
