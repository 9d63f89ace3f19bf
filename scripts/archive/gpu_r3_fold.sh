#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -m gpu -x -q -k "sweeps_on_a_tripole" > gpurun_out/fold_tests.log 2>&1
grep -a "passed\|failed\|Error\|assert\|cice4_amd:" gpurun_out/fold_tests.log | cut -c1-600 | tail -14
