#!/bin/bash
# boundary-module drop-in tests + the two Fortran drop-in tests that share the drop-in library
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_boundary.py tests/test_gpu_evp.py tests/test_gpu_thermo.py -m gpu -x -q > gpurun_out/boundary.log 2>&1
tail -5 gpurun_out/boundary.log
