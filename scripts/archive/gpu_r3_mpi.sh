#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multiproc.py -m gpu -q -k "mpi_job" > gpurun_out/mpi_job.log 2>&1
grep -a "passed\|failed\|Error\|MPI-EVP\|abort\|cice4_amd" gpurun_out/mpi_job.log | cut -c1-500 | tail -14
