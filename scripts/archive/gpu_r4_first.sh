#!/bin/bash
# round 4, first call: in-kernel clock of both EVP loops (diagnostic build), slab costs at HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python scripts/inkernel_clock.py build/ab/lib_stamps.so gpurun_out/r04_inkernel_clock.csv 2>&1 | tee gpurun_out/r4_clock.txt &&
bash scripts/gpu_r4_slabs.sh
