#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 scripts/soak_fold_granules_w11.py ${1:-cice4_amd/libcice4_amd.so} ${2:-1,2,3,4,8} > gpurun_out/r5_24.txt 2> gpurun_out/r5_24.err
echo rc=$?
cat gpurun_out/r5_24.txt; head -20 gpurun_out/r5_24.err
