#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scripts/sweep_balance_trace.py full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_28_trace_full.txt
