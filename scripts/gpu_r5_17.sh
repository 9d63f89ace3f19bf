#!/bin/bash
# Round 5, call 17: the round-5 profile set (scripts/gpu_profiles_r05.sh) + the in-kernel clock (diagnostic build)
cd "$GRAFT_REPO_ROOT"
bash scripts/gpu_profiles_r05.sh all
timeout -k 10 300 python scripts/inkernel_clock.py build/ab/lib_stamps.so gpurun_out/r05prof/inkernel_clock.csv > gpurun_out/r05prof/inkernel_clock.log 2>&1 || echo "clock failed"
tail -3 gpurun_out/r05prof/inkernel_clock.log
ls gpurun_out/r05prof | head -40
