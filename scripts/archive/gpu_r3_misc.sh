#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_thermo.py tests/test_boundary.py tests/test_gpu_step.py -x -q -m gpu > gpurun_out/misc_tests.log 2>&1
rc=$?; echo "rc=$rc"; grep -E "passed|failed" gpurun_out/misc_tests.log | tail -2
[ $rc = 0 ] || { grep -v "^ \|Domain\|^$" gpurun_out/misc_tests.log | tail -40; exit 1; }
python scripts/driver_timers.py 24 dropin,ref gx1 > gpurun_out/driver_timers_gx1.log 2>&1; tail -30 gpurun_out/driver_timers_gx1.log
python scripts/driver_timers.py 24 dropin,ref gx3 > gpurun_out/driver_timers_gx3.log 2>&1; tail -30 gpurun_out/driver_timers_gx3.log
