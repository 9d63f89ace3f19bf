#!/bin/bash
# copies spread over side streams: parity of everything that moves host arrays, then the PCIe-inclusive timings per setting
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py tests/test_gpu_thermo.py tests/test_gpu_transport.py tests/test_gpu_auscom.py tests/test_gpu_step.py -m gpu -x -q > gpurun_out/fan_tests.log 2>&1 || { grep -a -v "^ " gpurun_out/fan_tests.log | tail -30; exit 1; }
grep -a "passed\|failed" gpurun_out/fan_tests.log | tail -2
for n in 1 2 3 4; do
  echo "== CICE4_AMD_COPY_STREAMS=$n"
  CICE4_AMD_COPY_STREAMS=$n timeout -k 10 300 python scripts/pcie_evp.py 20 2>&1 | grep -a "PCIe\|upload"
  CICE4_AMD_COPY_STREAMS=$n timeout -k 10 300 python bench.py --no-tenth --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['pcie_inclusive']
print('evp %.3f ms  step_therm1 %.3f ms (abl %.3f)  transport %.3f ms' % (p['ms_per_call'], p['step_therm1']['ms_per_call'], p['step_therm1'].get('with_atmo_boundary_layer_on_device_ms', 0), p['transport_remap']['ms_per_call']))"
done
