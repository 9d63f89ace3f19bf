#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_thermo.py tests/test_gpu_step.py tests/test_gpu_atmo.py tests/test_gpu_auscom.py -m gpu -x -q > gpurun_out/thermo_tests.log 2>&1 || { grep -a -v "^ " gpurun_out/thermo_tests.log | tail -30; exit 1; }
grep -a "passed\|failed" gpurun_out/thermo_tests.log | tail -2
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -a "smoke\|rror" | tail -3
for e in 0 1 0 1; do
  echo "== CICE4_AMD_EARLY_DOWNLOAD=$e"
  CICE4_AMD_EARLY_DOWNLOAD=$e timeout -k 10 300 python scripts/pcie_evp.py 60 2>&1 | grep -a "PCIe"
done
timeout -k 10 300 python bench.py --no-tenth --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['pcie_inclusive']
print('evp %.3f ms  step_therm1 %.3f ms (abl %.3f)  transport %.3f ms  per-call thermo: %s' % (p['ms_per_call'], p['step_therm1']['ms_per_call'], p['step_therm1'].get('with_atmo_boundary_layer_on_device_ms', 0), p['transport_remap']['ms_per_call'], {k: v for k, v in p.items() if 'thermo' in k}))"
