#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in "1 0" "2 0" "2 1" "2 1" "2 0"; do
  set -- $cfg
  echo "== COPY_STREAMS=$1 EARLY_DOWNLOAD=$2"
  CICE4_AMD_COPY_STREAMS=$1 CICE4_AMD_EARLY_DOWNLOAD=$2 timeout -k 10 300 python bench.py --no-tenth --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['pcie_inclusive']
f = lambda x: '%.2f [%.2f, %.2f]' % (x['ms_per_call'], *x['ms_per_call_min_max'])
print('evp', f(p), ' step_therm1', f(p['step_therm1']), 'abl %.2f' % p['step_therm1'].get('with_atmo_boundary_layer_on_device_ms', 0), ' transport', f(p['transport_remap']))"
done
