#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py tests/test_boundary.py tests/test_capi.py -m gpu -x -q > gpurun_out/pin_tests.log 2>&1 || { grep -v "^ " gpurun_out/pin_tests.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/pin_tests.log | tail -1
for wl in gx1 tenth; do
  extra=""; [ $wl = tenth ] && extra="--steps 2 --warmup 1"
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --no-thermo $extra > gpurun_out/pin.json 2> gpurun_out/pin.err
  python - $wl <<'PY'
import json, sys
d = json.loads(open("gpurun_out/pin.json").read().strip().splitlines()[-1])
print(sys.argv[1], "value", round(d["value"], 1), "pcie-inclusive ms/call", round(d["pcie_inclusive"]["ms_per_call"], 3))
PY
done
