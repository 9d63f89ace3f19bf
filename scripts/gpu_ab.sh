#!/bin/bash
# quick A/B of the subcycle kernels: tests, then gx1 / 0.1 degree / gx3 / one-of-8-ranks slab
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -m gpu -x -q > gpurun_out/evp_tests.log 2>&1 || { grep -v "^ " gpurun_out/evp_tests.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/evp_tests.log | tail -1
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
for wl in gx1 gx1 gx3 320x96 tenth; do
  extra=""; [ $wl = tenth ] && extra="--steps 3 --warmup 1"
  timeout -k 10 300 python bench.py --workload $wl $B $extra > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - $wl <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "value", round(d["value"], 1), "us/subcycle", round(r["us_per_launch"] / r["subcycles_per_launch"], 3), "frac", round(r["frac"], 3))
PY
done
