#!/bin/bash
# thermo parity tests + bench lines (thermo throughput is in the bench JSON's config block)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_thermo.py -m gpu -x -q > gpurun_out/thermo_tests.log 2>&1 || { tail -30 gpurun_out/thermo_tests.log; exit 1; }
tail -3 gpurun_out/thermo_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing > gpurun_out/bench_gx1_t.json 2> gpurun_out/bench_gx1_t.err
timeout -k 10 400 python bench.py --workload tenth --steps 2 --warmup 1 --no-cpu-baseline --no-dropin-timing > gpurun_out/bench_tenth_t.json 2> gpurun_out/bench_tenth_t.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_gx1_t.json", "gpurun_out/bench_tenth_t.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["frac"], d.get("thermo", {}).get("value"), d.get("thermo", {}).get("roofline"))
PY
