#!/bin/bash
# the whole GPU suite three times in a row + the default bench line: looks for flakiness, not for speed
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/soak_$rep.log 2>&1 || { grep -v "^ " gpurun_out/soak_$rep.log | tail -30; exit 1; }
  grep -E "passed|failed" gpurun_out/soak_$rep.log | tail -1
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/soak_bench.json 2> gpurun_out/soak_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/soak_bench.json").read().strip().splitlines()[-1])
print("bench value", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 3))
PY
