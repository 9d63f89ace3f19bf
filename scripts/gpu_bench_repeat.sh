#!/bin/bash
# default bench line, repeated: run-to-run spread on one box
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2 3 4; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing > gpurun_out/rep.json 2> gpurun_out/rep.err
  python - $rep <<'PY'
import json, sys
d = json.loads(open("gpurun_out/rep.json").read().strip().splitlines()[-1])
print("rep", sys.argv[1], "value", round(d["value"], 1), "us/launch", round(d["roofline"]["us_per_launch"], 2), "thermo", round(d["thermo"]["value"] / 1e9, 3))
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing --ramp-seconds 0 > gpurun_out/rep.json 2> gpurun_out/rep.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/rep.json").read().strip().splitlines()[-1])
print("no ramp: value", round(d["value"], 1))
PY
