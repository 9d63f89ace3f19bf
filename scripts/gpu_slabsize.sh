#!/bin/bash
# kernel-only time of the slab one rank of N works on at gx1 (overlap rows included), and 0.1 degree over 8
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
: > gpurun_out/slabsize.jsonl
for wl in 320x384 320x232 320x144 320x96 3600x312; do
  timeout -k 10 300 python bench.py --workload $wl $B >> gpurun_out/slabsize.jsonl 2>> gpurun_out/slabsize.err
done
python - <<'PY'
import json
for l in open("gpurun_out/slabsize.jsonl"):
    d = json.loads(l)
    r = d["roofline"]
    print(d["config"]["nx_global"], d["config"]["ny_global"], d["config"]["tile"][:44], "| us/subcycle", round(r["us_per_launch"] / r["subcycles_per_launch"], 2), "value", round(d["value"]))
PY
