#!/bin/bash
# A/B: one vs two subcycles per launch, workgroup heights
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
: > gpurun_out/fuse_bench.jsonl
for opt in "--no-fuse" "--fused-waves 8" "--fused-waves 12" "--fused-waves 13" "--fused-waves 14" "--fused-waves 16" ""; do
  for wl in gx1 tenth gx3 320x96; do
    extra=""; [ $wl = tenth ] && extra="--steps 2 --warmup 1"
    timeout -k 10 300 python bench.py --workload $wl $B $opt $extra >> gpurun_out/fuse_bench.jsonl 2>> gpurun_out/fuse_bench.err
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/fuse_bench.jsonl"):
    d = json.loads(l)
    r = d["roofline"]
    print(d["config"]["nx_global"], d["config"]["ny_global"], d["config"]["tile"][:52], "| value", round(d["value"], 1), "us/subcycle", round(r["us_per_launch"] / r["subcycles_per_launch"], 2))
PY
