#!/bin/bash
# A/B: one vs two subcycles per launch, workgroup heights
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
: > gpurun_out/fuse_bench.jsonl
for opt in "--no-fuse" "--fused-waves 8" "--fused-waves 12" "--fused-waves 16" ""; do
  timeout -k 10 300 python bench.py $B $opt >> gpurun_out/fuse_bench.jsonl 2>> gpurun_out/fuse_bench.err
  timeout -k 10 300 python bench.py --workload tenth --steps 2 --warmup 1 $B $opt >> gpurun_out/fuse_bench.jsonl 2>> gpurun_out/fuse_bench.err
  timeout -k 10 300 python bench.py --workload gx3 $B $opt >> gpurun_out/fuse_bench.jsonl 2>> gpurun_out/fuse_bench.err
done
python - <<'PY'
import json
for l in open("gpurun_out/fuse_bench.jsonl"):
    d = json.loads(l)
    print(d["config"]["nx_global"], d["config"]["tile"][:60], "| value", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 3), "us/launch", round(d["roofline"]["us_per_launch"], 2))
PY
