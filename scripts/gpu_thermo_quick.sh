#!/bin/bash
# thermo throughput: spatially organised regimes (bench default) vs white noise
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for L in 24 8 0; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-dropin-timing --thermo-coherence $L > gpurun_out/tq_gx1.json 2> gpurun_out/tq.err
timeout -k 10 400 python bench.py --workload tenth --steps 1 --warmup 1 --no-cpu-baseline --no-dropin-timing --thermo-coherence $L > gpurun_out/tq_tenth.json 2>> gpurun_out/tq.err
python - $L <<'PY'
import json, sys
for w in ("gx1", "tenth"):
    d = json.loads(open(f"gpurun_out/tq_{w}.json").read().strip().splitlines()[-1])
    print("coherence", sys.argv[1], w, "thermo", round(d["thermo"]["value"] / 1e9, 3), "G/s  evp", round(d["value"], 1))
PY
done
