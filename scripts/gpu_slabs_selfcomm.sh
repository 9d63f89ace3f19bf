#!/bin/bash
# the N-rank decomposition (slabs, overlap rows, paired launches, refresh through pack/RCCL/unpack) on ONE GPU:
# all slabs on this device, messages to the own rank.  Checks that the multi-rank configuration bench.py picks runs.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing --no-thermo --steps 5 --warmup 1"
for n in 2 4 8; do
  CICE4_AMD_SELF_COMM=1 timeout -k 10 300 python bench.py $B --slabs $n > gpurun_out/slabs.json 2> gpurun_out/slabs.err || { tail -5 gpurun_out/slabs.err; exit 1; }
  python - $n <<'PY'
import json, sys
d = json.loads(open("gpurun_out/slabs.json").read().strip().splitlines()[-1])
print("slabs", sys.argv[1], d["config"]["decomposition"][-90:], "| tile", d["config"]["tile"][:40], "| us/subcycle (all slabs on one GPU)", round(1e6 / d["value"], 2))
PY
done
CICE4_AMD_SELF_COMM=1 timeout -k 10 300 python bench.py --workload tenth --steps 1 --warmup 1 --no-cpu-baseline --no-dropin-timing --no-thermo --slabs 8 > gpurun_out/slabs.json 2> gpurun_out/slabs.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/slabs.json").read().strip().splitlines()[-1])
print("tenth slabs 8", d["config"]["decomposition"][-90:], "| us/subcycle", round(1e6 / d["value"], 2))
PY
