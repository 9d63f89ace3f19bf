#!/bin/bash
# Round 4, call 32: extra tiles to runs of neighbouring strips: trace (which strips), traffic (PMC), bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python scripts/sweep_balance_trace.py full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_32_trace_full.txt | grep -v "^   strip " | cut -c1-400 | tail -2
python - <<'PY'
import numpy as np
PY
O=gpurun_out/r4_32
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --workload tenth --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/r4_32/pmc_{c}/**/*counter_collection.csv", recursive=True)
    tot = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if "k_subcycle_skew" in row["Kernel_Name"] and row["Counter_Name"] == c:
            k = row["Kernel_Name"][:40]
            tot[k][0] += 1; tot[k][1] += float(row["Counter_Value"])
    for k, (n, v) in tot.items():
        print(c, k, n, "launches", round(v / n / 1024 * (2 if c == "FETCH_SIZE" else 1), 1), "MB per launch")
PY
find $O -name "*.csv" -size +8M -delete; find $O -name "*.db" -delete
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_32.json 2> gpurun_out/r4_32.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/r4_32.json'))
print('cover full:', round(d['value'],1), 'subcycles/s =', round(1e6/d['value'],1), 'us per subcycle')
"
done
