#!/bin/bash
# Round 4, call 38: inputs along the hand-off WITH the measured list of tiles: time against the default (alternating), traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r4_38_ab.txt
for i in 1 2 3; do
  for v in "default bench.py" "tp scripts/bench_with_lib.py build/ab/lib_tp.so"; do
    set -- $v
    N=$1; shift
    timeout -k 10 300 python "$@" --no-thermo --workload tenth > gpurun_out/r4_38.json 2> gpurun_out/r4_38.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_38.json'))
print('$N:', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_38_ab.txt
  done
done
O=gpurun_out/r4_38
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-dropin-timing --no-thermo --workload tenth"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python scripts/bench_with_lib.py build/ab/lib_tp.so --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
  C=$c python - <<'PY' | tee -a gpurun_out/r4_38_ab.txt
import csv, glob, os
c = os.environ["C"]
f = glob.glob(f"gpurun_out/r4_38/pmc_{c}/**/*counter_collection.csv", recursive=True)
tot = [0, 0.0]
for row in csv.DictReader(open(f[0])):
    if "k_subcycle_skew<4, false" in row["Kernel_Name"] and row["Counter_Name"] == c:
        tot[0] += 1; tot[1] += float(row["Counter_Value"])
print("inputs along the hand-off, measured list:", c, tot[0], "launches,", round(tot[1] / tot[0] / 1024 * (2 if c == "FETCH_SIZE" else 1), 1), "MB per launch")
PY
done
rm -rf $O
