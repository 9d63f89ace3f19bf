#!/bin/bash
# Round 4, call 37: the configuration that moves the fewest bytes: inputs along the hand-off (-DSKEW_TPASS=1 build), equal segments
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r4_37
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-dropin-timing --no-thermo --workload tenth --skew-gen-pct 0 --skew-balance 0"
: > gpurun_out/r4_37_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  CICE4_AMD_SKEW_FILL=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python scripts/bench_with_lib.py build/ab/lib_tp.so --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
  C=$c python - <<'PY' | tee -a gpurun_out/r4_37_traffic.txt
import csv, glob, os
c = os.environ["C"]
f = glob.glob(f"gpurun_out/r4_37/pmc_{c}/**/*counter_collection.csv", recursive=True)
tot = [0, 0.0]
for row in csv.DictReader(open(f[0])):
    if "k_subcycle_skew<4, false" in row["Kernel_Name"] and row["Counter_Name"] == c:
        tot[0] += 1; tot[1] += float(row["Counter_Value"])
print("inputs along the hand-off, equal segments:", c, tot[0], "launches,", round(tot[1] / tot[0] / 1024 * (2 if c == "FETCH_SIZE" else 1), 1), "MB per launch")
PY
done
for v in "tp scripts/bench_with_lib.py build/ab/lib_tp.so" "default_kernel bench.py"; do
  set -- $v
  N=$1; shift
  CICE4_AMD_SKEW_FILL=0 timeout -k 10 300 python "$@" $B > gpurun_out/r4_37.json 2> gpurun_out/r4_37.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/r4_37.json'))
print('$N, equal segments:', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_37_traffic.txt
done
timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_37.json 2> gpurun_out/r4_37.err || exit 1
python -c "
import json
d=json.load(open('gpurun_out/r4_37.json'))
print('default:', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_37_traffic.txt
rm -rf $O
