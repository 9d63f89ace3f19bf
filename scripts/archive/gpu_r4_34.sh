#!/bin/bash
# Round 4, call 34: aligned list of tiles (a band of tiles on one dispatch order and XCD pair), one set of boundaries for all strips
# where the strips are alike: parity, traces, reads (PMC) and time against the natural order on the same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu > gpurun_out/r4_34_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r4_34_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_34_tests.log | cut -c1-300 | tail -60; exit 1; }
for c in full caps; do
  timeout -k 10 300 python scripts/sweep_balance_trace.py $c 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_34_trace_$c.txt | grep -v "^   strip " | cut -c1-420 | tail -2
done
O=gpurun_out/r4_34
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-dropin-timing --no-thermo"
: > gpurun_out/r4_34_reads.txt
for v in "natural 0" "aligned 1"; do
  set -- $v
  CICE4_AMD_SKEW_ALIGN=$2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$1 -- python bench.py --workload tenth --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/pmc_$1.log 2>&1 || echo "pmc $1 failed"
  V=$1 python - <<'PY' | tee -a gpurun_out/r4_34_reads.txt
import csv, glob, collections, os
v = os.environ["V"]
f = glob.glob(f"gpurun_out/r4_34/pmc_{v}/**/*counter_collection.csv", recursive=True)
tot = [0, 0.0]
for row in csv.DictReader(open(f[0])):
    if "k_subcycle_skew<4, false" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
        tot[0] += 1; tot[1] += float(row["Counter_Value"])
print(v, tot[0], "launches, reads", round(tot[1] / tot[0] / 1024 * 2, 1), "MB per launch")
PY
  for i in 1 2; do
    CICE4_AMD_SKEW_ALIGN=$2 timeout -k 10 300 python bench.py --no-thermo --workload tenth > gpurun_out/r4_34.json 2> gpurun_out/r4_34.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/r4_34.json'))
print('$1:', round(1e6/d['value'],1), 'us per subcycle')
" | tee -a gpurun_out/r4_34_reads.txt
  done
done
rm -rf $O
