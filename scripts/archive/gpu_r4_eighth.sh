#!/bin/bash
# round 4, eighth call: parity of the trimmed-extension / split sweeps on every multi-rank test; kernel cost of the split form at the deployment geometry (probe)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_evp.py tests/test_gpu_multiproc.py tests/test_gpu_fullsize.py -x -q -k "wide_halo or ranks_in_one_process or slabs or rank_processes or bench or tenth_degree_width_tripole" > gpurun_out/r4_tests8.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4_tests8.log | tail -3 | cut -c1-300 | tee gpurun_out/r4_tests8.txt
grep -q "passed" gpurun_out/r4_tests8.txt && ! grep -q "failed" gpurun_out/r4_tests8.txt || { grep -B30 "short test summary" gpurun_out/r4_tests8.log | tail -45 | cut -c1-250; exit 1; }
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_probe.txt
for rep in 1 2 3; do
  for cfg in "3600x316x240 0" "3600x316x240 8" "3600x300x240 0" "3600x308x240 4" "3600x616x240 0" "3600x616x240 8"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --workload $1 --skew-split-probe $2 $B > gpurun_out/probe_one.json 2> gpurun_out/probe_one.err || { echo "$cfg FAILED" | tee -a gpurun_out/r4_probe.txt; tail -3 gpurun_out/probe_one.err; continue; }
    echo "rep$rep $1 probe=$2: $(python -c "import json;d=json.load(open('gpurun_out/probe_one.json'));print(round(1e6/d['value'],2), 'us per subcycle')")" | tee -a gpurun_out/r4_probe.txt
  done
done
bash scripts/gpu_r4_split.sh
timeout -k 10 300 python scripts/resident_phases.py build/ab/lib_stamps.so gpurun_out/r04_resident_phases.csv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_phases.txt
