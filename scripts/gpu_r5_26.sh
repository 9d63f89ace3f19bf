#!/bin/bash
# Round 5, call 26: cice_evp as a pipeline (keep_state / lazy_stresses)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_evp.py -x -q -m gpu -k "pcie or page_locked or three_steps or stepwise or whole_evp or degenerate" > gpurun_out/r5_26_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_26_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B60 "short test summary" gpurun_out/r5_26_tests.log | cut -c1-400 | tail -90; exit 1; }
timeout -k 10 300 python bench.py --no-thermo --no-tenth --no-cpu-baseline > gpurun_out/r5_26.json 2>gpurun_out/r5_26.err || { tail -20 gpurun_out/r5_26.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r5_26.json')); print(json.dumps(d['pcie_inclusive'], indent=1)); print(d['value'])"
