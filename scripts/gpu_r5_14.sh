#!/bin/bash
# Round 5, call 14: the whole-model MPI jobs (2 x 2 tasks in one launch per task), per-rank costs N = 1 .. 8 on one box, several blocks on one rank
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_multiproc.py -x -q -m gpu > gpurun_out/r5_14_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_14_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B70 "short test summary" gpurun_out/r5_14_tests.log | cut -c1-600 | tail -100; exit 1; }
: > gpurun_out/r5_14_rank_costs.jsonl
for n in 1 2 4 8; do
  timeout -k 10 900 python scripts/rank_costs.py --workload tenth --ranks $n 2>gpurun_out/r5_14_rc.err | tee -a gpurun_out/r5_14_rank_costs.jsonl || { tail -20 gpurun_out/r5_14_rc.err; exit 1; }
done
timeout -k 10 600 python scripts/rank_costs.py --workload gx1 --ranks 1,8 2>gpurun_out/r5_14_rc.err | tee -a gpurun_out/r5_14_rank_costs.jsonl || { tail -20 gpurun_out/r5_14_rc.err; exit 1; }
: > gpurun_out/r5_14_blocks.txt
for cfg in "320 384 320 384" "320 384 160 192" "320 384 320 96" "100 116 50 58" "100 116 10 10"; do
  timeout -k 10 200 python scripts/blocks_rate.py $cfg 2>/dev/null | tee -a gpurun_out/r5_14_blocks.txt
done
