#!/bin/bash
# Round 5, call 32: the whole GPU suite, then the default bench line (archived as profiles/r05_bench_gx1.json)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_32_tests.log 2>&1
rc=$?; grep -E "passed|failed|error" gpurun_out/r5_32_tests.log | tail -2
[ $rc -eq 0 ] || { grep -B120 "short test summary" gpurun_out/r5_32_tests.log | cut -c1-700 | tail -150; exit 1; }
timeout -k 10 500 python bench.py > gpurun_out/r5_32_bench.json 2> gpurun_out/r5_32_bench.err || { tail -20 gpurun_out/r5_32_bench.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r5_32_bench.json')); r=d['roofline']
print('gx1', round(d['value']), 'subcycles/s; stale:', d['counters_stale'], '; frac_valu_issue', round(r['frac_valu_issue'],3), 'at clock', round(r['frac_valu_issue_at_measured_clock'],3), 'counting polls', round(r['frac_valu_issue_counting_the_polls'],3))
print('tenth', round(d['tenth']['value'],1), d['tenth']['roofline']['frac_valu_issue_at_measured_clock'])"
