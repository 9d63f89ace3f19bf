#!/bin/bash
# round 4, ninth call: the whole GPU suite at HEAD, two rank processes of the 0.1-degree bench on one GPU (shm link), the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/r4_full_tests.log | tail -3 | cut -c1-300; echo "full suite rc=$rc"
[ $rc -eq 0 ] || { grep -B30 "short test summary" gpurun_out/r4_full_tests.log | tail -45 | cut -c1-250; exit $rc; }
CICE4_AMD_BENCH_DEVICE=0 CICE4_AMD_BENCH_LINK=shm timeout -k 10 500 python bench.py --gpus 2 --workload tenth --steps 2 --warmup 1 --no-thermo > gpurun_out/r4_two_ranks_tenth.json 2> gpurun_out/r4_two_ranks_tenth.err; echo "two ranks rc=$?"
grep -a "refresh overlap\|overlap rows" gpurun_out/r4_two_ranks_tenth.err | tail -3 | cut -c1-400
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_two_ranks_tenth.json"))
c = d["config"]
print("two rank processes on ONE GPU, 0.1 degree:", round(d["value"], 1), "subcycles/s;", c["decomposition"], "| launches per step", c.get("launches_per_step"),
      "| most common launch runs", c.get("subcycles_in_the_most_common_launch"), "subcycles |", c.get("refresh_overlap"))
PY
timeout -k 10 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench.json"))
r, t = d["roofline"], d["tenth"]["roofline"]
print("gx1", round(d["value"]), "subcycles/s; frac_valu", round(r.get("frac_valu_issue", 0), 3), "at clock", r.get("clock_ghz"), round(r.get("frac_valu_issue_at_measured_clock", 0), 3),
      "| thermo", round(d["thermo"]["value"] / 1e9, 3), "G/s | tenth", round(d["tenth"]["value"], 1), "=", round(1e6 / d["tenth"]["value"], 1), "us; frac", round(t["frac"], 3),
      "valu", round(t.get("frac_valu_issue", 0), 3), round(t.get("frac_valu_issue_at_measured_clock", 0), 3), "| timed_region_s", d.get("timed_region_s"))
PY
