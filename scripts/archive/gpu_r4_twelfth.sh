#!/bin/bash
# round 4, twelfth call: whole GPU suite + default bench + round-4 profiles at HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/r4_full_tests.log | tail -3 | cut -c1-300; echo "full suite rc=$rc"
[ $rc -eq 0 ] || { grep -B40 "short test summary" gpurun_out/r4_full_tests.log | tail -60 | cut -c1-250; exit $rc; }
bash scripts/gpu_profiles_r04.sh
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04prof/bench_gx1.json"))
r, t = d["roofline"], d["tenth"]["roofline"]
print("gx1", round(d["value"]), "subcycles/s; frac_valu", round(r.get("frac_valu_issue", 0), 3), "at clock", r.get("clock_ghz"), round(r.get("frac_valu_issue_at_measured_clock", 0), 3),
      "| thermo", round(d["thermo"]["value"] / 1e9, 3), "G/s | tenth", round(d["tenth"]["value"], 1), "=", round(1e6 / d["tenth"]["value"], 1), "us; frac", round(t["frac"], 3),
      "valu", round(t.get("frac_valu_issue", 0), 3), round(t.get("frac_valu_issue_at_measured_clock", 0), 3), "| pcie", round(d["pcie_inclusive"]["ms_per_call"], 2))
PY
