#!/bin/bash
# round 4, fourth call: the whole GPU suite, the default bench line, the shared-division probe, parity of the early-load sweep build
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4_full_tests.log | cut -c1-300; echo "full suite rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r4_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench.json"))
print("gx1", round(d["value"]), "subcycles/s; thermo", round(d["thermo"]["value"] / 1e9, 3), "G/s; tenth", round(d["tenth"]["value"], 1), "subcycles/s =",
      round(1e6 / d["tenth"]["value"], 1), "us; tenth thermo", round(d["tenth"]["thermo"]["value"] / 1e9, 3), "; pcie evp", round(d["pcie_inclusive"]["ms_per_call"], 2),
      "therm1", round(d["pcie_inclusive"]["step_therm1"]["ms_per_call"], 2), "remap", round(d["pcie_inclusive"]["transport_remap"]["ms_per_call"], 2))
PY
cd scripts/probe && /opt/rocm/bin/hipcc -O3 -ffp-contract=off --offload-arch=gfx950 div_shared_probe.hip -o /tmp/div_shared_probe 2>/dev/null && timeout -k 10 120 /tmp/div_shared_probe | tee "$GRAFT_REPO_ROOT/gpurun_out/r4_div_probe.txt"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python scripts/test_with_lib.py build/ab/lib_early.so tests/test_gpu_evp.py -x -q -k "k_subcycles_per_sweep or wide_halo or sweeps_on_a_tripole" 2>&1 | tail -1 | sed 's/^/early-load build parity: /' | tee gpurun_out/r4_early_parity.txt
