#!/bin/bash
# Round 4, last call: smoke(), the whole GPU suite, soak of the sweep, round-4 profiles and the default bench line at HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" 2>&1 | tail -2 | cut -c1-200
bash scripts/gpu_r4_14.sh
