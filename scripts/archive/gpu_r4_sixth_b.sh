#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for w in "96 70" "55 18" "111 14" "119 5" "95 14" "300 40"; do timeout -k 10 200 python scripts/skew_debug.py $w 2>&1 | grep -v "OK$" | grep -v amdgpu.ids | sed "s/^/[$w] /"; done
echo debug-done
bash scripts/gpu_r4_sixth.sh
