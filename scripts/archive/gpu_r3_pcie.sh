#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -m gpu -x -q -k "resident or page_locked" > gpurun_out/resident_tests.log 2>&1 || { grep -v "^ " gpurun_out/resident_tests.log | tail -30; exit 1; }
grep -a "passed\|failed" gpurun_out/resident_tests.log | tail -2
timeout -k 10 300 python scripts/pcie_evp.py 20 2>&1 | grep -v "^ \|^$" | tee gpurun_out/pcie_evp.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pcie_prof -o pcie -- python3 $GRAFT_REPO_ROOT/scripts/pcie_evp.py 6 > $GRAFT_REPO_ROOT/gpurun_out/pcie_prof.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pcie_prof | head
