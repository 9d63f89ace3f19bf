#!/bin/bash
# end-of-round evidence: the whole GPU suite, the PCIe break-down, then the profiles at this commit
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q -s > gpurun_out/all_gpu_tests.log 2>&1 || { grep -a -v "^ " gpurun_out/all_gpu_tests.log | tail -40; exit 1; }
grep -a "passed\|failed" gpurun_out/all_gpu_tests.log | tail -2
timeout -k 10 300 python scripts/pcie_evp.py 20 2>&1 | grep -a "PCIe\|upload" | tee gpurun_out/pcie_evp.log
bash scripts/gpu_profiles_r03.sh
