#!/bin/bash
# round 4, second call: new tests (MPI abort, fine-grained peer buffers, smoke), fine- vs coarse-grained exchange buffers, sweep-kernel A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r4_smoke.txt &&
timeout -k 10 900 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_evp.py -x -q -k "mpi_job or failed_library_call or rank_processes or ranks_in_one_process or wide_halo" 2>&1 | tail -5 | tee gpurun_out/r4_tests2.txt &&
for rep in 1 2; do
  timeout -k 10 200 python scripts/peer_two_slabs.py 2 2>&1 | grep "peer loop" | sed "s/^/fine   rep$rep /" | tee -a gpurun_out/r4_peer_mem.txt
  CICE4_AMD_PEER_COARSE=1 timeout -k 10 200 python scripts/peer_two_slabs.py 2 2>&1 | grep "peer loop" | sed "s/^/coarse rep$rep /" | tee -a gpurun_out/r4_peer_mem.txt
done &&
bash scripts/gpu_r4_ab.sh base trim prio hoist tpass all3
