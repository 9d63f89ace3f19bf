#!/bin/bash
# soaks of the one-launch loop across ranks (contexts of one process on one GPU) and of the K-level sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for q in 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python scripts/soak_peer.py 2 1000 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee gpurun_out/soak_peer2_q$q.log
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python scripts/soak_peer.py 3 500 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee gpurun_out/soak_peer3_q$q.log
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python scripts/soak_peer.py 3 300 384 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee gpurun_out/soak_peer3b_q$q.log
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python scripts/soak_peer.py 4 300 256 2>&1 | grep -a "SOAK\|gave up" | cut -c1-300 | tee gpurun_out/soak_peer4_q$q.log
done
timeout -k 10 600 python -m pytest tests/test_gpu_evp.py -m gpu -q -k "ranks_in_one_process or whole_loop or resident" 2>&1 | grep -a "passed\|failed" | tail -2
