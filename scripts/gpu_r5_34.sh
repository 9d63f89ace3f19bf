#!/bin/bash
# Round 5, call 34: `bench.py --gpus 4` as FOUR RANK PROCESSES on the one GPU (the box admits six processes on a card; shared-memory
# link, torch.distributed over gloo for the barrier): gx1 with wide-halo slabs, gx1 with the cross-rank one-launch loop verified
# against the message path, 0.1 degree with sweeps.  Interior ranks 1 and 2 have two neighbours.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export CICE4_AMD_BENCH_DEVICE=0 CICE4_AMD_BENCH_LINK=shm GPU_MAX_HW_QUEUES=8
timeout -k 10 400 python bench.py --gpus 4 --no-thermo --no-tenth --no-peer-try > gpurun_out/r5_34_gx1.json 2> gpurun_out/r5_34_gx1.err; echo "gx1 slabs rc=$?"
tail -c 600 gpurun_out/r5_34_gx1.json | cut -c1-600; echo
timeout -k 10 400 python bench.py --gpus 4 --no-thermo --no-tenth --peer-loop --peer-verify --no-peer-try > gpurun_out/r5_34_gx1_peer.json 2> gpurun_out/r5_34_gx1_peer.err; echo "gx1 peer rc=$?"
tail -c 400 gpurun_out/r5_34_gx1_peer.json; echo; grep -i "verif\|bit" gpurun_out/r5_34_gx1_peer.err | tail -5
timeout -k 10 600 python bench.py --gpus 4 --workload tenth --no-thermo --no-peer-try --steps 2 --warmup 1 > gpurun_out/r5_34_tenth.json 2> gpurun_out/r5_34_tenth.err; echo "tenth rc=$?"
tail -c 400 gpurun_out/r5_34_tenth.json; echo
grep -v amdgpu.ids gpurun_out/r5_34_gx1.err | tail -5; grep -v amdgpu.ids gpurun_out/r5_34_tenth.err | tail -5
