#!/bin/bash
# bench.py --gpus 2 end to end on a ONE-GPU box: both rank processes on device 0, joined by the shared-memory link instead of RCCL
set -o pipefail
mkdir -p gpurun_out
export CICE4_AMD_BENCH_DEVICE=0 CICE4_AMD_BENCH_LINK=shm
( time timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --tenth-steps 1 > gpurun_out/two_procs_slabs.json 2> gpurun_out/two_procs_slabs.err ); echo "slabs rc=$?"
tail -3 gpurun_out/two_procs_slabs.err
python -c "
import json;d=json.load(open('gpurun_out/two_procs_slabs.json'))
print('N=2 slabs: value',round(d['value'],1),'ranks_seen',d['ranks_seen'],d['config']['decomposition'],'|',d['config']['tile'][:60]); t=d.get('tenth')
print('  tenth:', round(t['value'],1), t['config']['tile'][:70]) if t else None"
( time timeout -k 10 300 python bench.py --gpus 2 --peer-loop --no-tenth --steps 5 --warmup 1 > gpurun_out/two_procs_peer.json 2> gpurun_out/two_procs_peer.err ); echo "peer rc=$?"
tail -3 gpurun_out/two_procs_peer.err
python -c "
import json;d=json.load(open('gpurun_out/two_procs_peer.json'))
print('N=2 peer loop: value',round(d['value'],1),'us/subcycle',round(1e6/d['value'],2),'ranks_seen',d['ranks_seen'],d['config']['decomposition'][:120],'|',d['config']['tile'][:80])"
