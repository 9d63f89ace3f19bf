#!/bin/bash
# round 4: the sweep in front of a wide-halo refresh as edge + interior launches -- cost on ONE GPU, two slabs of the 0.1-degree
# grid's 4- and 8-rank size on one rank, refresh through pack / RCCL (1-rank communicator) / unpack: split on / off, interleaved
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 4 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_split.txt
for rep in 1 2 3; do
  for wl in 3600x600x240 3600x1200x240; do
    for split in 1 0; do
      CICE4_AMD_SELF_COMM=1 CICE4_AMD_SKEW_SPLIT=$split timeout -k 10 300 python bench.py --workload $wl --slabs 2 --overlap 8 $B > gpurun_out/split_one.json 2> gpurun_out/split_one.err || { echo "$wl split=$split FAILED" | tee -a gpurun_out/r4_split.txt; tail -3 gpurun_out/split_one.err; continue; }
      echo "rep$rep $wl two slabs on one rank, H = 8, split=$split: $(python -c "import json;d=json.load(open('gpurun_out/split_one.json'));print(round(1e6/d['value'],2), 'us per subcycle,', d['config']['launches_per_step'], 'launches per step')")" | tee -a gpurun_out/r4_split.txt
    done
  done
done
