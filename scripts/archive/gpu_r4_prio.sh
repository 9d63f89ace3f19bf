#!/bin/bash
# round 4: issue priority among the three workgroups of a CU in the one-launch loop (gx1, dense shape)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 20 --warmup 3 --no-thermo --no-cpu-baseline --no-dropin-timing --no-tenth"
: > gpurun_out/r4_prio.txt
for rep in 1 2 3; do
  for wl in gx1 gx3 250x200; do
    for prio in 0 1 2; do
      timeout -k 10 200 python bench.py --workload $wl --resident-prio $prio $B > gpurun_out/prio_one.json 2> gpurun_out/prio_one.err || { echo "$wl prio=$prio FAILED" | tee -a gpurun_out/r4_prio.txt; tail -3 gpurun_out/prio_one.err; continue; }
      echo "rep$rep $wl prio=$prio: $(python -c "import json;d=json.load(open('gpurun_out/prio_one.json'));print(round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle;', d['config']['tile'][-60:])")" | tee -a gpurun_out/r4_prio.txt
    done
  done
done
