#!/bin/bash
# HBM traffic of the batched thermo kernel (separate FETCH_SIZE / WRITE_SIZE passes, calibration stream included)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing"
for wl in gx1 tenth; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/thermo_pmc_${c}_$wl -- python bench.py --workload $wl --steps 1 --warmup 0 $B --calibrate > gpurun_out/thermo_pmc_${c}_$wl.log 2>&1
  done
done
echo thermo-pmc-done
