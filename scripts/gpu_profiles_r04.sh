#!/bin/bash
# Round-4 evidence for profiles/ (taken at the commit named in gpurun_out/r04prof/commit.txt): the default bench line, rocprofv3
# kernel stats of the same command, separate PMC passes (FETCH_SIZE / WRITE_SIZE with the 8-byte-lane calibration stream;
# SQ counters) for gx1 and 0.1 degree, and the kernel statistics of the reference's whole model with the drop-in modules.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04prof
rm -rf $O; mkdir -p $O
cp commit.txt $O/commit.txt 2>/dev/null || echo unknown > $O/commit.txt
B="--no-cpu-baseline --no-dropin-timing"
timeout -k 10 500 python bench.py > $O/bench_gx1.json 2> $O/bench_gx1.err || echo "bench failed"
echo bench-done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python bench.py --steps 5 --warmup 1 $B > $O/stats_default.log 2>&1 || echo "stats failed"
echo stats-done
for wl in gx1 tenth; do
  X="--no-tenth"; [ $wl = tenth ] && X="--workload tenth"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$wl -- python bench.py $X --steps 1 --warmup 0 --ramp-seconds 0 $B --calibrate > $O/pmc_${c}_$wl.log 2>&1 || echo "pmc $c $wl failed"
  done
  echo pmc-$wl-done
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq${i}_$wl -- python bench.py $X --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/sq${i}_$wl.log 2>&1 || echo "sq$i $wl failed"
  done
  echo sq-$wl-done
done
# the whole model with the four drop-in modules, gx1 size, 6 steps: transport, boundary-layer, merge, halo kernels
RD=$(python - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from oracle import driver
rd = tempfile.mkdtemp(prefix="cice_prof_")
driver.write_rundir(rd, grid="rect", npt=6, istep0=19)
print(rd)
PY
)
( cd $RD && ulimit -s unlimited && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/model -- $GRAFT_REPO_ROOT/oracle/_ref/cice_dropin_gx1 > $GRAFT_REPO_ROOT/$O/model.log 2>&1 ) || echo "model stats failed"
echo model-done
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
