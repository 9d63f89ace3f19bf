#!/bin/bash
# rocprofv3 kernel statistics of the reference's whole model with the four drop-in modules, gx1 size, 6 steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/model_kernels
rm -rf $O; mkdir -p $O
RD=$(python - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from oracle import driver
rd = tempfile.mkdtemp(prefix="cice_prof_")
driver.write_rundir(rd, grid="rect", npt=6, istep0=19)
print(rd)
PY
)
cd $RD
ulimit -s unlimited
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $GRAFT_REPO_ROOT/oracle/_ref/cice_dropin_gx1 > $O/model.log 2>&1
grep -A16 "Timing information" $O/model.log | head -20
find $O -name "*kernel_trace.csv" -size +20M -delete
head -30 $O/stats/*/*_kernel_stats.csv | cut -c1-160
