#!/bin/bash
# kernel statistics of the tripole paths (one-launch loop with the fold inside at gx1 size; sweeps + band at 0.1 degree size)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03tri
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gx1 -- python scripts/tripole_rate.py 320 384 > $O/gx1.log 2>&1 || echo "gx1 failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tenth -- python scripts/tripole_rate.py 3600 2400 > $O/tenth.log 2>&1 || echo "tenth failed"
grep -a "us per" $O/gx1.log $O/tenth.log
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
ls -R $O | head -20
