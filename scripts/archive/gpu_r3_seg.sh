#!/bin/bash
# step time of the sweep kernel against the number of workgroups per CU (rows per workgroup): args K seg...
set -o pipefail
mkdir -p gpurun_out
K=$1; shift
for seg in "$@"; do
  timeout -k 10 300 python bench.py --workload tenth --steps 3 --warmup 1 --no-thermo --no-cpu-baseline --no-dropin-timing --skew-levels $K --skew-seg-rows $seg $SKEW_EXTRA > gpurun_out/seg.json 2> gpurun_out/seg.err || { echo "seg $seg FAILED"; tail -3 gpurun_out/seg.err; continue; }
  echo "K=$K seg=$seg $(python -c "
import json;d=json.load(open('gpurun_out/seg.json'));us=d['roofline']['us_per_launch'];K=$K;seg=$seg
import math
nseg=math.ceil(2400/seg); strips=math.ceil(3601/(62-2*K)); steps=seg+2*K-1+2*(K-1)
print('value',round(d['value'],1),'us/launch',round(us,1),'WGs',nseg*strips,'steps',steps,'us/step',round(us/steps,2))")"
done
