#!/bin/bash
# bench.py on a folded grid: one GPU (gx1 size: fold inside the one-launch loop; 0.1 degree size: sweeps + band), and two
# processes on the one GPU (wide-halo slabs, the fold on the upper rank; shared-memory link instead of RCCL)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in tripole tripoleT; do
  timeout -k 10 400 python bench.py --north $n --no-thermo > gpurun_out/north_$n.json 2> gpurun_out/north_$n.err || { echo "$n failed"; tail -5 gpurun_out/north_$n.err; }
  python -c "
import json,sys;d=json.load(open('gpurun_out/north_$n.json'));print('$n gx1',round(d['value']),d['config'].get('north_boundary'));t=d['tenth'];print('$n tenth',round(t['value'],1),t['config'].get('north_boundary'), t['roofline']['us_per_launch'])"
done
CICE4_AMD_BENCH_LINK=shm CICE4_AMD_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --north tripole --workload tenth --steps 2 --warmup 1 --no-thermo > gpurun_out/north_2.json 2> gpurun_out/north_2.err || { echo "2 ranks failed"; tail -8 gpurun_out/north_2.err; }
python -c "
import json;d=json.load(open('gpurun_out/north_2.json'));print('2 ranks tenth',round(d['value'],1),d['config']['decomposition'],'|',d['config'].get('north_boundary'))"
