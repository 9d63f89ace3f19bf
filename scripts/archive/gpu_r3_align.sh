#!/bin/bash
for wl in gx1 318x384 382x510 384x510 ; do
  timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-tenth --no-cpu-baseline --no-dropin-timing > gpurun_out/al.json 2> gpurun_out/al.err || { echo "$wl FAILED"; tail -3 gpurun_out/al.err; continue; }
  echo "$wl $(python -c "import json;d=json.load(open('gpurun_out/al.json'))['thermo'];print('G/s',round(d['value']/1e9,3),'ms',round(d['ms_per_pass'],4), d['updates_per_pass'])")"
done
