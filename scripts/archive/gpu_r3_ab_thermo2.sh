#!/bin/bash
# thermo rate of build/ab libraries, each optionally with the column sort switched on: name[@chunk,group]
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%@*}; opt=""; [ "$lib" != "$spec" ] && opt=${spec#*@}
    if [ -n "$opt" ]; then export CICE4_AMD_THERMO_SORT=${opt%,*} CICE4_AMD_THERMO_GROUP=${opt#*,}; else unset CICE4_AMD_THERMO_SORT CICE4_AMD_THERMO_GROUP; fi
    timeout -k 10 300 python scripts/bench_with_lib.py build/ab/$lib.so --steps 8 --warmup 2 --no-tenth --no-cpu-baseline --no-dropin-timing > gpurun_out/abt.json 2> gpurun_out/abt.err || { echo "$spec FAILED"; tail -3 gpurun_out/abt.err; continue; }
    echo "rep$rep $spec $(python -c "import json;d=json.load(open('gpurun_out/abt.json'))['thermo'];print('G/s',round(d['value']/1e9,3),'ms',round(d['ms_per_pass'],4))")"
  done
done
