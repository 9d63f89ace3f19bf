#!/bin/bash
# kernel times of the sorted column update: how much is the sort, how much the permuted column kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03sort
rm -rf $O; mkdir -p $O
A="--steps 4 --warmup 1 --ramp-seconds 0 --no-tenth --no-cpu-baseline --no-dropin-timing"
for opt in "$@"; do
  export CICE4_AMD_THERMO_SORT=${opt%,*} CICE4_AMD_THERMO_GROUP=${opt#*,}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$opt -- python scripts/bench_with_lib.py build/ab/lib_slots.so $A > $O/s_$opt.log 2>&1 || echo "$opt failed"
  echo "== $opt"; grep -h "k_thermo" $O/s_$opt/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
