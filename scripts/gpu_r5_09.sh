#!/bin/bash
# Round 5, call 9: TIMING EXPERIMENT (wrong results): the free-running loop if the interior wavefronts had no east-west hand-off to wait for
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_09.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 200 python scripts/bench_with_lib.py build/ab/lib_fake.so --no-thermo --no-tenth --no-cpu-baseline --no-dropin-timing $extra > gpurun_out/r5_09.json 2>gpurun_out/r5_09.err || { tail -20 gpurun_out/r5_09.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5_09.json')); print('gx1', ' '.join(sys.argv[1:]), ':', round(d['value']), 'subcycles/s =', round(1e6/d['value'],3), 'us per subcycle')" "$extra" "$@" | tee -a gpurun_out/r5_09.txt
}
run "" CICE4_AMD_RESIDENT_FAKE_EW=0
for p in 0 2 4; do
  run "--resident-prio $p" CICE4_AMD_RESIDENT_FAKE_EW=1
  run "--resident-prio $p --resident-waves 12" CICE4_AMD_RESIDENT_FAKE_EW=1
done
run "--resident-prio 1" CICE4_AMD_RESIDENT_FAKE_EW=1
run "--resident-prio 4" CICE4_AMD_RESIDENT_FAKE_EW=1 CICE4_AMD_RESIDENT_POLL_DELAY=0
