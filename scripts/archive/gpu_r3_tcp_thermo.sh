#!/bin/bash
# address-translation and memory-latency counters of the column kernel (one library of build/ab per argument)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03tcp
rm -rf $O; mkdir -p $O
A="--steps 2 --warmup 0 --ramp-seconds 0 --no-tenth --no-cpu-baseline --no-dropin-timing"
for lib in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $O/${lib}_a -- python scripts/bench_with_lib.py build/ab/$lib.so $A > $O/${lib}_a.log 2>&1 || echo "$lib a failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/${lib}_b -- python scripts/bench_with_lib.py build/ab/$lib.so $A > $O/${lib}_b.log 2>&1 || echo "$lib b failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $O/${lib}_c -- python scripts/bench_with_lib.py build/ab/$lib.so $A > $O/${lib}_c.log 2>&1 || echo "$lib c failed"
done
python - "$@" <<'PY'
import csv, glob, sys, collections
for lib in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for d in "abc":
        for f in glob.glob(f"gpurun_out/r03tcp/{lib}_{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_thermo_dense" not in r["Kernel_Name"]: continue
                a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for k, a in sorted(acc.items()):
        print(lib, k, "launches", a[0], "per launch %.4g" % (a[1] / max(a[0], 1)), "us %.1f" % (a[2] / max(a[0], 1)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
