#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03sqt
rm -rf $O; mkdir -p $O
for lib in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/$lib -- python scripts/bench_with_lib.py build/ab/$lib.so --steps 2 --warmup 0 --ramp-seconds 0 --no-tenth --no-cpu-baseline --no-dropin-timing > $O/$lib.log 2>&1 || echo "$lib failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/${lib}_b -- python scripts/bench_with_lib.py build/ab/$lib.so --steps 2 --warmup 0 --ramp-seconds 0 --no-tenth --no-cpu-baseline --no-dropin-timing > $O/${lib}_b.log 2>&1 || echo "$lib failed"
done
python - "$@" <<'PY'
import csv, glob, sys, collections
for lib in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for d in (lib, lib + "_b"):
        for f in glob.glob(f"gpurun_out/r03sqt/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_thermo" not in r["Kernel_Name"]: continue
                a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    w = acc["SQ_WAVES"][1]
    print(lib, "us", round(acc["SQ_WAVES"][2] / max(acc["SQ_WAVES"][0], 1), 1), "VALU/wave", round(acc["SQ_INSTS_VALU"][1] / w), "lanes", round(acc["SQ_THREAD_CYCLES_VALU"][1] / acc["SQ_ACTIVE_INST_VALU"][1], 1),
          "wave_cycles/wave", round(acc["SQ_WAVE_CYCLES"][1] / w), "active_valu/wave", round(acc["SQ_ACTIVE_INST_VALU"][1] / w), "wait_any/wave", round(acc["SQ_WAIT_ANY"][1] / w), "wait_inst/wave", round(acc["SQ_WAIT_INST_ANY"][1] / w), "vmem_rd/wave", round(acc["SQ_INSTS_VMEM_RD"][1] / w, 1))
PY
