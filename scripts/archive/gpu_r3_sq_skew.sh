#!/bin/bash
# SQ counter passes for the sweep kernel at 0.1 degree (separate --pmc runs, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03sq
mkdir -p $O
B="--workload tenth --steps 1 --warmup 0 --ramp-seconds 0 --no-thermo --no-cpu-baseline --no-dropin-timing $SKEW_EXTRA"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python bench.py $B > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for f in glob.glob("gpurun_out/r03sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_subcycle" not in k: continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, d in acc.items():
    print(k)
    for c, (n, v, us) in sorted(d.items()):
        print(f"  {c:28s} launches {n:4d} per-launch {v / n:14.1f}  us/launch {us / n:9.1f}")
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
