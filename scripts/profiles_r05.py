#!/usr/bin/env python3
"""gpurun_out/r05prof/ (scripts/gpu_profiles_r05.sh) -> the round-5 summaries under profiles/.  Every CSV carries `commit` and
`source_sha` (bench.kernel_source_sha of the tree the profiles were taken from): bench.py and tests/test_profiles.py compare it with
the tree they run in.
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (re-verified by the k_diag_copy8 calibration stream of
the same run: 268.4 MB read must come out); counters are per launch (sum over the chip / launches)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r05prof")
P = os.path.join(ROOT, "profiles")
commit = open(os.path.join(O, "commit.txt")).read().strip()
sha = open(os.path.join(O, "source_sha.txt")).read().strip()
WANT = ("k_evp_resident", "k_subcycle", "k_thermo", "k_diag_copy8")


def fresh(files):
    """gpurun merges a run's files into what earlier runs left in gpurun_out/: only the files of the newest run count."""
    if not files:
        return files
    newest = max(os.path.getmtime(f) for f in files)
    return [f for f in files if os.path.getmtime(f) > newest - 1800]


def collect(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    for f in fresh(glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def short(k):
    return k.replace("void cice::(anonymous namespace)::", "").replace("cice::(anonymous namespace)::", "").replace("void cice::", "").replace("cice::", "")


# ---- HBM traffic
ACT = {"gx1": 121980.0, "tenth": 8627996.0}
rows = []
for wl in ("gx1", "tenth"):
    F, W = collect(f"pmc_FETCH_SIZE_{wl}"), collect(f"pmc_WRITE_SIZE_{wl}")
    for k in sorted(F):
        if not any(w in k for w in WANT) or "FETCH_SIZE" not in F[k] or "WRITE_SIZE" not in W.get(k, {}):
            continue
        n, fs, us = F[k]["FETCH_SIZE"]; nw, ws, _ = W[k]["WRITE_SIZE"]
        read_mb, write_mb = 2.0 * fs / n * 1024 / 1e6, ws / nw * 1024 / 1e6
        sub = 120 if "k_evp_resident" in k else (int(k.split("k_subcycle_skew<")[1].split(",")[0]) if "k_subcycle_skew" in k else
                                                 2 if "k_subcycle2" in k else 1 if "k_subcycle" in k else "")
        per = f"{(read_mb + write_mb) * 1e6 / (ACT[wl] * sub):.1f}" if sub else ""
        rows.append([wl, short(k), n, f"{us / n:.2f}", f"{fs / n:.1f}", f"{read_mb:.2f}", f"{ws / nw:.1f}", f"{write_mb:.2f}",
                     f"{read_mb + write_mb:.2f}", sub, per, commit, sha])
with open(os.path.join(P, "r05_pmc_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["workload", "kernel", "launches", "avg_duration_us(under PMC)", "FETCH_SIZE_KB_per_launch(raw)",
                "read_MB_per_launch(=2xFETCH_SIZE, factor re-verified on k_diag_copy8)", "WRITE_SIZE_KB_per_launch",
                "write_MB_per_launch", "total_MB_per_launch", "subcycles_per_launch", "bytes_per_active_T_cell_per_subcycle", "commit", "source_sha"])
    w.writerows(rows)
# ---- SQ counters
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_THREAD_CYCLES_VALU"]
rows = []
for wl in ("gx1", "tenth"):
    A = collect(f"sq1_{wl}"); Bc = collect(f"sq2_{wl}")
    for k in sorted(A):
        if not any(w in k for w in WANT[:3]):
            continue
        d = dict(A[k]); d.update(Bc.get(k, {}))
        if "SQ_WAVES" not in d:
            continue
        n = d["SQ_WAVES"][0]
        v = {c: d[c][1] for c in names if c in d}
        sub = 120 if "k_evp_resident" in k else (int(k.split("k_subcycle_skew<")[1].split(",")[0]) if "k_subcycle_skew" in k else
                                                 2 if "k_subcycle2" in k else 1)
        per_wave = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_ACTIVE_INST_VALU", 1), 1) if "SQ_THREAD_CYCLES_VALU" in v else ""
        rows.append([wl, short(k), n, sub] + [f"{v.get(c, 0):.4e}" for c in names] +
                    [f"{per_wave:.0f}", f"{per_wave / sub:.1f}", f"{lanes:.1f}" if lanes != "" else "", f"{d['SQ_WAVES'][2] / n:.1f}", commit, sha])
with open(os.path.join(P, "r05_sq_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["workload", "kernel", "launches", "subcycles_per_launch"] + names +
               ["VALU_instr_per_wave", "VALU_instr_per_wave_per_subcycle", "thread_cycles_per_active_inst_cycle(=avg active lanes)",
                "avg_duration_us(under PMC)", "commit", "source_sha"])
    w.writerows(rows)
# ---- kernel statistics: one file per workload
for src, dst in (("stats_gx1", "r05_kernel_stats_gx1.csv"), ("stats_gx1_thermo", "r05_kernel_stats_gx1_with_thermo.csv"),
                 ("stats_tenth_full", "r05_kernel_stats_tenth_full_cover.csv"), ("stats_tenth_caps", "r05_kernel_stats_tenth_polar_caps.csv")):
    fs = fresh(glob.glob(os.path.join(O, src, "**", "*kernel_stats.csv"), recursive=True))
    if fs:
        with open(fs[0]) as fi, open(os.path.join(P, dst), "w") as fo:
            fo.write(f"# rocprofv3 --kernel-trace --stats of ONE workload ({src}), commit {commit}, source_sha {sha}\n")
            fo.write(fi.read())
if os.path.exists(os.path.join(O, "bench_gx1.json")):
    shutil.copy(os.path.join(O, "bench_gx1.json"), os.path.join(P, "r05_bench_gx1.json"))
print(open(os.path.join(P, "r05_pmc_hbm_traffic.csv")).read())
print(open(os.path.join(P, "r05_sq_counters.csv")).read())
