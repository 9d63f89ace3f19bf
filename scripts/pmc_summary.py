#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of the
same bench.py command) into HBM bytes per launch per kernel, as MI355X_MICROARCH.md prescribes for
gfx950: FETCH_SIZE is in KiB-like units of 1 KB and undercounts 8-byte-lane reads by 2x (re-verified
by the k_diag_copy8 calibration stream in the same run: 268.4 MB read must be reported).

usage: pmc_summary.py <workload> <fetch_dir> <write_dir> <active_T_cells> [subcycles_per_launch]  >> profiles/*.csv
"""
import csv
import glob
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    wl, fdir, wdir, ncell = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    per = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    F, Wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    w = csv.writer(sys.stdout)
    for k in sorted(F):
        if not ("k_subcycle" in k or "k_diag_copy8" in k or "k_thermo_dense" in k):
            continue
        n, fs, us = F[k]
        nw, ws, _ = Wr.get(k, (0, 0.0, 0.0))
        if n == 0 or nw == 0:
            continue
        fetch_kb, write_kb = fs / n, ws / nw
        read_mb = 2.0 * fetch_kb * 1024 / 1e6
        write_mb = write_kb * 1024 / 1e6
        tot = read_mb + write_mb
        cell = f"{tot * 1e6 / (ncell * (per if 'k_subcycle2' in k else 1.0)):.1f}" if "k_subcycle" in k else ""
        w.writerow([wl, k, n, f"{us / n:.2f}", f"{fetch_kb:.1f}", f"{read_mb:.2f}", f"{write_kb:.1f}", f"{write_mb:.2f}",
                    f"{tot:.2f}", cell])


if __name__ == "__main__":
    main()
