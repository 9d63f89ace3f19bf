"""The counter passes under profiles/ that bench.py quotes (HBM traffic, VALU counts, in-kernel clock) must describe the kernels
of THIS tree: every round-5 CSV carries `source_sha`, the hash of the kernel sources it was taken from
(bench.kernel_source_sha: evp.hip, therm.hip and the headers they compile with); it has to equal the hash of the files in the
tree.  A kernel edit without a new profile run fails here (scripts/gpu_profiles_r05.sh + scripts/profiles_r05.py re-take
them); bench.py reports the same comparison as `counters_stale` in its line (there is no git on the GPU box)."""
import csv
import glob
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")

NEWEST = ("r05_pmc_hbm_traffic.csv", "r05_sq_counters.csv", "r05_inkernel_clock.csv")


def test_counter_passes_were_taken_from_these_kernel_sources():
    sha = bench.kernel_source_sha()
    for name in NEWEST:
        path = os.path.join(ROOT, "profiles", name)
        assert os.path.exists(path), f"{name} missing: run scripts/gpu_profiles_r05.sh and scripts/profiles_r05.py"
        rows = list(csv.DictReader(open(path)))
        assert rows, name
        for r in rows:
            assert r.get("source_sha") == sha, (name, r.get("kernel"), r.get("commit"), "taken from other kernel sources: re-take the profiles")


def test_one_kernel_statistics_file_per_workload():
    """the dominant kernel's average must be the headline workload's: no file mixes full-cover and polar-cap launches"""
    for wl in ("gx1", "tenth_full_cover", "tenth_polar_caps"):
        path = os.path.join(ROOT, "profiles", f"r05_kernel_stats_{wl}.csv")
        assert os.path.exists(path), path
        head = open(path).readline()
        assert "ONE workload" in head and bench.kernel_source_sha() in head, (wl, head)


def test_bench_finds_the_newest_passes_and_calls_them_fresh():
    bench.STALE.clear()
    rw = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_sq_counters.csv"))):
        for r in csv.DictReader(open(path)):
            if r["workload"] == "gx1" and "k_evp_resident" in r["kernel"]:
                rw = r["kernel"].split("k_evp_resident<")[1].split(",")[0]
    assert rw, "no pass of the one-launch loop at gx1"
    t, src = bench.pmc_traffic("gx1", f"k_evp_resident<{rw}, false")
    sq = bench.pmc_counters("gx1", f"k_evp_resident<{rw}, false")
    ghz, _ = bench.inkernel_clock("gx1")
    assert t and sq and ghz and "r05_" in src and "r05_" in sq["source"]
    assert not bench.STALE, bench.STALE


def test_archived_bench_line_is_of_these_kernel_sources():
    """profiles/r05_bench_gx1.json -- the line DESIGN.md and README.md quote -- was printed by these kernels, with fresh counters"""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_gx1.json")))
    assert d["counters"]["kernel_source_sha"] == bench.kernel_source_sha(), "archive a new bench line (scripts/gpu_r5_32.sh)"
    assert d["counters_stale"] is False and d["counters"]["stale_sources"] == []
    assert d["metric"] == "EVP subcycles/sec" and d["n_gpus"] == 1 and d["config"]["nx_global"] == 320 and d["config"]["ndte"] == 120
    for k in ("roofline", "cpu_baseline", "tenth", "thermo", "pcie_inclusive"):
        assert k in d, k
