"""Host-side pieces of bench.py that the reported numbers depend on (no GPU)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_launch_count_matches_the_pairing_rule():
    # one launch per subcycle without pairing
    assert bench.launches_per_step(120, False, 0) == (120, 1)
    # pairs; an odd count ends with a single
    assert bench.launches_per_step(120, True, 0) == (60, 2)
    assert bench.launches_per_step(7, True, 0)[0] == 4
    assert bench.launches_per_step(1, True, 0) == (1, 1)
    # K subcycles per sweep: the rest as pairs / singles
    assert bench.launches_per_step(240, True, 0, 4) == (60, 4)
    assert bench.launches_per_step(13, True, 0, 4) == (4, 4)      # 4 + 4 + 4 + 1
    assert bench.launches_per_step(120, True, 6, 4) == (40, 4)    # refresh every 6: 4 + 2 between refreshes
    # wide-halo slabs: refresh after every `overlap`-th subcycle; pairs never straddle one
    for ndte in (120, 240, 7):
        for h in (2, 4, 6, 12):
            n, k, refreshes = 0, 1, 0
            while k <= ndte:                       # independent restatement of Evp::launch_range
                if k + 1 <= ndte and k % h != 0:
                    assert (k + 1) % h == 0 or k % h != 0
                    k += 2
                else:
                    k += 1
                n += 1
            assert bench.launches_per_step(ndte, True, h)[0] == n
    assert bench.launches_per_step(120, True, 12)[0] == 60   # even overlap: all pairs


@pytest.mark.parametrize("nxg,rows", [(320, 48), (320, 96), (320, 192), (3600, 300), (3600, 1200), (100, 29), (100, 8)])
def test_auto_overlap_is_even_and_fits(nxg, rows):
    h = bench.auto_overlap(nxg, rows)
    assert h >= 2 and h % 2 == 0 and h <= rows


def test_auto_overlap_keeps_sweeps_between_the_refreshes():
    """slabs that the library sweeps (K = 4 subcycles per launch) get an overlap that is a multiple of
    K -- every launch between two refreshes is then a sweep, none falls back to the pair kernel."""
    for nxg, nyg, ndte in ((3600, 2400, 240), (1440, 1080, 240)):
        for world in (2, 4, 8):
            rows = nyg // world
            h = bench.auto_overlap(nxg, rows)
            if nxg * (rows + 2 * h) >= bench.SKEW_MIN_CELLS:
                assert h % bench.SKEW_K == 0 and h >= bench.SKEW_K, (nxg, rows, h)
                n, main = bench.launches_per_step(ndte, True, h, bench.SKEW_K)
                assert main == bench.SKEW_K and n == ndte // bench.SKEW_K, (nxg, rows, h, n, main)
    assert bench.auto_overlap(3600, 300) == 8      # the 8-rank slab of the 0.1-degree grid
    # small slabs keep the pair rule
    assert bench.auto_overlap(320, 48) % 2 == 0


def test_workload_names():
    assert bench.workload("gx1")[:3] == (320, 384, 120)
    assert bench.workload("tenth")[:3] == (3600, 2400, 240)
    assert bench.workload("320x72")[:3] == (320, 72, 120)
    assert bench.workload("64x40x8")[:3] == (64, 40, 8)
    with pytest.raises(SystemExit):
        bench.workload("nonsense")


def test_archived_counter_passes_are_found_for_the_kernels_the_bench_reports():
    """roofline.traffic / not_hbm_bound come from archived rocprofv3 passes looked up by kernel name: a renamed kernel
    or a reshaped workgroup would silently turn them into null.  The names the library uses
    today must be in profiles/."""
    for wl, k in (("gx1", "k_evp_resident<4, false"), ("gx1", "k_evp_resident<11, false"),
                  ("gx1", "k_subcycle2<13, false, false, true>"), ("tenth", "k_subcycle2<16, false, false, true>"),
                  ("tenth", "k_subcycle_skew<4, false, false"),
                  ("gx1", "k_thermo_dense<true>"), ("tenth", "k_thermo_dense<true>")):
        traffic, src = bench.pmc_traffic(wl, k)
        assert traffic and traffic > 1e6 and src.startswith("archived PMC pass profiles/"), (wl, k)
    for k in ("k_evp_resident<4, false", "k_evp_resident<11, false"):
        sq = bench.pmc_counters("gx1", k)
        assert sq and 500 < sq["valu_per_wave_subcycle"] < 700, k
    sq = bench.pmc_counters("tenth", "k_subcycle_skew<4, false, false")
    assert sq and 500 < sq["valu_per_wave_subcycle"] and sq["valu_insts_per_launch"] > 1e8 and sq.get("commit")


def test_archived_inkernel_clock_is_found():
    """the issue fractions are also reported at the clock the kernels really hold (in-kernel stamps)."""
    for wl in ("gx1", "tenth"):
        ghz, src = bench.inkernel_clock(wl)
        assert ghz and 1.2 < ghz < 2.6 and "inkernel_clock" in src, wl
