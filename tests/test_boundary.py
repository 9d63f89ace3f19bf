"""Boundary module (ice_HaloCreate / ice_HaloUpdate / ice_HaloExtrapolate) against the compiled
reference, one process per boundary combination (tests/boundary_case.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Rectangular grid.  'closed' edges cannot be initialised with any grid the reference ships
# (rectgrid aborts for closed E-W, ice_grid.F90:1124; the gx3 land mask has ocean on the northern
# edge, ice_domain.F90:313-347); for ghost copies the reference treats 'closed' exactly like 'open'
# (no neighbour: ice_blocks.F90:455-458,482-485,504-507,522-525), which is what 'open' covers here.
CASES = [("gx3b4", "cyclic", "open"), ("pad", "cyclic", "open"), ("pad", "open", "open"),
         ("pad", "open", "cyclic"), ("pad", "cyclic", "cyclic"), ("padx", "cyclic", "open"),   # padx: max_blocks > blocks
         # tripole (U-fold) north boundary, serial/ice_boundary.F90:705-869: every field location x kind x type
         ("small", "cyclic", "tripole"), ("pad", "cyclic", "tripole"),
         # the fold through T points (serial/ice_boundary.F90:725-776: three buffer rows, other offsets and symmetric pairs)
         # (2 x 2 blocks, and 10 x 12 blocks with a padded last column and six rows in the top block row; the 'pad'
         #  configurations have two rows there: the reference itself stops, "not enough points in block for tripole")
         ("small", "cyclic", "tripoleT"), ("gx3b4", "cyclic", "tripoleT"),
         # land-block elimination on the reference's own gx3 grid and land mask: 10 x 12 blocks, the 4 all-land ones dropped by
         # create_distribution; ghost cells facing them take the fill value (mpi/ice_boundary.F90:5108-5111)
         ("gx3e", "cyclic", "open", "gx3"),
         # ... and the same block distribution under either fold
         ("gx3e", "cyclic", "tripole", "gx3"), ("gx3e", "cyclic", "tripoleT", "gx3")]


def run_case(mode, cfg, ew, ns, *extra):
    from oracle import refapi
    need = [("ref", cfg)] + ([("dropin", cfg)] if mode == "gpu" else [])
    for kind, c in need:
        if not refapi.available(c, kind):
            pytest.skip(f"oracle/_ref/libcice_{kind}_{c}.so not built")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "boundary_case.py"), mode, cfg, ew, ns, *extra],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "BOUNDARY-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(c))
def test_halo_lists_equal_reference_update(case):
    """Host topology (cice4_amd/csrc/domain.cpp) == serial/ice_boundary.F90 for padded blocks and
    every boundary type the path supports."""
    run_case("lists", *case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(c))
def test_boundary_module_dropin(case):
    """The reference's callers (ice_domain, ice_grid, ice_state) linked with our ice_boundary
    module, ghost cells filled on the MI355X: same grid, same updates, bit for bit."""
    run_case("gpu", *case)
