"""Multi-rank block topology of the MPI build: the reference's own distribution (mpi/ modules under
mpiexec) against the product's cice_domain_create, through OUR boundary module's ice_HaloCreate.
CPU only (tests/mpi_topology_case.py keeps RCCL out)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


CASES = [("gx3b4", 2, "slenderX1"), ("gx3b4", 2, "slenderX2"), ("gx3b4", 4, "square-ice"), ("pad", 2, "slenderX1"),
         ("pad", 3, "slenderX1"), ("pad", 4, "square-ice"), ("pad", 2, "slenderX2")]


@pytest.mark.parametrize("cfg,nprocs,shape", CASES)
def test_multi_rank_halo_lists_equal_the_reference_mpi_exchange(cfg, nprocs, shape):
    """The pure reference (MPI build) really exchanges ghost cells among the ranks; the product's per-rank
    copy and message lists fill every ghost cell with the same value (tests/mpi_halo_case.py)."""
    from oracle import refapi
    if not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec")
    if not refapi.available(cfg, "refmpi"):
        pytest.skip(f"oracle/_ref/libcice_refmpi_{cfg}.so not built")
    p = subprocess.run([MPIEXEC, "-n", str(nprocs), sys.executable, os.path.join(ROOT, "tests", "mpi_halo_case.py"),
                        cfg, str(nprocs), shape], capture_output=True, text=True, timeout=300, cwd="/tmp")
    ok = [l for l in p.stdout.splitlines() if l.startswith("HALO-OK")]
    assert p.returncode == 0 and len(ok) == nprocs, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("cfg,nprocs,shape", [("gx3b4", 2, "slenderX1"), ("gx3b4", 2, "slenderX2"),
                                               ("gx3b4", 4, "square-ice"), ("pad", 2, "slenderX1"),
                                               ("pad", 3, "slenderX1"), ("pad", 4, "square-ice"),
                                               ("pad", 2, "slenderX2")])
def test_block_to_task_map_matches_the_reference(cfg, nprocs, shape):
    from oracle import refapi
    if not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec")
    if not refapi.available(cfg, "dropinmpi"):
        pytest.skip(f"oracle/_ref/libcice_dropinmpi_{cfg}.so not built")
    p = subprocess.run([MPIEXEC, "-n", str(nprocs), sys.executable, os.path.join(ROOT, "tests", "mpi_topology_case.py"),
                        cfg, str(nprocs), shape], capture_output=True, text=True, timeout=300, cwd="/tmp")
    ok = [l for l in p.stdout.splitlines() if l.startswith("TOPO-OK")]
    assert p.returncode == 0 and len(ok) == nprocs, p.stdout[-2000:] + p.stderr[-2000:]
    owned = sorted(int(g) for l in ok for g in l.split("[")[1].rstrip("]").replace(",", " ").split())
    nblocks = {"gx3b4": 4, "pad": 9}[cfg]
    assert owned == list(range(nblocks)), ok      # every block exactly once
