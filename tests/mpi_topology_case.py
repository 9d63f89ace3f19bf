"""One rank of `mpiexec -n P python tests/mpi_topology_case.py <cfg> <P> <processor_shape>`:
the reference's MPI build (mpi/ modules, MPICH) with OUR boundary module decides the block
distribution; ice_HaloCreate rebuilds it on the product side (cice_domain_create with the model's
process grid) and aborts if any local block differs.  No GPU: CICE4_AMD_SKIP_COMM keeps RCCL out.
Prints 'TOPO-OK <rank> <nprocsX>x<nprocsY> <global ids of the local blocks>'."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CICE4_AMD_SKIP_COMM"] = "1"

import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402

from __graft_entry__ import REF_CONFIGS  # noqa: E402
from cice4_amd import lib  # noqa: E402
from oracle import refapi  # noqa: E402


def main():
    cfg, nprocs, shape = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg, kind="dropinmpi")
    wd = tempfile.mkdtemp()
    with open(os.path.join(wd, "ice_in"), "w") as f:
        f.write("&domain_nml\n  nprocs = %d\n  processor_shape = '%s'\n  distribution_type = 'cartesian'\n"
                "  distribution_wght = 'latitude'\n  ew_boundary_type = 'cyclic'\n  ns_boundary_type = 'open'\n/\n"
                % (nprocs, shape))
    os.chdir(wd)
    info = np.zeros(4, np.int32)
    ref.lib.ref_init_topology.restype = C.c_int
    nb = ref.lib.ref_init_topology(info.ctypes.data_as(C.c_void_p))
    rank, npx, npy = int(info[0]), int(info[1]), int(info[2])
    gids = [ref.block_info(k + 1)["block_id"] - 1 for k in range(nb)]
    # the same map from the product's host logic, independently of the Fortran check
    dom = lib.Context().domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0, rank=rank, npx=npx, npy=npy)
    assert [int(g) for g in dom["gid"]] == gids, (rank, dom["gid"], gids)
    print(f"TOPO-OK {rank} {npx}x{npy} {gids}\n", end="", flush=True)   # one write: the tasks share the pipe
    ref.lib.ref_end_run()


if __name__ == "__main__":
    main()
