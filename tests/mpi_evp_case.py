"""One rank of `mpiexec -n P python tests/mpi_evp_case.py <cfg> <P> <processor_shape>` on a box with ONE GPU:
the reference's MPI build (mpi/ modules, MPICH) with OUR ice_dyn_evp and boundary modules; the P tasks share device 0 and
exchange through the shared-memory link (CICE4_AMD_LINK=shm) instead of RCCL.  `call evp(dt)` on every task's own blocks
must reproduce the single-domain checker bit for bit.  With one full-width slab per task (cfg gx3s2) the drop-in connects
the neighbours' exchange buffers through IPC handles sent over MPI and the subcycling runs as ONE launch per task with
device-initiated exchange (argument `loop`: that it did is checked).  Prints 'MPI-EVP-OK <rank> <blocks>'."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CICE4_AMD_LINK"] = "shm"
if len(sys.argv) > 2:
    os.environ["CICE4_AMD_PEER_SHARE"] = sys.argv[2]      # the tasks share one device

import numpy as np  # noqa: E402

from __graft_entry__ import REF_CONFIGS  # noqa: E402
from cice4_amd import lib, synth  # noqa: E402
from oracle import oracle, refapi  # noqa: E402

DT, NDTE = 3600.0, 120
OUT = ("uvel", "vvel", "strength", "divu", "shear", "strocnxT", "strocnyT", "strintx", "strinty", "fm") + synth.SIG_NAMES


def main():
    cfg, nprocs, shape = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    ref = refapi.Ref(cfg, kind="dropinmpi")
    wd = tempfile.mkdtemp()
    os.makedirs(wd, exist_ok=True)
    with open(os.path.join(wd, "ice_in"), "w") as f:
        f.write("&domain_nml\n  nprocs = %d\n  processor_shape = '%s'\n  distribution_type = 'cartesian'\n"
                "  distribution_wght = 'latitude'\n  ew_boundary_type = 'cyclic'\n  ns_boundary_type = 'open'\n/\n"
                % (nprocs, shape))
    os.chdir(wd)
    import ctypes as C
    ref.lib.ref_init_domain.restype = C.c_int
    nb = ref.lib.ref_init_domain(C.c_int(0), b"\0", b"\0", C.c_double(DT), C.c_int(NDTE), C.c_int(0))
    ref.nblocks = nb
    gids = [ref.block_info(k + 1)["block_id"] - 1 for k in range(nb)]
    # the whole domain on one rank: synthetic fields and the checker's answer
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=0)
    grid1 = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05), dom1)
    s1 = synth.evp_state(grid1, dom1, cover="patchy")
    orc = oracle.Oracle()
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    so = {k: v.copy() for k, v in s1.items()}
    orc.evp(orc.make_domain(dom1, grid1), so)
    ny, nx = dom1["ny"], dom1["nx"]
    nball = dom1["nblocks"]

    def mine(a, per=1):      # my blocks of a (nblocks_total*per, ny, nx) array, padded to max_blocks
        a = a.reshape(nball, per, ny, nx)
        out = np.zeros((mxb, per, ny, nx), a.dtype)
        for l, g in enumerate(gids):
            out[l] = a[g]
        return out.reshape(mxb * per, ny, nx) if per > 1 else out.reshape(mxb, ny, nx)

    for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear", "uarear", "tinyarea", "fcor",
              "HTN", "HTE"):
        ref.set(k, mine(grid1[k]))
    ref.set("tmask", mine(grid1["tmask"].astype(float))); ref.set("umask", mine(grid1["umask"].astype(float)))
    ref.set_strength_parameters(1, 0, 0, 4.0)
    ref.evp_gpu_setup()
    if len(sys.argv) > 4 and sys.argv[4] == "badsize" and int(os.environ.get("PMI_RANK", "0")) == nprocs - 1:
        # ONE task is handed a size the library was not built for: its cice_gpu_check has to end the JOB (MPI_ABORT, as the
        # reference's abort_ice does, mpi/ice_exit.F90:78); the other tasks are on their way into evp's first exchange
        ref.lib.ref_gpu_bad_size()
        raise SystemExit("cice_gpu_check returned after a failed library call")
    for k in ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx", "ss_tlty", "uvel", "vvel", "fm",
              "strtltx", "strtlty", "strocnx", "strocny", "strintx", "strinty") + synth.SIG_NAMES:
        ref.set(k, mine(s1[k]))
    ref.set("iceumask", mine(s1["iceumask"].astype(float)))
    ncat = s1["aicen"].size // (nball * ny * nx)
    ref.set("aicen", mine(s1["aicen"].reshape(-1, ny, nx), ncat)); ref.set("vicen", mine(s1["vicen"].reshape(-1, ny, nx), ncat))
    ref.set_evp_parameters(DT, NDTE, False)
    ref.evp(DT)
    if len(sys.argv) > 4 and sys.argv[4] == "loop":
        assert ref.evp_info("resident_peer") == 1, "the cross-task one-launch loop was not used (or fell back)"
    for k in OUT:
        got = ref.get(k)
        for l, g in enumerate(gids):
            if not np.array_equal(got[l], so[k][g]):
                bad = np.argwhere(got[l] != so[k][g])
                raise AssertionError((k, "local block", l, "global", g, len(bad), bad[:5].tolist()))
    assert np.abs(so["uvel"]).max() > 0.01
    # one write per task: the tasks share the pipe, and the pieces of a multi-argument print can interleave
    sys.stdout.write("MPI-EVP-OK %d %s\n" % (int(os.environ.get("PMI_RANK", "-1")), gids))
    sys.stdout.flush()
    ref.lib.ref_end_run()


if __name__ == "__main__":
    main()
