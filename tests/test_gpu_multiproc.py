"""Rank PROCESSES on one GPU: what a multi-GPU job does, with every rank on device 0 and a shared-memory link in place of
RCCL (which refuses two ranks on one device).  Covers what the in-process test (test_ranks_in_one_process) cannot: the
neighbours' exchange buffers mapped through IPC handles (hipIpcGetMemHandle / hipIpcOpenMemHandle), separate address spaces,
and bench.py --gpus N end to end from the plain command."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DT, NDTE = 3600.0, 120
NXG, NYG = 96, 72


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


RANK_CODE = r'''
import os, sys, pickle
import numpy as np
ROOT, mode, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.is_available()
from cice4_amd import lib, synth
DT, NDTE, NXG, NYG = 3600.0, 120, 96, 72
c = lib.Context(device=0); c.sync()
if mode == "slabs":
    dom = c.domain_create_slabs(NXG, NYG, world, ew=1, ns=0, rank=rank, nranks=world, overlap=4)
else:
    dom = c.domain_create(NXG, NYG, NXG, NYG // world, ew=1, ns=0, rank=rank, npx=1, npy=world)
c.comm_init_shm("/cice4_amd_test_%s" % os.environ["MASTER_PORT"], rank, world, 8 << 20)
gg = synth.global_grid(NXG, NYG, perturb=0.15, land_frac=0.05, seed=31)
grid = synth.block_fields(gg, dom)
s = synth.evp_state(grid, dom, seed=31, cover="patchy")
c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
if mode == "peer":
    c.evp_set_option("resident_peer_share", world)
    every = [None] * world
    dist.all_gather_object(every, c.evp_peer_export_ipc())
    if rank > 0: c.evp_peer_connect_ipc(0, every[rank - 1])
    if rank < world - 1: c.evp_peer_connect_ipc(1, every[rank + 1])
    assert c.evp_get_info("resident_peer") == 1
    assert c.evp_get_info("resident_peer_fine") == 1     # IPC-mapped exchange copies / progress words: fine-grained memory
    dist.barrier()
else:
    c.evp_set_option("resident", 0)
c.evp(DT, s)
fell_back = mode == "peer" and c.evp_get_info("resident_peer") != 1
r0, r1 = int(dom["own_jlo"][0]) - 1, int(dom["own_jhi"][0])
j0 = int(dom["j0"][0] + dom["own_jlo"][0] - dom["jlo"][0])
keys = ("uvel", "vvel", "divu", "shear", "strength", "strintx", "prs_sig", "stressp_1", "stress12_4")
pickle.dump(dict(j0=j0, fell_back=fell_back, **{k: s[k][0, r0:r1, 1:-1] for k in keys}), open(os.path.join(outdir, "r%d.pkl" % rank), "wb"))
dist.barrier()                       # nobody unmaps buffers a neighbour may still be writing to
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,world", [("classic", 2), ("slabs", 2), ("peer", 2), ("peer", 3)])
def test_rank_processes_on_one_gpu_reproduce_one_domain(tmp_path, mode, world):
    import pickle
    sys.path.insert(0, ROOT)
    from cice4_amd import lib, synth
    from oracle import oracle
    orc = oracle.Oracle()
    c1 = lib.Context()
    dom1 = c1.domain_create(NXG, NYG, NXG, NYG, ew=1, ns=0)
    gg = synth.global_grid(NXG, NYG, perturb=0.15, land_frac=0.05, seed=31)
    grid1 = synth.block_fields(gg, dom1)
    s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
    orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
    orc.evp(orc.make_domain(dom1, grid1), s1)
    orc.set_strength_parameters()
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", RANK_CODE, ROOT, mode, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    parts = sorted((pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)), key=lambda d: d["j0"])
    assert not any(p["fell_back"] for p in parts), "the cross-rank loop timed out and fell back"
    for k in ("uvel", "vvel", "divu", "shear", "strength", "strintx", "prs_sig", "stressp_1", "stress12_4"):
        got = np.concatenate([p[k] for p in parts], axis=0)
        assert np.array_equal(got, s1[k][0, 1:-1, 1:-1]), (mode, world, k)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra", [[], ["--peer-loop"]])
def test_bench_two_ranks_from_the_plain_command(extra):
    """`python bench.py --gpus 2` (no torchrun): the bench starts its two rank processes itself; with
    CICE4_AMD_BENCH_DEVICE=0 / CICE4_AMD_BENCH_LINK=shm both sit on the one GPU and exchange through the shared-memory
    link.  One JSON line from rank 0, ranks_seen 2."""
    env = dict(os.environ, CICE4_AMD_BENCH_DEVICE="0", CICE4_AMD_BENCH_LINK="shm")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "gx3", "--steps", "3",
                        "--warmup", "1", "--no-tenth", "--no-cpu-baseline"] + extra, capture_output=True, text=True,
                       timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    if extra:
        assert "cross-rank one-launch loop" in d["config"]["decomposition"], d["config"]


@pytest.mark.timeout(900)
def test_bench_tries_and_verifies_the_cross_rank_loop_by_itself():
    """`python bench.py --gpus 2` at gx1: after the main measurement (wide-halo slabs) every rank starts a child process;
    the children form their own job, run one evp(dt) through the per-subcycle exchange and one through the cross-rank
    one-launch loop, compare bit for bit on every rank, and only then time the loop.  The line carries both decompositions;
    `value` is the faster one."""
    env = dict(os.environ, CICE4_AMD_BENCH_DEVICE="0", CICE4_AMD_BENCH_LINK="shm")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-tenth", "--no-cpu-baseline", "--no-thermo"], capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["value"] > 0
    other = d["other_decomposition"]
    used_loop = "peer_loop_verified" in d["config"]
    assert used_loop or other.get("config", {}).get("peer_loop_verified"), (d["config"], other)
    assert d["value"] >= other["value"] > 0
    print("two rank processes on one GPU: value %.0f (%s), other decomposition %.0f" %
          (d["value"], "cross-rank loop" if used_loop else "wide-halo slabs", other["value"]))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cfg,nprocs,shape", [("gx3b4", 2, "slenderX1"), ("gx3b4", 2, "slenderX2"), ("gx3b4", 4, "square-ice"),
                                              ("gx3s2", 2, "slenderX1")])
def test_mpi_job_of_the_fortran_dropin_on_one_gpu(cfg, nprocs, shape):
    """`mpiexec -n P`: the reference's MPI build with our ice_dyn_evp and boundary modules, P tasks on the one GPU joined by
    the shared-memory link (CICE4_AMD_LINK=shm; RCCL would refuse them): block distribution by the reference's own
    create_distribution, ghost cells between tasks after every subcycle, `call evp(dt)` = the single-domain checker.
    cfg gx3s2: one full-width slab per task -- the drop-in hands the IPC handles of the exchange buffers to the neighbouring
    task over MPI and the whole subcycling is one launch per task (checked: the loop was used and did not fall back)."""
    import shutil
    from oracle import refapi
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    if not refapi.available(cfg, "dropinmpi"):
        pytest.skip(f"oracle/_ref/libcice_dropinmpi_{cfg}.so not built")
    p = subprocess.run([mpiexec, "-n", str(nprocs), sys.executable, os.path.join(ROOT, "tests", "mpi_evp_case.py"), cfg,
                        str(nprocs), shape] + (["loop"] if cfg == "gx3s2" else []), capture_output=True, text=True, timeout=500,
                       cwd="/tmp")
    # counted, not matched line by line: the tasks write to one pipe and two of their lines can run together
    assert p.returncode == 0 and p.stdout.count("MPI-EVP-OK") == nprocs, p.stdout[-2500:] + p.stderr[-2500:]


@pytest.mark.timeout(300)
def test_a_failed_library_call_on_one_task_ends_the_mpi_job():
    """mpi/ice_exit.F90:41-80: the reference aborts through abort_ice -> MPI_ABORT.  One task of an `mpiexec -n 2` job of the
    drop-in build is handed a bad size (ncat + 1): cice_gpu_check prints the library's message and calls MPI_ABORT, so the job
    ends non-zero within seconds -- an `error stop` of that task alone would leave the other one waiting in evp's first
    exchange until the link's time-out."""
    import shutil
    import time
    from oracle import refapi
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    if not refapi.available("gx3b4", "dropinmpi"):
        pytest.skip("oracle/_ref/libcice_dropinmpi_gx3b4.so not built")
    t0 = time.time()
    p = subprocess.run([mpiexec, "-n", "2", sys.executable, os.path.join(ROOT, "tests", "mpi_evp_case.py"), "gx3b4", "2",
                        "slenderX2", "badsize"], capture_output=True, text=True, timeout=250, cwd="/tmp")
    took = time.time() - t0
    assert p.returncode != 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "MPI-EVP-OK" not in p.stdout
    assert "aborting the MPI job" in p.stderr and "ncat" in (p.stdout + p.stderr), p.stdout[-2000:] + p.stderr[-2000:]
    assert took < 90.0, f"the job took {took:.0f} s to end: a task was left waiting"
