#!/usr/bin/env python3
"""Benchmark of the MI355X-native CICE4 hot path (contract: see DESIGN.md section "Measurement").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gx1|tenth|gx3]

One "step" = one pass of the EVP hot loop over the whole grid = ndte subcycles of
stress + stepu + halo update (source/ice_dyn_evp.F90:347-404 of the reference) on
device-resident state.  `value` = EVP subcycles per second for the whole job.  The same
JSON line carries the column-thermodynamics rate ((cell,category) updates per second),
the HBM roofline of the dominant kernel and a CPU baseline timed on this host.

N > 1: one rank per GPU; the grid is cut into N j-slabs (strong scaling) and ghost rows travel by RCCL
point-to-point inside the library.  Either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment), or -- a plain `python bench.py --gpus N` -- this process starts the N ranks
itself as fresh children BEFORE anything touches a GPU (launch_ranks) and passes rank 0's JSON line through.
Every wait of a rank on its peers (rendezvous, communicator creation, first exchange) is bounded: a rank that
cannot proceed exits non-zero with a message, and the launcher then stops the others.

  --host-only: no GPU.  The same launcher, rendezvous (gloo) and slab decomposition; the ranks exchange a test
  field through the library's message lists over gloo and check every ghost / overlap row (tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cice4_amd import lib, synth  # noqa: E402

WORKLOADS = {
    # name: (nx_global, ny_global, ndte, description)
    "gx3": (100, 116, 120, "gx3-size 100x116 rectangular synthetic grid, full ice cover, ncat=5, ndte=120"),
    "gx1": (320, 384, 120, "gx1-size 320x384 rectangular synthetic grid, full ice cover, ncat=5, ndte=120"),
    "tenth": (3600, 2400, 240, "0.1-degree-size 3600x2400 rectangular synthetic grid, full ice cover, ncat=5, ndte=240"),
}
DT = 3600.0
EVP_BYTES_PER_CELL = 384.0      # compulsory bytes per active cell per subcycle
THERMO_BYTES_PER_COLUMN = 304.0  # per (cell,category) update (+376 B per cell shared by the categories)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_T0 = time.time()


def progress(msg):
    """Stage markers on stderr (the JSON line is the only thing on stdout): a long run must not look hung."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def workload(name):
    if name in WORKLOADS:
        return WORKLOADS[name]
    try:
        parts = [int(x) for x in name.lower().split("x")]
        nxg, nyg = parts[0], parts[1]
        ndte = parts[2] if len(parts) > 2 else 120
    except (ValueError, IndexError):
        raise SystemExit(f"unknown workload {name!r}")
    return (nxg, nyg, ndte, f"diagnostic {nxg}x{nyg} rectangular synthetic grid, full ice cover, ncat=5, ndte={ndte}")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="gx1",
                   help="gx3, gx1, tenth, or NXxNY[xNDTE] (diagnostic: a synthetic grid of that size, e.g. the "
                        "320x72 slab one of 8 ranks works on at gx1)")
    p.add_argument("--waves", type=int, default=0, help="wavefronts per EVP workgroup (4/8/16); 0 = auto")
    p.add_argument("--rows", type=int, default=0, help="T-rows per wavefront (1/2/4/8); 0 = auto")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--north", choices=("open", "tripole", "tripoleT"), default="open",
                   help="north boundary of the synthetic grid.  'tripole' / 'tripoleT': the fold COSIMA's production grids have "
                        "(ice_blocks.F90:228-233), ocean and ice up to it -- not BASELINE.json's configuration (no CPU baseline, "
                        "no drop-in timing, no cross-rank-loop attempt for it)")
    p.add_argument("--overlap", type=int, default=-1,
                   help="N > 1: overlap rows of the wide-halo slabs = subcycles between ghost exchanges "
                        "(-1 = auto: 8, or a quarter of a rank's rows if that is smaller; 0 = exchange every subcycle)")
    p.add_argument("--slabs", type=int, default=0,
                   help="N = 1 only (diagnostic): cut the grid into this many wide-halo slabs on the one GPU; with "
                        "CICE4_AMD_SELF_COMM=1 their ghost refresh goes through pack/RCCL/unpack")
    p.add_argument("--resident-waves", type=int, default=0, help="wavefronts per workgroup of k_evp_resident (0 = library's choice)")
    p.add_argument("--resident-prio", type=int, default=-1, help="issue priority among the workgroups of a CU in k_evp_resident (0, 1, 2); -1 = library's choice")
    p.add_argument("--no-resident", action="store_true", help="do not run the whole subcycle loop in one launch (k_evp_resident)")
    p.add_argument("--no-fuse", action="store_true", help="one subcycle per launch (k_subcycle) even where two are possible")
    p.add_argument("--fused-waves", type=int, default=0, help="wavefronts per workgroup of k_subcycle2 (8/12/13/14/16); 0 = auto")
    p.add_argument("--no-skew", action="store_true", help="no K-subcycle sweeps (k_subcycle_skew) on large grids")
    p.add_argument("--skew-levels", type=int, default=0, help="K of k_subcycle_skew (2, 3, 4, 5, 6, 8); 0 = library's choice")
    p.add_argument("--skew-seg-rows", type=int, default=0, help="rows a workgroup of k_subcycle_skew owns; 0 = auto")
    p.add_argument("--skew-gen-pct", type=int, default=-1, help="longer row segments for the workgroups dispatched first; -1 = library's choice")
    p.add_argument("--skew-balance", type=int, default=-1, help="segments of the sweep follow the measured cost of their rows (0 / 1); -1 = library's choice")
    p.add_argument("--cover", choices=("full", "patchy", "caps"), default="full",
                   help="ice cover of the synthetic state on one GPU (diagnostic; the headline metric is quoted on full cover)")
    p.add_argument("--skew-prio", type=int, default=-1, help="rotate issue priorities among workgroups of a CU; -1 = library's choice")
    p.add_argument("--skew-stagger-ns", type=int, default=-1, help="start delay between workgroups sharing a CU; -1 = library's choice")
    p.add_argument("--skew-split-probe", type=int, default=0,
                   help="measurement aid: cut a one-block grid as an interior slab with this many overlap rows and run every "
                        "sweep as edge + interior launches (results are not meaningful)")
    p.add_argument("--no-derive", action="store_true", help="load the 9 T-cell metrics instead of recomputing them")
    p.add_argument("--calibrate", action="store_true",
                   help="also run the 8-B-per-lane calibration copy (k_diag_copy8, 2 x 256 MiB) for PMC runs")
    p.add_argument("--ramp-seconds", type=float, default=0.3,
                   help="untimed device warm-up before the W warm-up steps (clock ramp)")
    p.add_argument("--no-thermo", action="store_true")
    p.add_argument("--no-tenth", action="store_true",
                   help="gx1 run: skip the short 0.1-degree (3600x2400, ndte=240) sub-record")
    p.add_argument("--tenth-steps", type=int, default=3)
    p.add_argument("--no-caps", action="store_true", help="skip the 0.1-degree run under polar-cap ice cover (diagnostic part of the line)")
    p.add_argument("--thermo-coherence", type=int, default=THERMO_COHERENCE,
                   help="correlation length (cells) of the melting/cold, snow/bare, day/night regions of the synthetic "
                        "thermo columns; 0 = every column drawn independently (white noise)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-dropin-timing", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget")
    p.add_argument("--cpu-baseline-worker", default="", help=argparse.SUPPRESS)
    p.add_argument("--peer-loop", action="store_true",
                   help="N > 1: one classic slab per rank and the cross-rank one-launch loop (device-initiated exchange through "
                        "IPC-mapped buffers of the neighbours) instead of wide-halo slabs; grids that fit the chip only (gx1, gx3)")
    p.add_argument("--peer-verify", action="store_true",
                   help="with --peer-loop: first run one step through the per-subcycle message exchange and one through the "
                        "cross-rank loop from the same state and compare velocities and stresses bit for bit on every rank; "
                        "exit 21 if they differ or the loop fell back (22: fell back during the timed steps)")
    p.add_argument("--no-peer-try", action="store_true",
                   help="N > 1, gx1: do not start the second set of rank processes that tries the cross-rank one-launch loop")
    p.add_argument("--peer-try-timeout", type=float, default=240.0)
    p.add_argument("--host-only", action="store_true",
                   help="no GPU: launcher + rendezvous + slab decomposition + one ghost exchange over gloo, checked")
    p.add_argument("--comm-timeout", type=float, default=120.0,
                   help="N > 1: seconds a rank may wait for its peers in the rendezvous, in the creation of the RCCL "
                        "communicator and in the first exchange before it gives up (exit code 14)")
    p.add_argument("--launch-timeout", type=float, default=2400.0,
                   help="N > 1 without torchrun: seconds the self-started ranks may run before the launcher stops them")
    return p.parse_args()


class bounded:
    """`with bounded(seconds, what):` -- a wait on other ranks that must not hang.  If the block has not finished
    after `seconds`, the process says what it was waiting for and leaves with exit code 14 (the native call it sits
    in -- ncclCommInitRank, a stream synchronise behind an ncclRecv -- cannot be interrupted any other way; the
    launcher, or torchrun, then stops the other ranks)."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self.timer = seconds, what, None

    def _fire(self):
        rank = os.environ.get("RANK", "0")
        sys.stderr.write(f"[bench] rank {rank}: {self.what} did not complete within {self.seconds:.0f} s "
                         f"(a peer is missing, has failed, or shares this rank's device) -- giving up\n")
        sys.stderr.flush()
        os._exit(14)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def visible_gpus():
    """Number of GPUs this host shows, counted in a CHILD process: the launcher itself must never touch a GPU
    (its children are started afterwards), and the count needs the HIP runtime."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from cice4_amd import lib; "
            "print(lib.load().cice_device_count())" % ROOT)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else 0
    except (subprocess.SubprocessError, ValueError, IndexError, OSError):
        return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as fresh child processes -- before this process
    has made any GPU call, and it never makes one -- with the environment torch.distributed.run would give them,
    pass rank 0's stdout (the one JSON line) through, and return the first non-zero exit code.  A rank that fails
    takes the others with it (each child is its own process group; exactly those groups are signalled)."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    if not args.host_only and os.environ.get("CICE4_AMD_BENCH_DEVICE") is None:
        have = visible_gpus()
        if have < n:
            sys.stderr.write(f"[bench] --gpus {n}: this host shows {have} GPU(s).  One rank needs one GPU of its own "
                             f"(RCCL refuses two ranks on one device); nothing was started.\n")
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CICE4_AMD_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True))
    deadline = time.time() + args.launch_timeout
    rc, why = 0, ""
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = bad[0][1], f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            return 0
        if time.time() > deadline:
            rc, why = 15, f"the ranks did not finish within --launch-timeout {args.launch_timeout:.0f} s"
            break
        time.sleep(0.2)
    sys.stderr.write(f"[bench] {why}; stopping the other ranks\n")
    for sig in (signal.SIGTERM, signal.SIGKILL):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)          # the rank and the CPU-baseline child it may have started
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.time() + 5.0
        while time.time() < t_end and any(p.poll() is None for p in procs):
            time.sleep(0.1)
    return rc if rc > 0 else 1


def try_peer_loop(args, rank, world, dist):
    """N > 1, gx1: after the main measurement every rank starts ONE child process; the children form their own job (own
    rendezvous port, own RCCL communicator) and run `bench.py --peer-loop --peer-verify`: the cross-rank one-launch loop
    with device-initiated exchange, which has to reproduce the per-subcycle exchange bit for bit in that run before it is
    timed.  Whatever happens to the children -- no peer mapping between the devices, a time-out, a mismatch -- stays in
    the children: the parents wait a bounded time, stop them, and keep their own result.  Returns rank 0's record of the
    child job (None on the other ranks and whenever any child did not finish cleanly)."""
    import signal
    import socket
    import subprocess
    port = [None]
    if rank == 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port[0] = s.getsockname()[1]
        s.close()
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, MASTER_PORT=str(port[0]), CICE4_AMD_BENCH_PEER_CHILD="1")
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        del env[k]      # (under torchrun the parents use the agent's store; the children bring their own on their own port)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--peer-loop", "--peer-verify", "--no-peer-try",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--ramp-seconds", str(args.ramp_seconds), "--no-tenth",
           "--no-thermo", "--no-cpu-baseline", "--no-dropin-timing", "--comm-timeout", str(min(args.comm_timeout, 60.0))]
    progress("gx1: trying the cross-rank one-launch loop in a second set of rank processes")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, start_new_session=True)
    out = b""
    try:
        out, _ = p.communicate(timeout=args.peer_try_timeout)
    except subprocess.TimeoutExpired:
        pass
    if p.poll() is None:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass
            try:
                p.wait(5.0)
                break
            except subprocess.TimeoutExpired:
                continue
    rcs = [None] * world
    with bounded(args.comm_timeout, "collecting the exit codes of the cross-rank-loop attempt"):
        dist.all_gather_object(rcs, p.returncode)
    if any(c != 0 for c in rcs):
        progress(f"gx1: the cross-rank-loop attempt is not used (exit codes {rcs}: 21 = not bit-identical / fell back, "
                 f"22 = fell back while timed, 14 = a wait on a peer ran out)")
        return {"used": False, "exit_codes": rcs} if rank == 0 else None
    if rank != 0:
        return None
    try:
        rec = json.loads(out.decode().strip().splitlines()[-1])
    except Exception:       # noqa: BLE001
        return {"used": False, "exit_codes": rcs, "why": "no JSON line from the child job"}
    if not rec.get("config", {}).get("peer_loop_verified"):
        return {"used": False, "exit_codes": rcs, "why": "the child job did not verify the loop"}
    return {"used": None, "value": rec["value"], "ms_per_step": rec["ms_per_step"], "timing": rec.get("timing"),
            "config": rec["config"], "roofline": rec.get("roofline")}


def init_dist(n_gpus, comm_timeout=120.0):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {n_gpus}, or without RANK / WORLD_SIZE in the environment")
    dist = None
    hook = os.environ.get("CICE4_AMD_BENCH_TEST_HOOK", "")      # tests/test_bench_launcher.py: "die:R" / "hang:R"
    if hook and world > 1 and hook.split(":")[1] == str(rank):
        if hook.startswith("die"):
            raise SystemExit(7)
        time.sleep(3600)
    if world > 1:
        import datetime
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barriers, the ncclUniqueId, max-over-ranks) on gloo; the data path
        # (ghost rows) is RCCL inside libcice4_amd.so
        with bounded(comm_timeout + 10.0, "rendezvous of the ranks (gloo)"):
            dist.init_process_group("gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=max(5.0, comm_timeout)))
    return rank, world, local, dist


def host_only(args):
    """--host-only: what a multi-rank run does before and around the device work, without a device: rendezvous, the
    slab decomposition bench.py --gpus N uses, and ONE ghost exchange of a test field (value = global cell number)
    through the library's own send / receive lists over gloo, in the order Halo::update works (wrap list, pack,
    messages, on-rank refresh, unpack).  Every cell of every rank's slab -- owned rows, overlap rows, ghost rows and
    columns -- must then hold the number of the global cell it mirrors."""
    rank, world, local, dist = init_dist(args.gpus, args.comm_timeout)
    import torch
    ctx = lib.Context()                       # host-side domain logic only
    nxg, nyg, ndte, _ = workload(args.workload)
    rows = nyg // world
    overlap = args.overlap if args.overlap >= 0 else auto_overlap(nxg, rows)
    dom = ctx.domain_create_slabs(nxg, nyg, world, ew=1, ns=0, rank=rank, nranks=world, overlap=overlap)
    ny, nx = dom["ny"], dom["nx"]
    jlo, jhi, j0 = int(dom["jlo"][0]), int(dom["jhi"][0]), int(dom["j0"][0])
    own_lo, own_hi = int(dom["own_jlo"][0]), int(dom["own_jhi"][0])
    jg = j0 + (np.arange(ny) + 1 - jlo)                         # global row of local row j (1-based j = index + 1)
    ig = (np.arange(nx) - 1) % nxg                              # cyclic east-west
    want = (jg[:, None] * nxg + ig[None, :]).astype(np.float64)
    f = np.full((ny, nx), -1.0)
    f[own_lo - 1:own_hi, 1:nx - 1] = want[own_lo - 1:own_hi, 1:nx - 1]   # only what this rank owns
    flat = f.reshape(-1)
    sends, recvs = ctx.halo_msgs(0), ctx.halo_msgs(1)
    with bounded(args.comm_timeout, "first ghost exchange (gloo)"):
        flat[dom["hdst"]] = flat[dom["hsrc"]]
        reqs, bufs = [], []
        for peer, addr in recvs:
            b = torch.empty(len(addr), dtype=torch.float64)
            bufs.append((addr, b))
            reqs.append(dist.irecv(b, src=peer))
        for peer, addr in sends:
            reqs.append(dist.isend(torch.from_numpy(flat[addr].copy()), dst=peer))
        for r in reqs:
            r.wait()
        if len(dom["rsrc"]):
            flat[dom["rdst"]] = flat[dom["rsrc"]]
        for addr, b in bufs:
            flat[addr] = b.numpy()
    # rows inside the global grid (the outermost ranks' rows beyond an open edge are never written)
    inside = (jg >= 0) & (jg < nyg)
    inside[:max(0, jlo - 2)] = False                            # padding below the extended slab, if any
    inside[jhi + 1:] = False
    ok = bool(np.array_equal(f[inside], want[inside]))
    res = [ok]
    if dist is not None:
        t = torch.tensor([1 if ok else 0, 1], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        res = [int(t[0]) == world]
        seen = int(t[1])
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"host_only": True, "n_gpus": world, "ranks_seen": seen, "exchange_ok": res[0],
                          "config": {"workload": workload(args.workload)[3],
                                     "decomposition": f"1x{world} j-slabs, {overlap} overlap rows",
                                     "messages_per_rank": {"send": len(sends), "recv": len(recvs)}}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if res[0] else 1


SKEW_MIN_CELLS = 600000   # the library's own threshold for K-subcycle sweeps (Evp::skew_min_cells)
SKEW_K = 4                # ... and its K (Evp::skew_levels)


def auto_overlap(nxg, rows, skew_k=SKEW_K):
    """Rows of overlap H = subcycles between two ghost refreshes of a wide-halo slab of `rows` owned rows.

    Slabs large enough for K-subcycle SWEEPS (k_subcycle_skew; the library uses them from SKEW_MIN_CELLS cells): H is a
    MULTIPLE OF K, or part of every refresh interval would fall back to the pair kernel (round 3: H = 6 with K = 4 sent a
    third of the subcycles through it).  Cost per subcycle of one rank, measured at HEAD on one MI355X with grids of the
    slab's size (scripts/gpu_r4_slabs.sh, profiles/r04_slab_costs_*.txt; 3600 columns, K = 4):
        kernel    9.5 + 0.1163 x (rows + 2 H) us      (300 rows: H = 4 / 8 / 12 -> 44.0 / 44.9 / 46.0 us)
        exchange  (t_fix + (H + 1) x nxg x 14 x 8 B / link) / H:  one packed message per neighbour, u, v and 12 stresses of
                  H + 1 rows; t_fix ~ 20 us (pack + RCCL p2p + unpack, measured through the one-GPU message path), link ~ 60 GB/s
                  (one xGMI link, estimated -- nothing here has run between two devices)
    The byte term hardly depends on H ((H + 1) / H), the fixed term falls like 1 / H, the redundant rows cost 0.23 us per
    unit of H: H = 8 and 12 are within 0.5 % of each other at every slab height, H = 4 is 4 % worse.  -> H = 2 K.
    Smaller slabs (pairs of subcycles per launch): H even; latency-bound slabs take many rows, they cost nothing there."""
    if nxg * (rows + 4 * skew_k) >= SKEW_MIN_CELLS and rows >= 8 * skew_k:
        return 2 * skew_k
    if nxg * rows <= 200 * 200:
        # latency-bound slabs (measured: 320x72 5.05 us, 320x120 5.28 us per subcycle): a row of overlap
        # costs ~0.005 us per subcycle, an exchange ~10 us -> the optimum is flat around 24-32 rows
        h = max(2, min(24, rows // 2))
    else:
        # pair kernel (k_subcycle2), round-1 calibration: 50e-6 us per cell and subcycle, 12 us per exchange
        t_kernel = nxg * rows * 50e-6
        h = max(2, min(rows // 4, int(round((12.0 * rows / (2.0 * t_kernel)) ** 0.5))))
    return h + (h & 1) if h + (h & 1) <= rows else max(2, h - (h & 1))


def launches_per_step(ndte, fused, overlap, skew_k=0):
    """Kernel launches of one step's subcycle loop and the subcycles the most common launch runs
    (Evp::launch_range): K subcycles per sweep or pairs where the domain allows, never across a wide-halo refresh."""
    n, k, sizes = 0, 1, {}
    def clear(length):
        if k + length - 1 > ndte:
            return False
        return not (overlap > 0 and any(q % overlap == 0 for q in range(k, k + length - 1)))
    while k <= ndte:
        if skew_k and clear(skew_k):
            step = skew_k
        elif fused and clear(2):
            step = 2
        else:
            step = 1
        sizes[step] = sizes.get(step, 0) + 1
        k += step
        n += 1
    return n, max(sizes, key=lambda z: sizes[z] * z)


NORTH = {"open": 0, "tripole": 3, "tripoleT": 4}   # Boundary codes of include/cice4_amd.h


def build_case(ctx, wl, rank, world, overlap=-1, slabs=0, peer_loop=False, north="open"):
    nxg, nyg, ndte, _ = workload(wl)
    ns = NORTH[north]
    if nyg % world:
        raise SystemExit(f"ny_global={nyg} not divisible by {world} ranks")
    if world > 1 and peer_loop:
        # one classic slab per rank (ghost rows owned by the neighbours): the decomposition of the cross-rank one-launch loop
        dom = ctx.domain_create(nxg, nyg, nxg, nyg // world, ew=1, ns=ns, rank=rank, npx=1, npy=world)
        dom["overlap"] = 0
        gg = synth.global_grid(nxg, nyg)
        grid = synth.block_fields(gg, dom)
        return dom, grid, synth.evp_state(grid, dom, cover="full"), ndte
    if world == 1 and slabs > 1:
        rows = nyg // slabs
        if overlap < 0:
            overlap = auto_overlap(nxg, rows)
        dom = ctx.domain_create_slabs(nxg, nyg, slabs, ew=1, ns=ns, rank=0, nranks=1, overlap=overlap)
        if dom["nsend"]:        # CICE4_AMD_SELF_COMM: messages to the own rank through a 1-rank communicator
            ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    elif world == 1:
        dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
        dom["overlap"] = 0
    else:
        rows = nyg // world
        if overlap < 0:
            overlap = auto_overlap(nxg, rows)
        # j-slabs, one per GPU, each extended by `overlap` rows that are recomputed and refreshed
        # only every `overlap` subcycles
        dom = ctx.domain_create_slabs(nxg, nyg, world, ew=1, ns=ns, rank=rank, nranks=world, overlap=overlap)
    # uniform 30 km rectangular grid (ice_grid.F90:976); under a fold no land rows close the domain: ocean and ice up to it
    gg = synth.global_grid(nxg, nyg, land_rows=0) if ns else synth.global_grid(nxg, nyg)
    grid = synth.block_fields(gg, dom, north_ocean=bool(ns))
    state = synth.evp_state(grid, dom, cover=COVER)
    return dom, grid, state, ndte


COVER = "full"          # --cover: the ice cover of the synthetic state ("caps": what a global grid looks like, diagnostic)
THERMO_COHERENCE = 24   # cells; see synth.thermo_columns(coherent=...)


def thermo_case(dom, seed=20261003, coherent=None):
    """Module-array-shaped inputs of the batched thermo step for this rank's blocks: full ice cover,
    'mixed' regime (40 % melting / 60 % cold columns, snow-covered and bare, day and night), organised
    in regions with a correlation length of `coherent` cells (0: drawn independently per cell)."""
    if coherent is None:
        coherent = THERMO_COHERENCE
    nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
    NC, NI, NS = 5, 4, 1
    z = lambda *s: np.zeros(s)
    b = dict(aicen=z(nb, NC, ny, nx), trcrn=z(nb, NC, 5, ny, nx), vicen=z(nb, NC, ny, nx),
             vsnon=z(nb, NC, ny, nx), eicen=z(nb, NC * NI, ny, nx), esnon=z(nb, NC * NS, ny, nx),
             lhcoef=z(nb, NC, ny, nx), shcoef=z(nb, NC, ny, nx), fswsfc=z(nb, NC, ny, nx),
             fswint=z(nb, NC, ny, nx), fswthrun=z(nb, NC, ny, nx), Sswabs=z(nb, NC, NS, ny, nx),
             Iswabs=z(nb, NC, NI, ny, nx), mlt_onset=z(nb, ny, nx), frz_onset=z(nb, ny, nx))
    for k in lib.THERMO_FORCING:
        b[k] = z(nb, ny, nx)
    for k in lib.THERMO_OUT:
        b[k] = z(nb, NC, ny, nx)
    cols = {}
    # Large planes (0.1 degree: 8.6 M cells x 5 categories) are tiled from a 512 x 768 patch of the same statistics
    # instead of drawn cell by cell: the host spent 45 s there per run, with the device idle.
    tile = (ny - 2) * (nx - 2) > 2_000_000
    py, px = (512, 768) if tile else (ny - 2, nx - 2)

    def fill(dst, src):
        """dst[..., ny, nx] <- the interior of src[..., py+2, px+2] repeated over the physical cells (block copies, no
        temporary); the ghost ring stays zero"""
        for y0 in range(1, ny - 1, py):
            h = min(py, ny - 1 - y0)
            for x0 in range(1, nx - 1, px):
                w = min(px, nx - 1 - x0)
                dst[..., y0:y0 + h, x0:x0 + w] = src[..., 1:1 + h, 1:1 + w]

    for ib in range(nb):
        for n in range(NC):
            a, icells, ii, jj = synth.thermo_columns(py + 2, px + 2, n, regime="mixed",
                                                     seed=seed + 31 * int(dom["gid"][ib]), ice_frac=1.0,
                                                     coherent=coherent)
            put = fill if tile else (lambda dst, src: dst.__setitem__(Ellipsis, src))
            cols[(ib, n)] = None if tile else (a, icells, ii, jj)    # (the per-category lists feed the CPU baseline only)
            for k in ("aicen", "vicen", "vsnon", "lhcoef", "shcoef", "fswsfc", "fswint", "fswthrun"):
                put(b[k][ib, n], a[k])
            put(b["trcrn"][ib, n], a["trcrn"])
            put(b["eicen"][ib, n * NI:(n + 1) * NI], a["eicen"])
            put(b["esnon"][ib, n * NS:(n + 1) * NS], a["esnon"])
            put(b["Sswabs"][ib, n], a["Sswabs"])
            put(b["Iswabs"][ib, n], a["Iswabs"])
            if n == 0:
                for k in lib.THERMO_FORCING + ("mlt_onset", "frz_onset"):
                    put(b[k][ib], a[k])
    return b, cols


def host_cores():
    """Threads of the all-cores baseline: the cores this process may run on, at most 16 (the CPU share of one
    GPU on the bench hosts; CICE4_AMD_BENCH_CORES overrides)."""
    if os.environ.get("CICE4_AMD_BENCH_CORES"):
        return max(1, int(os.environ["CICE4_AMD_BENCH_CORES"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))


def cpu_baseline(wl, grid, state, dom, ndte, tcols, budget_s):
    """Time the CPU checker on a bounded sample of the SAME workload, on this host.
    kind 'reference': the reference's own compiled Fortran (oracle/_ref), single thread;
    kind 'port': our plain-C restatement (oracle/), single thread."""
    from oracle import oracle as orc_mod
    out = {}
    orc = orc_mod.Oracle()
    orc.set_evp_parameters(DT, ndte)
    orc.set_strength_parameters()
    ref = None
    try:
        from oracle import refapi
        if refapi.available(wl):
            ref = refapi.Ref(wl)
    except Exception:
        ref = None
    ncpu = os.cpu_count()
    # --- EVP: subcycle loop
    if ref is not None:
        import tempfile
        ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=ndte)
        ref.set_strength_parameters()
        for k in ("dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarea", "uarea", "tarear",
                  "uarear", "tinyarea", "fcor"):
            ref.set(k, grid[k])
        ref.set("tmask", grid["tmask"].astype(float)); ref.set("umask", grid["umask"].astype(float))
        for k in ("aice", "vice", "vsno", "aice0", "strairxT", "strairyT", "uocn", "vocn", "ss_tltx",
                  "ss_tlty", "uvel", "vvel", "fm", "strtltx", "strtlty", "strocnx", "strocny", "strintx",
                  "strinty") + synth.SIG_NAMES:
            ref.set(k, state[k])
        ref.set("iceumask", state["iceumask"].astype(float))
        ny, nx = dom["ny"], dom["nx"]
        ref.set("aicen", state["aicen"].reshape(-1, ny, nx)); ref.set("vicen", state["vicen"].reshape(-1, ny, nx))
        ncalls, t = 0, 0.0
        while t < budget_s * 0.5 and ncalls < 50:
            t0 = time.perf_counter(); ref.evp(DT); t += time.perf_counter() - t0; ncalls += 1
        out["evp"] = dict(value=ndte * ncalls / t, unit="EVP subcycles/s", cores=1, kind="reference",
                          sample=f"{ncalls} call(s) of the reference's evp(dt) ({ndte} subcycles each, incl. its "
                                 f"once-per-step prep/finish ~3%), {wl} full cover, 1 thread of {ncpu} host cores")
    else:
        d = orc.make_domain(dom, grid)
        s = {k: v.copy() for k, v in state.items()}
        nsub = max(4, int(budget_s * 0.5 * 90 * (320 * 384) / (dom["nxg"] * dom["nyg"])))
        t = orc.evp_subcycles_only(d, s, nsub)
        out["evp"] = dict(value=nsub / t, unit="EVP subcycles/s", cores=1, kind="port",
                          sample=f"{nsub} subcycles of the C restatement (stress+stepu+halo), {wl} full cover, "
                                 f"1 thread of {ncpu} host cores")
    # --- thermo: thermo_vertical over the categories of block 0
    if tcols is not None:
        impl = ref if ref is not None else orc
        impl.init_thermo()
        nupd, t = 0, 0.0
        passes = 0
        while t < budget_s * 0.5 and passes < 20:
            for n in range(5):
                a, icells, ii, jj = tcols[(0, n)]
                ac = {k: v.copy() for k, v in a.items()}
                t0 = time.perf_counter()
                impl.thermo_vertical(DT, icells, ii, jj, ac)
                t += time.perf_counter() - t0
                nupd += icells
            passes += 1
        out["thermo"] = dict(value=nupd / t, unit="(cell,category) updates/s", cores=1,
                             kind="reference" if ref is not None else "port",
                             sample=f"{passes} pass(es) of thermo_vertical over 5 categories of block 0 "
                                    f"({nupd} column updates), 1 thread of {ncpu} host cores")
    # --- all host cores available to this process: the C restatement ("port"; bit-identical to the reference,
    # tests/test_oracle_vs_ref.py) with its stress / stepu loops spread over the cores by OpenMP, and
    # thermo_vertical on one slice of the cell list per thread.  A reported baseline, not a target.
    ncores = host_cores()
    if ncores > 1:
        try:
            orc_mp = orc_mod.Oracle(omp=True)
            orc_mp.set_evp_parameters(DT, ndte)
            orc_mp.set_strength_parameters()
            d = orc_mp.make_domain(dom, grid)
            s = {k: v.copy() for k, v in state.items()}
            orc_mp.evp_subcycles_only(d, s, 4)                       # thread start-up, first touch
            nsub = max(8, int(budget_s * 0.25 * 90 * min(ncores, 8) * (320 * 384) / (dom["nxg"] * dom["nyg"])))
            t = orc_mp.evp_subcycles_only(d, s, nsub)
            out["evp_all_cores"] = dict(value=nsub / t, unit="EVP subcycles/s", cores=ncores, kind="port",
                                        sample=f"{nsub} subcycles of the C restatement (stress + stepu + halo), OpenMP over "
                                               f"{ncores} threads (the cores this process may use; host has {ncpu}), {wl} full cover")
        except OSError:
            pass
        if tcols is not None:
            from concurrent.futures import ThreadPoolExecutor
            orc.init_thermo()
            jobs = []
            for n in range(5):
                a, icells, ii, jj = tcols[(0, n)]
                cut = np.linspace(0, icells, ncores + 1).astype(int)
                for c0_, c1_ in zip(cut[:-1], cut[1:]):
                    if c1_ > c0_:     # own arrays per slice: thermo_vertical zeroes its outputs over the whole block
                        li = np.zeros_like(ii); lj = np.zeros_like(jj)
                        li[:c1_ - c0_] = ii[c0_:c1_]; lj[:c1_ - c0_] = jj[c0_:c1_]
                        jobs.append(({k: v.copy() for k, v in a.items()}, int(c1_ - c0_), li, lj))
            nupd = sum(j[1] for j in jobs)
            with ThreadPoolExecutor(ncores) as ex:                   # ctypes releases the GIL during the call
                t0 = time.perf_counter()
                list(ex.map(lambda j: orc.thermo_vertical(DT, j[1], j[2], j[3], j[0]), jobs))
                t = time.perf_counter() - t0
            out["thermo_all_cores"] = dict(value=nupd / t, unit="(cell,category) updates/s", cores=ncores, kind="port",
                                           sample=f"one pass of the C restatement's thermo_vertical over 5 categories of block 0 "
                                                  f"({nupd} column updates), cell lists cut into {ncores} slices, one thread each")
    return out


ARITH_VALU_PER_WAVE_SUBCYCLE = 575.0   # stress + momentum of one row, one subcycle (SQ_INSTS_VALU of the barrier loop / wavefronts / subcycles)
KERNEL_SOURCES = ("evp.hip", "evp.h", "therm.hip", "therm.h", "common.h", "libm_exact.h")


def kernel_source_sha():
    """Hash of the sources the EVP / thermo kernels are compiled from.  The archived counter passes under profiles/ carry the
    hash of the tree they were taken from (scripts/profiles_r05.py); a pass whose hash differs from this tree's describes
    OTHER kernels: the line then says `counters_stale: true` (tests/test_profiles.py holds the committed tree to it)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cice4_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


STALE = []      # archived passes used by this run whose source hash is not this tree's


def _fresh(row, path):
    ok = row.get("source_sha", "") == kernel_source_sha()
    if not ok and path not in STALE:
        STALE.append(path)
    return ok


def pmc_traffic(workload, kernel_substr):
    """HBM bytes per launch of a kernel from an ARCHIVED rocprofv3 PMC pass (profiles/r0N_pmc_hbm_traffic*.csv:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes and re-verified on the k_diag_copy8 calibration stream).  Not measured in this run: the source
    file (newest round first) is named next to the value; None if no pass exists for this kernel variant."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                if row["workload"] == workload and kernel_substr in row["kernel"]:
                    rel = os.path.relpath(path, ROOT)
                    return float(row["total_MB_per_launch"]) * 1e6, ("archived PMC pass " + rel +
                                                                     ("" if _fresh(row, rel) else " (STALE: taken from other kernel sources)"))
        except (OSError, KeyError):
            continue
    return None, None


def inkernel_clock(workload):
    """Shader clock the dominant EVP kernel of `workload` holds under sustained load, from the archived in-kernel stamp run
    (profiles/r*_inkernel_clock.csv: scripts/inkernel_clock.py with the diagnostic build -DCICE4_AMD_STAMPS -- s_memtime /
    s_memrealtime around the loop after 2.5 s of back-to-back launches; the product build holds no stamp).  (GHz, source) or
    (None, None)."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_inkernel_clock*.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                if row["workload"] == workload:
                    rel = os.path.relpath(path, ROOT)
                    return float(row["clock_ghz_median"]), ("archived in-kernel stamps " + rel + " (commit " + row.get("commit", "?") + ")" +
                                                             ("" if _fresh(row, rel) else " (STALE: taken from other kernel sources)"))
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_counters(workload, kernel_substr):
    """VALU instructions per wavefront (and per subcycle) and per launch from an archived SQ counter pass
    (profiles/r*_sq_counters*.csv), newest round first; None if there is none for this kernel."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters*.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                name = row.get("kernel") or row.get("kernel (gx1)") or ""
                if kernel_substr in name and row.get("workload", workload) == workload and row.get("subcycles_per_launch"):
                    per_wave = float(row["VALU_instr_per_wave"])
                    rel = os.path.relpath(path, ROOT)
                    out = {"valu_per_wave_subcycle": per_wave / float(row["subcycles_per_launch"]),
                           "source": "archived SQ pass " + rel + ("" if _fresh(row, rel) else " (STALE: taken from other kernel sources)")}
                    if row.get("SQ_INSTS_VALU") and row.get("launches"):
                        out["valu_insts_per_launch"] = float(row["SQ_INSTS_VALU"]) / float(row["launches"])
                    if row.get("commit"):
                        out["commit"] = row["commit"]
                    return out
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline_worker(args):
    """Child process: no GPU is touched.  Rebuilds the same synthetic case on the host, times
    the checker, writes JSON to the given file.  Keeps the Fortran runtime's stdout away from
    the parent's single JSON line."""
    ctx = lib.Context()                       # host-side domain logic only
    dom, grid, state, ndte = build_case(ctx, args.workload, 0, 1)
    tcols = None if args.no_thermo else thermo_case(dom, coherent=args.thermo_coherence)[1]
    res = cpu_baseline(args.workload, grid, state, dom, ndte, tcols, args.cpu_seconds)
    with open(args.cpu_baseline_worker, "w") as f:
        json.dump(res, f)


def run_cpu_baseline(args):
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".json", delete=False) as tf:
        path = tf.name
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--cpu-seconds",
           str(args.cpu_seconds), "--cpu-baseline-worker", path, "--thermo-coherence",
           str(args.thermo_coherence)] + (["--no-thermo"] if args.no_thermo else [])
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = str(host_cores())
    try:
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, env=env,
                       timeout=max(180.0, 20.0 * args.cpu_seconds))
        with open(path) as f:
            res = json.load(f)
    except (subprocess.SubprocessError, OSError, ValueError) as e:
        res = {"error": f"{type(e).__name__}: {e}"[:300]}
    if os.path.exists(path):
        os.unlink(path)
    return res


def measure_evp(ctx, args, wl, rank, world, dist, torch, have_torch_gpu, steps, warmup, ramp_seconds, tune=True,
                min_timed_s=1.0, max_repeats=400):
    """W warm-up + exactly K timed steps of the EVP hot loop on resident state for workload `wl`.
    Returns everything the JSON line needs (rank-local cell counts already reduced over the ranks)."""
    progress(f"{wl}: building the synthetic case")
    peer_loop = bool(getattr(args, "peer_loop", False)) and world > 1
    dom, grid, state, ndte = build_case(ctx, wl, rank, world, args.overlap, args.slabs, peer_loop, args.north)
    progress(f"{wl}: case built, device set-up")
    if world > 1:
        # the communicator is created once per context and handed to every decomposition built afterwards
        # (the uid is only used by the first call)
        if os.environ.get("CICE4_AMD_BENCH_LINK") == "shm":
            # diagnostic: the ranks are processes of this host joined by a shared-memory link instead of RCCL (all of them
            # may then sit on ONE device: CICE4_AMD_BENCH_DEVICE=0) -- the multi-process path on a one-GPU box
            if not getattr(ctx, "_comm_ready", False):
                with bounded(args.comm_timeout, "creation of the shared-memory link"):
                    ctx.comm_init_shm("/cice4_amd_bench_%s" % os.environ.get("MASTER_PORT", "0"), rank, world, 256 << 20)
        else:
            uid = [ctx.comm_unique_id() if rank == 0 and not getattr(ctx, "_comm_ready", False) else None]
            with bounded(args.comm_timeout, "creation of the RCCL communicator (ncclCommInitRank)"):
                if not getattr(ctx, "_comm_ready", False):
                    dist.broadcast_object_list(uid, src=0)
                ctx.comm_init(uid[0] if uid[0] is not None else bytes(128), rank, world)
        ctx._comm_ready = True
    ctx.evp_init(grid, ndte=ndte)
    if tune and args.waves:
        ctx.evp_set_option("waves", args.waves)
    if tune and args.rows:
        ctx.evp_set_option("rows_per_wave", args.rows)
    waves, rows = ctx.evp_get_info("waves"), ctx.evp_get_info("rows_per_wave")   # library's own choice by grid size
    tile = f"64x{waves * rows} T-cells ({waves} wavefronts x {rows} rows)"
    if args.no_graph:
        ctx.evp_set_option("use_graph", 0)
    ctx.evp_set_option("derive_metrics", 0 if args.no_derive else 1)
    derive = bool(ctx.evp_get_info("derive_metrics"))
    ctx.evp_set_option("fuse", 0 if args.no_fuse else 1)
    if tune and args.fused_waves:
        ctx.evp_set_option("fused_waves", args.fused_waves)
    fused = bool(ctx.evp_get_info("fused"))
    fw = ctx.evp_get_info("fused_waves") if fused else 0
    if fused:
        tile = f"two subcycles per launch; workgroup {fw} wavefronts x 64 lanes owns {fw - 3} rows x 59 columns"
    ctx.evp_set_option("skew", 0 if args.no_skew else 1)
    if args.skew_levels:
        ctx.evp_set_option("skew_levels", args.skew_levels)
    if args.skew_seg_rows:
        ctx.evp_set_option("skew_seg_rows", args.skew_seg_rows)
    if args.skew_gen_pct >= 0:
        ctx.evp_set_option("skew_gen_pct", args.skew_gen_pct)
    if args.skew_balance >= 0:
        ctx.evp_set_option("skew_balance", args.skew_balance)
    if args.skew_prio >= 0:
        ctx.evp_set_option("skew_prio", args.skew_prio)
    if args.skew_stagger_ns >= 0:
        ctx.evp_set_option("skew_stagger_ns", args.skew_stagger_ns)
    if getattr(args, "skew_split_probe", 0):
        ctx.evp_set_option("skew_split_probe", args.skew_split_probe)
        ctx.evp_set_option("use_graph", 0)
    # (on a folded grid the sweep runs with a band of top rows beside it: "skew_fold"; the sweep is still the launch that counts)
    skew_k = ctx.evp_get_info("skew_levels") if ctx.evp_get_info("skew") or ctx.evp_get_info("skew_fold") else 0
    if skew_k:
        subs = ctx.evp_get_info("skew_subs")
        tile = (f"{skew_k} subcycles per sweep; workgroup = {skew_k * subs} wavefronts ({subs} per time level, levels two rows "
                f"apart) x 64 lanes, owns {62 * subs + 2 - 2 * skew_k} columns x {ctx.evp_get_info('skew_seg_rows')} rows")
        if ctx.evp_get_info("skew_balance"):
            tile += (" to begin with; it walks only the runs of rows that hold ice, and the segments are re-cut from the workgroups' "
                     "measured times in the first loop (places the launch leaves empty go to the slowest strips)")
    if skew_k and dom.get("overlap") and dom["overlap"] % skew_k:
        progress(f"{wl}: {dom['overlap']} overlap rows are no multiple of K = {skew_k}: part of every refresh interval "
                 f"runs the pair kernel instead of sweeps (auto_overlap avoids this; --overlap was given)")
    ctx.evp_set_option("resident", 0 if args.no_resident else 1)
    if peer_loop:
        # every rank hands the IPC handles of its exchange copies / progress words to its neighbours (control plane: gloo)
        share = world if os.environ.get("CICE4_AMD_BENCH_DEVICE") is not None else 1
        ctx.evp_set_option("resident_peer_share", share)
        mine = ctx.evp_peer_export_ipc()
        every = [None] * world
        dist.all_gather_object(every, mine)
        with bounded(args.comm_timeout, "mapping the neighbours' exchange buffers (hipIpcOpenMemHandle)"):
            if rank > 0:
                ctx.evp_peer_connect_ipc(0, every[rank - 1])
            if rank < world - 1:
                ctx.evp_peer_connect_ipc(1, every[rank + 1])
        dist.barrier()
    if tune and args.resident_waves:
        ctx.evp_set_option("resident_waves", args.resident_waves)
    if tune and getattr(args, "resident_prio", -1) >= 0:
        ctx.evp_set_option("resident_prio", args.resident_prio)
    resident = bool(ctx.evp_get_info("resident"))
    rw = ctx.evp_get_info("resident_waves") if resident else 0
    if resident:
        dense = bool(ctx.evp_get_info("resident_dense"))
        tile = (f"whole subcycle loop in one launch, state in registers; workgroup = {rw} wavefronts x 64 lanes "
                f"(owns {rw - 1} rows x 63 columns), " + ("three workgroups per CU (every slot of the chip)" if dense
                                                          else "one workgroup per CU"))
    peer_verified = None
    if peer_loop and getattr(args, "peer_verify", False):
        # the cross-rank loop has to earn its place in THIS run, on THIS hardware: one whole evp(dt) through the
        # per-subcycle message exchange and one through the one-launch loop from the same state, bit for bit
        progress(f"{wl}: verifying the cross-rank loop against the per-subcycle exchange")
        keys = ("uvel", "vvel") + synth.SIG_NAMES
        res = []
        with bounded(args.comm_timeout, "verification of the cross-rank loop"):
            for use_loop in (False, True):
                ctx.evp_set_option("resident", 2 if use_loop else 0)
                sg = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in state.items()}
                ctx.evp(DT, sg)
                res.append(sg)
            same = all(np.array_equal(res[0][k][:, 1:-1], res[1][k][:, 1:-1]) for k in keys)
            moved = float(np.abs(res[1]["uvel"]).max()) > 0.0
            ok = torch.tensor([int(same and moved and ctx.evp_get_info("resident_peer") == 1)], dtype=torch.int64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        del res
        if int(ok[0]) != 1:
            progress(f"{wl}: the cross-rank loop did NOT reproduce the per-subcycle exchange on every rank (or fell back)")
            raise SystemExit(21)
        peer_verified = True
    ctx.evp_upload(state)
    with bounded(args.comm_timeout if world > 1 else 0, "first ghost exchange (evp_prepare: RCCL send / receive)"):
        ctx.evp_prepare(DT)
        ctx.sync()
    nt, nu = ctx.evp_active_cells()

    def sync_all():
        ctx.sync()
        if have_torch_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # bring the device to its sustained clocks first (a few ms of work right after start-up still run at
    # ramping clocks: 91 k instead of 112 k subcycles/s observed once), then the W warm-up steps
    # (a fixed number of steps, agreed by all ranks: every step of a multi-rank run exchanges ghost rows)
    with bounded(args.comm_timeout if world > 1 else 0, "first subcycle loop across the ranks"):
        ctx.evp_subcycles(1, ndte)          # builds the graph
        sync_all()
    t_ramp = time.perf_counter()
    ctx.evp_subcycles(1, ndte)
    sync_all()
    n_ramp = [int(min(5000, max(0, ramp_seconds / max(time.perf_counter() - t_ramp, 1e-6))))]
    if dist is not None:
        dist.broadcast_object_list(n_ramp, src=0)
    for _ in range(n_ramp[0]):
        ctx.evp_subcycles(1, ndte)
    sync_all()
    for _ in range(warmup):
        ctx.evp_subcycles(1, ndte)
    sync_all()
    # N > 1, sweeps between refreshes: the sweep in front of a refresh can run as edge + interior launches with the refresh
    # beside the interior (Evp::launch_subcycle_skew_split).  Whether that pays depends on what the exchange costs on THIS
    # machine -- timed here, both forms, a few steps each, the maximum over the ranks decides for all of them.
    refresh_overlap = None
    if world > 1 and skew_k and dom.get("overlap") and ctx.evp_get_info("skew_trim_ext"):
        tms = {}
        for split in (0, 1):
            ctx.evp_set_option("skew_split", split)     # (a rank that cannot split -- the fold's rank, a short slab -- runs the
            ctx.evp_subcycles(1, ndte)                  #  one-launch form both times: every rank takes part in both timings)
            sync_all()
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.evp_subcycles(1, ndte)
            sync_all()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tms[split] = float(tt[0]) / 3
        use = 1 if tms[1] < tms[0] else 0
        ctx.evp_set_option("skew_split", use)
        refresh_overlap = {"what": "the sweep in front of a refresh as edge + interior launches, the refresh beside the interior",
                           "ms_per_step_one_launch": 1e3 * tms[0], "ms_per_step_split": 1e3 * tms[1], "used": bool(use)}
        progress(f"{wl}: refresh overlap {'ON' if use else 'off'} ({refresh_overlap})")
    progress(f"{wl}: ramp ({n_ramp[0]} steps) and warm-up done, timing {steps} steps")
    # EXACTLY `steps` steps between barrier + synchronise on both sides, maximum over the ranks -- and that block
    # repeated until about `min_timed_s` of device time have been timed (a 13 ms block alone is not a measurement):
    # the reported block is the MEDIAN one; minimum, maximum and the number of blocks go into the line.
    # Two kinds of blocks, alternating: the ones `value` comes from run the steps as a caller does; the others carry the HIP-event
    # bracket of every launch (`us_per_launch` of the roofline) -- two event records per launch cost a step of the one-launch
    # loop 7 us of its 585 (scripts/step_overhead.py; 20 us before the events lived with the object), which is the
    # instrument's time, not the path's.  Both medians are in the line (timing.ms_per_step_with_event_bracket).
    def one_block(events=False):
        sync_all()
        t0 = time.perf_counter()
        dev = 0.0
        for _ in range(steps):
            dev += ctx.evp_subcycles(1, ndte, timed=events)
        sync_all()
        t = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([t, dev], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t, dev = float(tt[0]), float(tt[1])
        return t, dev
    measured_before = ctx.evp_get_info("skew_balanced") if skew_k else 0     # sweeps the library has measured (and synchronised on) so far
    blocks, eblocks = [one_block()], [one_block(True)]
    n_rep = [int(min(max_repeats, max(1, np.ceil(min_timed_s / max(blocks[0][0], 1e-6))))) - 1]
    if dist is not None:
        dist.broadcast_object_list(n_rep, src=0)
    for _ in range(n_rep[0]):
        blocks.append(one_block())
        eblocks.append(one_block(True))
    order = sorted(range(len(blocks)), key=lambda k: blocks[k][0])
    t_evp = blocks[order[len(order) // 2]][0]
    eorder = sorted(range(len(eblocks)), key=lambda k: eblocks[k][0])
    t_events, dev_ms = eblocks[eorder[len(eorder) // 2]]
    measured_inside = (ctx.evp_get_info("skew_balanced") - measured_before) if skew_k else 0
    timing = {"blocks": len(blocks), "steps_per_block": steps, "timed_region_s": sum(b[0] for b in blocks),
              # sweeps whose workgroup times the library read back INSIDE the timed blocks (eager launch + synchronisation each:
              # the re-cut of the segment table, one loop in 96; 0 = every timed sweep replayed the graph)
              "sweeps_measured_inside_timed_region": measured_inside,
              "ms_per_step_median": 1e3 * t_evp / steps, "ms_per_step_min": 1e3 * blocks[order[0]][0] / steps,
              "ms_per_step_max": 1e3 * blocks[order[-1]][0] / steps,
              "event_blocks": len(eblocks), "ms_per_step_with_event_bracket": 1e3 * t_events / steps,
              "note": "value: the median of `blocks` blocks of exactly `steps` steps without per-launch events; roofline.us_per_launch: HIP "
                      "events around every launch in `event_blocks` further blocks of the same steps, alternating with the first kind"}
    if dist is not None:
        cells = torch.tensor([nt, nu], dtype=torch.int64)
        dist.all_reduce(cells, op=dist.ReduceOp.SUM)
        nt_all, nu_all = int(cells[0]), int(cells[1])
    else:
        nt_all, nu_all = nt, nu
    nsub_total = ndte * steps
    value = nsub_total / t_evp
    # dominant kernel: the subcycle kernel; HIP-event time of the launches on the library's stream / launches
    resident = resident and bool(ctx.evp_get_info("resident"))    # 0 if a launch timed out and the library fell back
    if peer_verified and not (resident and ctx.evp_get_info("resident_peer") == 1):
        progress(f"{wl}: the cross-rank loop fell back during the timed steps")
        raise SystemExit(22)
    if resident and ctx.evp_get_info("resident_waves") != rw:     # the library changed the shape while the steps ran
        rw = ctx.evp_get_info("resident_waves")
        dense = bool(ctx.evp_get_info("resident_dense"))
        tile = (f"whole subcycle loop in one launch, state in registers; workgroup = {rw} wavefronts x 64 lanes "
                f"(owns {rw - 1} rows x 63 columns), " +
                ("three workgroups per CU (chosen after the first step: the ice cover leaves most tiles empty)" if dense
                 else "one workgroup per CU (the dense shape timed out on this box)"))
    n_step, main_sub = launches_per_step(ndte, fused, dom.get("overlap", 0), 0 if resident else skew_k)
    n_launch = steps if resident else n_step * steps
    us_per_launch = dev_ms * 1e3 / n_launch
    sub_per_launch = nsub_total / n_launch
    cells_rank = nt_all / world
    # Compulsory HBM bytes of ONE LAUNCH as the kernel is built: the 12 stresses and u, v are read and written
    # once and the 20 read-only inputs are read once per launch, however many subcycles the launch runs
    # (k_subcycle: 1, k_subcycle2: 2) = 384 B per active T-cell per launch.  SURVEY section 8(d) counts the same
    # 384 B per cell per SUBCYCLE; for a two-subcycle launch that figure is twice what has to cross HBM (it gave
    # "fractions" > 1 in round 1), so it is reported separately as `survey_8d` and is NOT a roofline fraction.
    bytes_per_launch = EVP_BYTES_PER_CELL * cells_rank
    achieved = bytes_per_launch / (us_per_launch * 1e-6) / 1e9
    granules = resident and bool(ctx.evp_get_info("resident_granules"))
    kname = ("k_evp_resident (all ndte subcycles in one launch: stresses, metrics, forcing resident in registers / LDS; "
             + ("free-running loop without a workgroup barrier, tile-edge velocities as data-tagged granules: one 16-byte sc1 store, "
                "polled by the lanes that need them)" if granules else
                "tile-edge velocities through agent-scope stores, progress words and agent-scope loads)") if resident else
             f"k_subcycle_skew ({skew_k} subcycles per sweep: a pipeline of {skew_k} time levels, rows handed on through LDS; "
             f"stress + stepu + on-rank halo per level)" if skew_k else
             "k_subcycle2 (two subcycles per launch: stress + stepu + stress + stepu + on-rank halo)" if fused
             else "k_subcycle (fused stress + stepu + on-rank halo)")
    ksub = (f"k_evp_resident<{rw}, false, false, false, {'true' if granules else 'false'}>" if resident else
            f"k_subcycle_skew<{skew_k}, false, false" if skew_k else
            f"k_subcycle2<{fw}, false, false, {'true' if derive else 'false'}>" if fused
            else f"k_subcycle<{waves}, {rows}, false, false, {'true' if derive else 'false'}>")
    traffic, traffic_src = pmc_traffic(wl, ksub) if world == 1 else (None, None)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "basis": "compulsory bytes of one launch: 384 B per active T-cell per launch (sigma, u, v read+written, "
                         "20 inputs read, once per launch whatever the number of subcycles in it)",
                "kernel": kname, "us_per_launch": us_per_launch, "bytes_per_launch": bytes_per_launch,
                "bytes_per_unit": EVP_BYTES_PER_CELL, "unit_of_work": "active T-cell x launch",
                "units_per_launch": cells_rank, "subcycles_per_launch": sub_per_launch,
                "compulsory_bytes_fused": bytes_per_launch,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "frac_measured_traffic": (traffic / (us_per_launch * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "survey_8d": {"bytes_per_unit": EVP_BYTES_PER_CELL, "unit_of_work": "active T-cell x subcycle",
                              "units_per_launch": cells_rank * sub_per_launch,
                              "algorithmic_GBps": achieved * sub_per_launch,
                              "ratio_to_peak": achieved * sub_per_launch / HBM_PEAK_GBS,
                              "note": "north_star's per-subcycle figure (what an unfused subcycle would have to move per "
                                      "second at this rate); exceeds what this kernel moves by the factor subcycles_per_launch"}}
    if resident:
        # The resident loop reads and writes the state once per ndte subcycles: HBM is no longer what bounds it, and the
        # fraction above says only that.  What does: fp64 issue on the busiest SIMD (3 of the 12 wavefronts of a
        # workgroup) and the cross-XCD hand-off of the tile-edge velocities once per subcycle.
        sq = pmc_counters(wl, ksub)
        us_sub = us_per_launch / sub_per_launch
        roofline["not_hbm_bound"] = {
            "us_per_subcycle": us_sub,
            "binding": "fp64 VALU issue of the busiest SIMD + one cross-XCD hand-off (write-through store -> load through memory: 1.3-1.4 us "
                       "store to arrival, measured) per subcycle; the VALU count includes the poll's instructions",
            "valu_inst_per_wave_per_subcycle": sq.get("valu_per_wave_subcycle") if sq else None,
            "issue_us_per_subcycle_3_waves_per_simd_at_2p4GHz": (3 * sq["valu_per_wave_subcycle"] * 4 / 2400.0) if sq else None,
            "counters_source": sq.get("source") if sq else None}
        if sq:   # the fraction that means something for this kernel: fp64 issue slots of the busiest SIMD that are used
            roofline["bound_effective"] = "valu_f64_issue"
            nh = roofline["not_hbm_bound"]
            counted = nh["issue_us_per_subcycle_3_waves_per_simd_at_2p4GHz"] / us_sub
            if granules:
                # The free-running loop WAITS by polling: its counted VALU instructions include the polls' (address arithmetic,
                # tag compares), which are not work.  The arithmetic of a subcycle is the same code as in the barrier loop,
                # whose waits issue next to nothing: 575 VALU instructions per wavefront (profiles/r04_sq_counters.csv,
                # profiles/r02_sq_counters_resident.csv).  `frac_valu_issue` is the fraction by THAT count.
                nh["valu_inst_per_wave_per_subcycle_arithmetic"] = ARITH_VALU_PER_WAVE_SUBCYCLE
                nh["arithmetic_issue_us_per_subcycle_3_waves_per_simd_at_2p4GHz"] = 3 * ARITH_VALU_PER_WAVE_SUBCYCLE * 4 / 2400.0
                roofline["frac_valu_issue_counting_the_polls"] = counted
                roofline["frac_valu_issue"] = nh["arithmetic_issue_us_per_subcycle_3_waves_per_simd_at_2p4GHz"] / us_sub
            else:
                roofline["frac_valu_issue"] = counted
            ghz, src = inkernel_clock(wl)
            if ghz:   # the same fraction at the clock the kernel really holds (2.4 GHz is the nominal figure)
                roofline["clock_ghz"] = ghz
                roofline["clock_source"] = src
                roofline["frac_valu_issue_at_measured_clock"] = roofline["frac_valu_issue"] * 2.4 / ghz
    if skew_k and not resident:
        # The sweep moves the state once per K subcycles and is bound by fp64 issue, not by HBM: both fractions side by side.
        # VALU instructions per launch from the archived SQ pass (every one takes 4 cycles on a 16-lane SIMD; 1,024 SIMDs
        # at 2.4 GHz), divided by the launch time measured in THIS run.
        sq = pmc_counters(wl, ksub) if world == 1 else None
        if sq and sq.get("valu_insts_per_launch"):
            issue_us = sq["valu_insts_per_launch"] * 4.0 / 1024.0 / 2400.0
            roofline["frac_valu_issue"] = issue_us / us_per_launch
            roofline["valu"] = {"insts_per_launch": sq["valu_insts_per_launch"], "issue_us_per_launch_at_2p4GHz": issue_us,
                                "counters_source": sq["source"], "counters_commit": sq.get("commit")}
            if roofline["frac_valu_issue"] > roofline["frac"]:
                roofline["bound_effective"] = "valu_f64_issue"
            ghz, src = inkernel_clock(wl)
            if ghz:   # the sweep holds about 2.1 GHz, not the nominal 2.4: its issue fraction at the clock it runs at
                roofline["clock_ghz"] = ghz
                roofline["clock_source"] = src
                roofline["frac_valu_issue_at_measured_clock"] = roofline["frac_valu_issue"] * 2.4 / ghz
    config = {"workload": workload(wl)[3] if COVER == "full" else workload(wl)[3].replace("full ice cover", f"ice cover '{COVER}' (diagnostic)"),
              "nx_global": dom["nxg"], "ny_global": dom["nyg"], "ndte": ndte,
              "subcycles_per_step": ndte,
              "decomposition": (f"1x{world} classic j-slabs, cross-rank one-launch loop (device-initiated exchange through the "
                                f"neighbours' IPC-mapped exchange copies)" if peer_loop and resident else
                                f"1x{world} j-slabs, one block per GPU") + (
                  f", {dom['overlap']} overlap rows (ghost exchange every {dom['overlap']} subcycles, "
                  f"u, v, 12 stresses in one RCCL message per neighbour)" if dom.get("overlap") else ""),
              "tile": tile, "launches_per_step": 1 if resident else n_step,
              "subcycles_in_the_most_common_launch": ndte if resident else main_sub,
              "metrics_recomputed_from_HTN_HTE": derive,
              "active_T_cells": nt_all, "active_U_cells": nu_all, "cell_subcycles_per_s": value * nt_all}
    if args.north != "open":
        config["north_boundary"] = (f"{args.north} fold (not BASELINE.json's configuration), ocean and ice up to it; "
                                    + ("the rank with the top slab carries the fold (sweeps with a band of top rows beside them "
                                       "where its slab is large enough, else one launch per subcycle), the others run as on "
                                       "any grid" if world > 1 else
                                       "the fold is part of the one-launch loop" if resident else
                                       "sweeps with a band of top rows that carries the fold" if ctx.evp_get_info("skew_fold") else
                                       "halo update with the fold after every subcycle"))
    if refresh_overlap:
        config["refresh_overlap"] = refresh_overlap
    if peer_verified:
        config["peer_loop_verified"] = ("one evp(dt) through the per-subcycle message exchange and one through the cross-rank "
                                        "one-launch loop from the same state: u, v and the 12 stresses bit-identical on every "
                                        "rank, no fall-back before or during the timed steps")
    return dict(dom=dom, grid=grid, state=state, ndte=ndte, value=value, t_evp=t_evp, config=config,
                roofline=roofline, steps=steps, warmup=warmup, timing=timing)


def measure_thermo(ctx, args, wl, dom, world, dist, torch, steps):
    """K batched passes of the column thermodynamics over every (cell, category) of the rank; state restored
    before each pass (not timed)."""
    progress(f"{wl}: thermo case")
    ctx.thermo_init()
    tb, tcols = thermo_case(dom, coherent=args.thermo_coherence)
    progress(f"{wl}: thermo passes")
    ctx.thermo_batch_alloc(dom["nx"], dom["ny"], dom["nblocks"])
    t_ms, nupd = 0.0, 0
    npass = max(2, min(steps, 10))
    for p in range(npass + 1):
        ctx.thermo_batch_upload(tb)
        st = ctx.thermo_batch_step(DT, yday=150.0, timed=True)
        if st["l_stop"]:
            raise SystemExit(f"thermo step failed at i={st['istop']} j={st['jstop']} n={st['nstop']}")
        if p > 0:   # first pass is warm-up
            t_ms += st["ms"]; nupd += st["n_updates"]
    if dist is not None:
        v = torch.tensor([t_ms / npass], dtype=torch.float64); dist.all_reduce(v, op=dist.ReduceOp.MAX)
        c = torch.tensor([nupd // npass], dtype=torch.int64); dist.all_reduce(c, op=dist.ReduceOp.SUM)
        ms_pass, upd_pass = float(v[0]), int(c[0])
    else:
        ms_pass, upd_pass = t_ms / npass, nupd // npass
    rate = upd_pass / (ms_pass * 1e-3)
    traffic, traffic_src = pmc_traffic(wl, "k_thermo_dense") if world == 1 else (None, None)
    thermo = dict(metric="grid-cell-cat-updates/sec", value=rate, unit="(cell,category) updates/s",
                  columns=(("a 512 x 768 patch repeated over the grid; " if (dom["ny"] - 2) * (dom["nx"] - 2) > 2_000_000 else "") +
                           "synthetic, full cover, 40 % melting / 60 % cold, snow-covered and bare, day and night; "
                           + (f"regions with correlation length {args.thermo_coherence} cells"
                              if args.thermo_coherence else "every column drawn independently (white noise)")),
                  updates_per_pass=upd_pass, ms_per_pass=ms_pass, passes=npass,
                  roofline=dict(bound="hbm", achieved=rate * THERMO_BYTES_PER_COLUMN / 1e9,
                                peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=rate * THERMO_BYTES_PER_COLUMN / 1e9 / HBM_PEAK_GBS,
                                traffic=traffic, traffic_unit="bytes per pass (all columns)", traffic_source=traffic_src,
                                frac_measured_traffic=(traffic * world / (ms_pass * 1e-3) / 1e9 / HBM_PEAK_GBS / world
                                                       if traffic else None),
                                kernel="k_thermo_dense", bytes_per_unit=THERMO_BYTES_PER_COLUMN,
                                note="fp64-issue-bound, not HBM-bound (DESIGN.md section 3.3); the fraction says how far "
                                     "the column solver is from streaming its state at HBM speed"))
    return thermo, tcols


def main():
    args = parse()
    global COVER
    COVER = args.cover
    if args.cover != "full":         # diagnostic cover: no CPU leg, no drop-in timing (they are quoted on full cover)
        args.no_cpu_baseline = args.no_dropin_timing = args.no_peer_try = True
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process has not touched a GPU and never will
        raise SystemExit(launch_ranks(args))
    if args.host_only:
        raise SystemExit(host_only(args))
    if args.north != "open":
        if args.peer_loop:
            raise SystemExit("--peer-loop: the cross-rank one-launch loop does not carry a tripole fold (DESIGN.md section 8)")
        args.no_cpu_baseline = args.no_dropin_timing = args.no_peer_try = True
    # NOTE on load order: torch is imported before the product library touches the device.  Both bring a
    # HIP runtime and a librccl.so.1; whichever is loaded first serves the whole process, and loading the
    # system ones first leaves torch's own runtime without a device ("no ROCm-capable device").  With
    # torch first, libcice4_amd.so runs on torch's bundled runtime and RCCL -- the combination
    # scripts/gpu_slabs_selfcomm.sh exercises.
    # stdout carries exactly ONE line (the JSON record, rank 0): everything else that native libraries print
    # there -- RCCL's version banner at communicator creation, Fortran runtime messages -- goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world, local, dist = init_dist(args.gpus, args.comm_timeout)
    if os.environ.get("CICE4_AMD_BENCH_DEVICE") is not None:   # diagnostic: several ranks on one device
        local = int(os.environ["CICE4_AMD_BENCH_DEVICE"])
    import torch                     # (torch.distributed is used for every N > 1: no fallback without it)
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(local)

    ctx = lib.Context(device=local)
    ctx.sync()                       # fails loudly without a GPU / HIP library
    calib = None
    if args.calibrate:
        nd = 32 * 1024 * 1024        # 256 MiB read + 256 MiB written per launch
        ms = ctx.diag_stream_copy(nd)
        calib = {"kernel": "k_diag_copy8", "bytes_read": nd * 8, "bytes_written": nd * 8, "ms": ms,
                 "GBps": 2 * nd * 8 / (ms * 1e-3) / 1e9}
    m = measure_evp(ctx, args, args.workload, rank, world, dist, torch, have_torch_gpu, args.steps, args.warmup,
                    args.ramp_seconds)
    dom, state, ndte = m["dom"], m["state"], m["ndte"]

    # ---- N > 1, gx1: the cross-rank one-launch loop, tried and verified in a second set of processes (see try_peer_loop)
    peer_try = None
    if (world > 1 and args.workload == "gx1" and not args.peer_loop and not args.no_peer_try
            and os.environ.get("CICE4_AMD_BENCH_PEER_CHILD") is None):
        try:
            peer_try = try_peer_loop(args, rank, world, dist)
        except BaseException as e:      # noqa: BLE001 -- nothing of the attempt may take the main measurement down
            if isinstance(e, (KeyboardInterrupt,)):
                raise
            progress(f"gx1: the cross-rank-loop attempt is not used ({e!r})")
            peer_try = {"used": False, "why": repr(e)} if rank == 0 else None

    # ---- the drop-in form evp(dt) with host arrays on both sides (PCIe-inclusive; never `value`)
    pcie = None
    if world == 1 and not args.no_dropin_timing:
        st2 = {k: v.copy() for k, v in state.items()}
        ctx.evp_pin_fields(st2)       # what the Fortran drop-in does with its module arrays on the first call

        def timed_calls(calls=9):
            for _ in range(3):        # the first calls create the copy streams and touch the page-locked ranges
                ctx.evp(DT, st2)
            ts = []
            for _ in range(calls):
                t1 = time.perf_counter()
                ctx.evp(DT, st2)
                ts.append(time.perf_counter() - t1)
            return ts
        ts_ = timed_calls()
        t1 = float(np.median(ts_))
        pcie = {"what": "cice_evp(dt): H2D of 33 fields + prepare + ndte subcycles + finish + D2H of 38 fields (7 of them while the "
                        "subcycle loop runs), host arrays page-locked once (cice_evp_pin_fields), as the Fortran drop-in does",
                "ms_per_call": 1e3 * t1, "ms_per_call_min_max": [1e3 * min(ts_), 1e3 * max(ts_)], "calls": len(ts_),
                "subcycles_per_s": ndte / t1}
        # the caller's two statements (include/cice4_amd.h: both hold for the reference's driver): planes that nobody touches stay
        for key, keep, lazy, what in (
                ("keep_state", 2, 0, "keep_state = 2: uvel, vvel, 12 stresses, iceumask not uploaded again, the 7 flux fields zeroed "
                                     "on the device: H2D of 18 fields, D2H of 38"),
                ("keep_state_lazy_stresses", 2, 1, "keep_state = 2 and lazy_stresses = 1: H2D of 18 fields, D2H of 26 (the stresses on "
                                                   "request: cice_evp_download_stresses)")):
            ctx.evp_set_option("keep_state", keep); ctx.evp_set_option("lazy_stresses", lazy)
            tk = timed_calls()
            pcie[key] = {"what": what, "ms_per_call": 1e3 * float(np.median(tk)), "ms_per_call_min_max": [1e3 * min(tk), 1e3 * max(tk)],
                         "calls": len(tk), "subcycles_per_s": ndte / float(np.median(tk))}
        ctx.evp_set_option("keep_state", 0); ctx.evp_set_option("lazy_stresses", 0)
        ctx.host_unregister_all()     # before the arrays are released (a stale page-locked range faults later)
        del st2

    # ---- thermo (secondary figure)
    thermo, tcols = (None, None) if args.no_thermo else measure_thermo(ctx, args, args.workload, dom, world, dist, torch, args.steps)

    # ---- PCIe-inclusive thermodynamic half-step: cice_step_therm1 (one upload, frzmlt + thermo x ncat + merge_fluxes
    # on the device, one download), host arrays page-locked once.  Never `value`.
    if pcie is not None and thermo is not None:
        progress("gx1: PCIe-inclusive step_therm1")
        tb, _ = thermo_case(dom, coherent=args.thermo_coherence)
        for k in ("fsensn", "fswabsn", "flwoutn", "evapn", "freshn", "fsaltn", "fhocnn"):
            del tb[k]      # locals of step_therm1 that only feed merge_fluxes (CICE_RunMod.F90:296-312): not downloaded
        nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
        fz = dict(aice=np.ascontiguousarray(tb["aicen"].sum(axis=1)), frzmlt=np.full((nb, ny, nx), -5.0),
                  Tf=np.full((nb, ny, nx), -1.8), sst=np.full((nb, ny, nx), -1.7), strocnxT=np.full((nb, ny, nx), 0.05),
                  strocnyT=np.full((nb, ny, nx), 0.02), Tbot=np.zeros((nb, ny, nx)), fbot=np.zeros((nb, ny, nx)),
                  rside=np.zeros((nb, ny, nx)))
        pc = {k: np.zeros(tb["aicen"].shape) for k in ("strairxn", "strairyn", "Trefn", "Qrefn")}
        acc = {k: np.zeros((nb, ny, nx)) for k in lib.MERGE_ORDER}
        atm = dict(uatm=np.full((nb, ny, nx), 5.0), vatm=np.full((nb, ny, nx), 5.0), wind=np.full((nb, ny, nx), 50.0 ** 0.5),
                   zlvl=np.full((nb, ny, nx), 10.0))
        keep = [tb, fz, pc, acc, atm]
        for d in keep:
            for v in d.values():
                if isinstance(v, np.ndarray):
                    ctx.host_register(v)
        state0 = {k: tb[k].copy() for k in lib.THERMO_STATE + lib.THERMO_SW + lib.THERMO_ONSET}
        times = []
        for _ in range(7):
            for k, v in state0.items():
                tb[k][...] = v
            t1 = time.perf_counter()
            st = ctx.step_therm1(DT, 150.0, tb, fz, pc, acc)
            times.append(time.perf_counter() - t1)
        times_abl = []
        for _ in range(7):     # the same with atmo_boundary_layer on the device: 26 planes fewer to upload
            for k, v in state0.items():
                tb[k][...] = v
            t1 = time.perf_counter()
            st_abl = ctx.step_therm1(DT, 150.0, tb, fz, {}, acc, atm=atm)
            times_abl.append(time.perf_counter() - t1)
        ctx.host_unregister_all()
        del keep, tb, fz, pc, acc, atm
        if not st["l_stop"]:
            pcie["step_therm1"] = {"what": "cice_step_therm1: ONE upload (state, forcing, shortwave, per-category atmo outputs, "
                                           "20 accumulators ~ 150 planes), frzmlt_bottom_lateral + thermo_vertical x 5 categories + "
                                           "merge_fluxes on the device, ONE download (~80 planes: state, shortwave, the per-category module arrays, accumulators, Tbot/fbot/rside); page-locked host arrays",
                                   "ms_per_call": 1e3 * float(np.median(times[2:])), "ms_per_call_min_max": [1e3 * min(times[2:]), 1e3 * max(times[2:])],
                                   "calls": len(times) - 2, "column_updates": st["n_updates"],
                                   "updates_per_s": st["n_updates"] / float(np.median(times[2:]))}
            if not st_abl["l_stop"]:
                pcie["step_therm1"]["with_atmo_boundary_layer_on_device_ms"] = 1e3 * float(np.median(times_abl[2:]))
            pcie["therm1_plus_evp_ms"] = pcie["ms_per_call"] + pcie["step_therm1"]["ms_per_call"]
            pcie["resident_ms"] = 1e3 * m["t_evp"] / args.steps + thermo["ms_per_pass"]

    # ---- PCIe-inclusive horizontal transport: cice_transport_remap = upload of the state,
    # state_to_tracers -> horizontal_remap -> tracers_to_state -> bound_state on the device, download.  Never `value`.
    if pcie is not None and thermo is not None:
        progress("gx1: PCIe-inclusive transport_remap")
        try:
            tb, _ = thermo_case(dom, coherent=args.thermo_coherence)
            g = m["grid"]
            ctx.transport_init({k: g[k] for k in ("HTN", "HTE", "dxt", "dyt", "dxu", "dyu", "tarear", "hm")},
                               ntrcr=2, trcr_depend=(0, 1))
            tot = tb["aicen"].sum(axis=1)
            sc_ = np.where(tot > 0.95, 0.95 / np.maximum(tot, 1e-30), 1.0)[:, None]
            ts = {k: np.ascontiguousarray(tb[k] * sc_) for k in ("aicen", "vicen", "vsnon", "eicen", "esnon")}
            ts["trcrn"] = np.ascontiguousarray(tb["trcrn"])
            ts["aice0"] = np.ascontiguousarray(1.0 - ts["aicen"].sum(axis=1))
            nb, ny, nx = dom["nblocks"], dom["ny"], dom["nx"]
            jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
            ts["uvel"] = np.array(np.broadcast_to(0.2 * np.sin(2 * np.pi * ii / nx) * np.cos(np.pi * jj / ny), (nb, ny, nx)))
            ts["vvel"] = np.array(np.broadcast_to(0.2 * np.cos(2 * np.pi * ii / nx) * np.sin(2 * np.pi * jj / ny), (nb, ny, nx)))
            for v in ts.values():
                ctx.host_register(v)
            t0s = {k: v.copy() for k, v in ts.items()}
            ttimes = []
            for _ in range(7):
                for k, v in t0s.items():
                    ts[k][...] = v
                t1 = time.perf_counter()
                rc = ctx.transport_remap(DT, ts)
                ttimes.append(time.perf_counter() - t1)
            ctx.host_unregister_all()
            if rc == (0, 0, 0):
                pcie["transport_remap"] = {
                    "what": "cice_transport_remap (advection = 'remap', ncat = 5, 9 tracers per category incl. 5 enthalpies): upload "
                            "of the state (36 planes) + velocities, seven kernels + nine halo updates on the device, download (34 planes); "
                            "page-locked host arrays", "ms_per_call": 1e3 * float(np.median(ttimes[2:])),
                    "ms_per_call_min_max": [1e3 * min(ttimes[2:]), 1e3 * max(ttimes[2:])], "calls": len(ttimes) - 2}
            del ts, t0s, tb
        except lib.CiceError as e:
            progress("transport timing skipped: %s" % e)

    # ---- the 0.1-degree configuration (BASELINE.json configs[4]) inside the same line: a short run, EVP only + thermo
    tenth = None
    if args.workload == "gx1" and not args.no_tenth:
        del state
        m["state"] = m["grid"] = None
        t = measure_evp(ctx, args, "tenth", rank, world, dist, torch, have_torch_gpu, args.tenth_steps, 1, 0.3, tune=False,
                        min_timed_s=1.0, max_repeats=8)
        tenth = {"metric": "EVP subcycles/sec", "value": t["value"], "unit": "subcycles/s", "n_gpus": world,
                 "steps": t["steps"], "warmup": t["warmup"], "ms_per_step": 1e3 * t["t_evp"] / t["steps"],
                 "timing": t["timing"], "config": t["config"], "roofline": t["roofline"]}
        if not args.no_thermo:
            tenth["thermo"] = measure_thermo(ctx, args, "tenth", t["dom"], world, dist, torch, 2)[0]
        del t
        # the same grid under the ice cover a global model has -- two polar caps, most rows open water (diagnostic: the
        # library walks only the rows that hold ice and re-cuts its segment table from measured workgroup times)
        if world == 1 and COVER == "full" and not args.no_caps:
            COVER = "caps"
            try:
                t = measure_evp(ctx, args, "tenth", rank, world, dist, torch, have_torch_gpu, args.tenth_steps, 1, 0.3, tune=False,
                                min_timed_s=0.5, max_repeats=4)
                tenth["polar_caps_cover"] = {"value": t["value"], "unit": "subcycles/s", "us_per_subcycle": 1e6 / t["value"],
                                             "steps": t["steps"], "timing": t["timing"], "workload": t["config"]["workload"],
                                             "note": "diagnostic, not the headline workload: ice on 25 % of the rows (synth.evp_state cover='caps')"}
                del t
            finally:
                COVER = "full"

    if rank == 0:
        out = {
            "metric": "EVP subcycles/sec", "value": m["value"], "unit": "subcycles/s", "n_gpus": world,
            "ranks_seen": ctx.comm_count() if world > 1 else 1,     # ncclCommCount: what RCCL itself says
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * m["t_evp"] / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            # `steps` x `ms_per_step` describes ONE block of exactly K steps (the median block); the timed region as a whole
            # is that block repeated `timing.blocks` times, each between barrier + synchronise: this many seconds
            "timed_region_s": m["timing"]["timed_region_s"], "timed_blocks": m["timing"]["blocks"],
            "timing": m["timing"], "config": m["config"], "roofline": m["roofline"],
        }
        if peer_try:
            # two decompositions were measured; `value` is the faster one, the other stays in the line
            slabs = {"value": out["value"], "ms_per_step": out["ms_per_step"], "timing": out["timing"], "config": out["config"]}
            if peer_try.get("value") and peer_try["value"] > out["value"]:
                peer_try["used"] = True
                out.update(value=peer_try["value"], ms_per_step=peer_try["ms_per_step"], timing=peer_try["timing"],
                           config=peer_try["config"], roofline=peer_try["roofline"] or out["roofline"])
                out["other_decomposition"] = dict(slabs, what="wide-halo slabs, RCCL exchange every `overlap` subcycles")
            else:
                if peer_try.get("value"):
                    peer_try["used"] = False
                out["other_decomposition"] = dict(peer_try, what="classic slabs, cross-rank one-launch loop with device-initiated "
                                                                 "exchange (second set of rank processes)")
        if thermo:
            out["thermo"] = thermo
        if tenth:
            out["tenth"] = tenth
        if calib:
            out["calibration"] = calib
        if pcie:
            out["pcie_inclusive"] = pcie
        # the archived counter passes this line quotes (traffic, VALU counts, in-kernel clock): do they describe THIS tree's kernels?
        out["counters_stale"] = bool(STALE)
        out["counters"] = {"kernel_source_sha": kernel_source_sha(), "stale_sources": list(STALE),
                           "note": "traffic, frac_valu_issue and clock_ghz come from archived rocprofv3 passes under profiles/ "
                                   "(not measured in this run); stale: taken from kernel sources other than this tree's"}
        if world == 1 and not args.no_cpu_baseline:
            progress("cpu baseline (child process)")
            cb = run_cpu_baseline(args)
            out["cpu_baseline"] = dict(cb.get("evp") or {"value": None, "unit": "EVP subcycles/s", "cores": 0,
                                                         "kind": "reference", "sample": "not measured: " + cb.get("error", "?")})
            if "evp_all_cores" in cb:
                out["cpu_baseline"]["all_cores"] = cb["evp_all_cores"]
            if "thermo" in cb and thermo:
                out["thermo"]["cpu_baseline"] = cb["thermo"]
                if "thermo_all_cores" in cb:
                    out["thermo"]["cpu_baseline"]["all_cores"] = cb["thermo_all_cores"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
