"""Child process of tests/test_gpu_fullsize.py::test_gx1_on_eight_ranks[peer]: EIGHT cross-rank one-launch loops on ONE GPU.
Every rank's loop waits for its neighbours' loops, so all eight launches have to RUN AT THE SAME TIME -- on a node each has a
GPU of its own; on one GPU each needs a hardware queue of its own, and the runtime multiplexes its streams on four unless
GPU_MAX_HW_QUEUES says otherwise.  The variable is read when the runtime starts: hence a process of its own (the parent
sets it).  Exit code 0: every owned cell of every rank equals the checker's whole-grid run, bit for bit.
usage: ranks_peer_case.py npx npy nxg nyg ndte [ns]
ns = 3 / 4: a tripole north boundary (full-width slabs: the rank of the top slab folds inside its loop); the comparison is then
with the one-block domain run through one launch per subcycle (pinned to the compiled reference on such grids:
tests/tripole_evp_case.py), the checker's C restatement has no fold."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
try:
    import torch
    torch.cuda.is_available()
except ImportError:
    pass
from cice4_amd import lib, synth  # noqa: E402
from oracle import oracle  # noqa: E402
import ranks_case  # noqa: E402

DT = 3600.0


def main():
    npx, npy, nxg, nyg, ndte = (int(x) for x in sys.argv[1:6])
    ns = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    R = npx * npy
    gg = synth.global_grid(nxg, nyg, perturb=0.1, land_frac=0.03, seed=31, **({"land_rows": 0} if ns else {}))
    # (the eight loops first: a context that has run kernels and copies keeps hardware queues busy that the loops need)
    out = ranks_case.run_ranks(gg, R, "peer", ndte, DT, ns=ns, seed=31, cover="patchy", npx=npx)
    c1 = lib.Context()
    dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=ns)
    if ns:
        grid1 = synth.block_fields(gg, dom1, ew_cyclic=True, north_ocean=True)
        s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
        c1.evp_init(grid1, ndte=ndte, krdg_partic=0, krdg_redist=0)
        c1.evp_set_option("resident", 0); c1.evp_set_option("skew", 0); c1.evp_set_option("skew_fold", 0)
        c1.evp(DT, s1)
        assert np.abs(s1["uvel"][0, -3:]).max() > 1e-4        # ice moves at the fold
    else:
        grid1 = synth.block_fields(gg, dom1)
        s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
        orc = oracle.Oracle()
        orc.set_evp_parameters(DT, ndte, False); orc.set_strength_parameters(1, 0, 0, 4.0)
        orc.evp(orc.make_domain(dom1, grid1), s1)
    one = dict(nxg=nxg, nyg=nyg, nblocks=1, j0=[0], jlo=dom1["jlo"], jhi=dom1["jhi"], own_jlo=dom1["jlo"],
               own_jhi=dom1["jhi"], ilo=dom1["ilo"], ihi=dom1["ihi"])
    bad = 0
    for k in ("uvel", "vvel") + synth.SIG_NAMES + ("divu", "shear", "strength", "strocnxT", "strocnyT", "strintx", "prs_sig"):
        want, got = ranks_case.owned(one, s1[k]), ranks_case.assemble_blocks(out, k, nxg, nyg)
        if not np.array_equal(got, want):
            print("MISMATCH", k, np.argwhere(got != want)[:5].tolist(), flush=True)
            bad += 1
    print("peer loop on", npx, "x", npy, "ranks:", "bit-identical" if not bad else f"{bad} fields differ",
          "| max |u| =", float(np.abs(s1["uvel"]).max()), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
