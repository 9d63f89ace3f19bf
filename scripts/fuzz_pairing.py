"""Randomised check of k_subcycle2 (two subcycles per launch) and of k_evp_resident (the whole loop in one launch)
against k_subcycle (one subcycle per launch): random grid sizes, boundary types, subcycle counts, damping, workgroup
heights and shapes, ice cover.  Bit-for-bit on every output field.  usage: python scripts/fuzz_pairing.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cice4_amd import lib, synth  # noqa: E402

OUT = ("uvel", "vvel", "strength", "divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strocnxT", "strocnyT",
       "strocnx", "strocny", "strintx", "strinty", "strairx", "strairy", "fm", "strtltx", "strtlty",
       "iceumask") + synth.SIG_NAMES


def run(ctx, grid, s, ndte, damping, **opts):
    sg = {k: v.copy() for k, v in s.items()}
    ctx.evp_init(grid, ndte=ndte, evp_damping=damping)
    for k, v in opts.items():
        ctx.evp_set_option(k, v)
    fused = ctx.evp_get_info("fused")
    ctx.evp(3600.0, sg)
    return sg, fused


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = lib.Context(); ctx.sync()
    for case in range(ncases):
        nxg = int(rng.choice([rng.integers(5, 70), rng.integers(55, 65), rng.integers(110, 125), rng.integers(170, 260)]))
        nyg = int(rng.choice([rng.integers(5, 12), rng.integers(12, 40), rng.integers(40, 90)]))
        ew = int(rng.choice([1, 1, 1, 0, 2]))
        ns = int(rng.choice([0, 0, 1, 2]))
        ndte = int(rng.choice([1, 2, 3, 8, 11, 24]))
        damping = bool(rng.integers(0, 2))
        cover = str(rng.choice(["full", "patchy"]))
        fw = int(rng.choice([0, 8, 12, 13, 14, 16]))
        rw = int(rng.choice([0, 4, 6, 8, 11, 12]))
        dense = int(rng.integers(0, 2))
        dom = ctx.domain_create(nxg, nyg, nxg, nyg, ew=ew, ns=ns)
        gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=100 + case)
        grid = synth.block_fields(gg, dom, ew_cyclic=(ew == 1))
        s = synth.evp_state(grid, dom, seed=case, cover=cover)
        # ghost cells the reference never writes may hold anything: make them differ from their mirrors
        for k in ("uvel", "vvel") + synth.SIG_NAMES:
            s[k][:, 0, :] += rng.uniform(-0.01, 0.01, s[k][:, 0, :].shape)
            s[k][:, -1, :] += rng.uniform(-0.01, 0.01, s[k][:, -1, :].shape)
            s[k][:, :, -1] += rng.uniform(-0.01, 0.01, s[k][:, :, -1].shape)
        ref, f0 = run(ctx, grid, s, ndte, damping, fuse=0, resident=0)
        variants = []
        if ns != 1:     # (a cyclic N-S edge rewrites ghost rows every subcycle: no pairing there)
            got, f1 = run(ctx, grid, s, ndte, damping, fuse=1, fused_waves=fw, resident=0)
            assert (f0, f1) == (0, 1)
            variants.append((f"pairs W={fw}", got))
        got, _ = run(ctx, grid, s, ndte, damping, fuse=1, resident=2, resident_waves=rw, resident_dense=dense)
        assert ndte < 2 or ctx.evp_get_info("resident") == 1, "the resident loop fell back"
        variants.append((f"one launch W={rw} dense={dense}", got))
        for name, got in variants:
            for k in OUT:
                if not np.array_equal(got[k], ref[k]):
                    bad = np.argwhere(got[k] != ref[k])
                    raise SystemExit(f"MISMATCH case {case} ({name}): nxg={nxg} nyg={nyg} ew={ew} ns={ns} ndte={ndte} "
                                     f"damping={damping} cover={cover} field={k} first={bad[0].tolist()} n={len(bad)}")
        print(f"case {case}: {nxg}x{nyg} ew={ew} ns={ns} ndte={ndte} damp={int(damping)} {cover} W={fw} rw={rw} dense={dense} ok", flush=True)
    print("FUZZ-OK", ncases)


if __name__ == "__main__":
    main()
