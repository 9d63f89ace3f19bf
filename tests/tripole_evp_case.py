"""One whole evp(dt) of the COMPILED REFERENCE (source/ice_dyn_evp.F90:119-432 with serial/ice_boundary.F90's tripole
fold after every subcycle, :397-402) on a one-block domain with a tripole north boundary, against cice_evp on the MI355X
running the one-launch loop with the fold inside.  Own process: the reference allows one init_domain per process.

    python tests/tripole_evp_case.py <tripole|tripoleT> [cfg]

Prints 'TRIPOLE-EVP-OK <n checks>'."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BND = {"tripole": 3, "tripoleT": 4}
DT, NDTE = 3600.0, 120


def main():
    ns = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "gx3"
    from __graft_entry__ import REF_CONFIGS
    from cice4_amd import lib, synth
    from oracle import refapi
    from test_oracle_vs_ref import inject, EVP_OUT
    nxg, nyg, bsx, bsy, mxb = REF_CONFIGS[cfg]
    assert mxb == 1
    ref = refapi.Ref(cfg)
    assert ref.init_domain(tempfile.mkdtemp(), dt=DT, ndte=NDTE, ew="cyclic", ns=ns) == 1
    ctx = lib.Context(); ctx.sync()
    dom = ctx.domain_create(nxg, nyg, bsx, bsy, ew=1, ns=BND[ns])
    grid = synth.block_fields(synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=4, land_rows=0), dom, ew_cyclic=True, north_ocean=True)   # ocean up to the fold
    nchk = 0
    for cover, damping, exact in (("patchy", False, True), ("full", True, True), ("patchy", False, False)):
        s = synth.evp_state(grid, dom, seed=4, cover=cover)
        kp = 0 if exact else 1      # exp-free strength: bit for bit whatever the host's libm
        ref.set_evp_parameters(DT, NDTE, damping); ref.set_strength_parameters(1, kp, kp, 4.0)
        inject(ref, grid, s, dom)
        ref.evp(DT)
        for fold_in_loop in (1, 0):
            sg = {k: v.copy() for k, v in s.items()}
            ctx.evp_init(grid, ndte=NDTE, evp_damping=damping, krdg_partic=kp, krdg_redist=kp)
            ctx.evp_set_option("resident", 2); ctx.evp_set_option("resident_fold", fold_in_loop)
            assert ctx.evp_get_info("resident") == fold_in_loop
            ctx.evp(DT, sg)
            assert ctx.evp_get_info("resident") == fold_in_loop, "fell back"
            from conftest import TOL_EXP
            for k in EVP_OUT:
                w = ref.get(k)
                if exact or TOL_EXP == 0.0:
                    assert np.array_equal(w, sg[k]), (ns, cover, damping, fold_in_loop, k, np.argwhere(w != sg[k])[:6].tolist())
                else:
                    assert np.abs(w - sg[k]).max() <= 1e-8 * max(np.abs(w).max(), 1e-300), k
                nchk += 1
            assert np.array_equal(ref.get("iceumask"), sg["iceumask"])
        assert np.abs(ref.get("uvel")).max() > 0.01 and np.abs(ref.get("uvel")[0, -3:]).max() > 1e-4   # ice moves at the fold
    print("TRIPOLE-EVP-OK", nchk)


if __name__ == "__main__":
    main()
