import sys; sys.path.insert(0, '.')
import numpy as np
from cice4_amd import lib, synth
from oracle import oracle
orc = oracle.Oracle()
DT, NDTE = 3600.0, 4
nxg, nyg, nb = 96, 72, 4
c1 = lib.Context(); dom1 = c1.domain_create(nxg, nyg, nxg, nyg, ew=1, ns=0)
gg = synth.global_grid(nxg, nyg, perturb=0.15, land_frac=0.05, seed=31)
grid1 = synth.block_fields(gg, dom1); s1 = synth.evp_state(grid1, dom1, seed=31, cover="patchy")
orc.set_evp_parameters(DT, NDTE, False); orc.set_strength_parameters(1, 0, 0, 4.0)
s1in = {k: v.copy() for k, v in s1.items()}
orc.evp(orc.make_domain(dom1, grid1), s1)
for ov in (0, 2):
    c = lib.Context(); c.sync()
    dom = c.domain_create_slabs(nxg, nyg, nb, ew=1, ns=0, overlap=ov)
    grid = synth.block_fields(gg, dom); s = synth.evp_state(grid, dom, seed=31, cover="patchy")
    # inputs consistent with the single-domain inputs?
    for k in ("aice", "uvel", "stressp_1", "strairxT"):
        for b in range(nb):
            r0 = dom["j0"][b]; nr = dom["jhi"][b] - dom["jlo"][b] + 1
            a = s[k][b, dom["jlo"][b]-1:dom["jhi"][b], 1:-1]; bref = s1in[k][0, 1 + r0:1 + r0 + nr, 1:-1]
            if not np.array_equal(a, bref): print("INPUT MISMATCH", ov, k, b, np.abs(a-bref).max())
    for k in ("dxt", "cxp", "fcor", "tmask"):
        for b in range(nb):
            r0 = dom["j0"][b]; nr = dom["jhi"][b] - dom["jlo"][b] + 1
            a = grid[k][b, dom["jlo"][b]-1:dom["jhi"][b], 1:-1]; bref = grid1[k][0, 1 + r0:1 + r0 + nr, 1:-1]
            if not np.array_equal(a, bref): print("GRID MISMATCH", ov, k, b)
    c.evp_init(grid, ndte=NDTE, krdg_partic=0, krdg_redist=0)
    c.evp(DT, s)
    for k in ("strength", "uvel", "stressp_1"):
        for b in range(nb):
            r0 = dom["j0"][b] + (dom["own_jlo"][b] - dom["jlo"][b]); nr = dom["own_jhi"][b] - dom["own_jlo"][b] + 1
            a = s[k][b, dom["own_jlo"][b]-1:dom["own_jhi"][b], 1:-1]; bref = s1[k][0, 1 + r0:1 + r0 + nr, 1:-1]
            bad = np.argwhere(a != bref)
            print(ov, k, b, "nbad", len(bad), "rows", sorted(set(bad[:, 0]))[:12] if len(bad) else "")
