import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cice4_amd import lib, synth
from oracle import oracle
import test_gpu_thermo as T
oracle.build(); orc = oracle.Oracle(); ctx = lib.Context()
DT = 3600.0
ny, nx, nb = 26, 40, 2
batch, percat = T._batch_inputs(ny, nx, nb, seed=33)
orc.init_thermo()
inputs = {}
for b in range(nb):
    for n in range(5):
        a, icells, ii, jj = percat[(b, n)]
        a = {k: v.copy() for k, v in a.items()}
        for k in lib.THERMO_FORCING:
            a[k] = percat[(b, 0)][0][k].copy()
        t = {k: v.copy() for k, v in a.items()}
        assert orc.thermo_vertical(DT, icells, ii, jj, t, yday=150.0)[0] == 0
        kb = synth.known_tsfc_inputs(a, t, seed=5 * b + n)
        inputs[(b, n)] = kb
        for k in ("fsurfn", "fcondtopn", "flatn"):
            batch[k][b, n] = kb[k]
        batch["trcrn"][b, n] = kb["trcrn"]
ctx.thermo_init(calc_Tsfc=False); orc.init_thermo(calc_Tsfc=False)
for b in range(nb):
    for n in range(5):
        _, icells, ii, jj = percat[(b, n)]
        ac = {k: v.copy() for k, v in inputs[(b, n)].items()}
        print("orc", b, n, orc.thermo_vertical(DT, icells, ii, jj, ac, yday=150.0), icells)
        ag = {k: v.copy() for k, v in inputs[(b, n)].items()}
        print("gpu list", b, n, ctx.thermo_vertical(DT, icells, ii, jj, ag, yday=150.0))
ctx.thermo_batch_alloc(nx, ny, nb)
ctx.thermo_batch_upload(batch)
print(ctx.thermo_batch_step(DT, yday=150.0))
