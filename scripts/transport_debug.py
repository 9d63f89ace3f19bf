"""Stage-by-stage comparison of the device remap with the compiled reference (GPU box; diagnostic)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "tests", "transport_case.py")).read()
# reuse the case set-up up to the reference call, then branch into the staged comparison
head = src[:src.index("        ref.transport_remap(DT)")]
body = head + '''
        NTR = 9
        aim, trm = ref.state_to_tracers(NTR)
        ctx.transport_debug(stop_stage=1)
        ctx.transport_remap(DT, {k: v.copy() for k, v in dev.items()})
        N = nbl * ny * nx
        mm = ctx.transport_debug(1, 0).reshape(NC + 1, nbl, ny, nx)
        tm = ctx.transport_debug(1, 1).reshape(NC, NTR, nbl, ny, nx)
        print("stage1 mm equal", np.array_equal(mm.transpose(1, 0, 2, 3), aim[:nbl]), "tm equal",
              np.array_equal(tm.transpose(2, 0, 1, 3, 4), trm[:nbl]))
        a2, t2 = aim.copy(), trm.copy()
        ee, en = ref.horizontal_remap(DT, a2, t2)
        ctx.transport_debug(stop_stage=4)
        ctx.transport_remap(DT, {k: v.copy() for k, v in dev.items()})
        mm4 = ctx.transport_debug(4, 0).reshape(NC + 1, nbl, ny, nx).transpose(1, 0, 2, 3)
        mflx = ctx.transport_debug(4, 5).reshape(2, NC + 1, nbl, ny, nx)
        fe, fn = mflx[0, 0], mflx[1, 0]
        tar = grid["tarear"]
        exp = aim[:nbl, 0].copy()
        w1 = np.zeros_like(exp)
        w1[:, 1:, 1:] = (fe[:, 1:, 1:] - fe[:, 1:, :-1]) + fn[:, 1:, 1:] - fn[:, :-1, 1:]
        phys = np.zeros_like(exp, bool)
        for b in range(nbl):
            phys[b, dom["jlo"][b] - 1:dom["jhi"][b], dom["ilo"][b] - 1:dom["ihi"][b]] = True
        exp = np.where(phys, exp - w1 * tar, exp)
        exp = np.where(phys & (exp < 0) & (exp >= -1e-11), 0.0, exp)
        print("update from device fluxes vs device mm:", np.abs(exp - mm4[:, 0]).max(), " vs reference mm:", np.abs(exp - a2[:nbl, 0]).max())
        d = np.abs(mm4[:, 0] - a2[:nbl, 0])
        bad = np.argwhere(d > 0)
        print("cells differing cat0:", len(bad), "rows", sorted(set(bad[:, 1].tolist()))[:40], "cols", sorted(set(bad[:, 2].tolist()))[:40])
        print("hm zero cells:", int((grid["hm"] == 0).sum()), "of", grid["hm"].size, " aim0 ghost row0:", aim[0, 0, 0, :5], aim[0, 0, 1, :5])
        break
'''
sys.argv = ["x", "small", "cyclic", "open"]
exec(compile(body + "\nmain()\n" if False else body.replace('if __name__ == "__main__":', 'if False:') + "\nmain()\n", "case", "exec"))
