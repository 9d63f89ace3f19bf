"""bound_state and a 2-d ice_HaloUpdate through the reference's own boundary module and through the drop-in one, gx1 size
(needs oracle/_ref/libcice_{ref,dropin}_gx1.so; the drop-in's init needs a GPU)."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, time, tempfile, numpy as np
sys.path.insert(0, %r)
from oracle import refapi
kind = sys.argv[1]
r = refapi.Ref("gx1", kind=kind) if kind != "ref" else refapi.Ref("gx1")
r.init_domain(tempfile.mkdtemp(), dt=3600.0, ndte=120, ew="cyclic", ns="open")
nb, ny, nx = r.max_blocks, r.ny_block, r.nx_block
rng = np.random.default_rng(1)
NC, NI, NS, NT = refapi.NCAT, refapi.NILYR, refapi.NSLYR, refapi.MAX_NTRCR
st = [rng.uniform(0, 1, (nb, NC, ny, nx)), rng.uniform(0, 1, (nb, NC, NT, ny, nx)), rng.uniform(0, 1, (nb, NC, ny, nx)),
      rng.uniform(0, 1, (nb, NC, ny, nx)), rng.uniform(0, 1, (nb, NC * NI, ny, nx)), rng.uniform(0, 1, (nb, NC * NS, ny, nx))]
r.bound_state(*st)
t = time.perf_counter()
for _ in range(20): r.bound_state(*st)
t1 = (time.perf_counter() - t) / 20 * 1e3
a = rng.uniform(0, 1, (nb, ny, nx))
t = time.perf_counter()
for _ in range(200): r.halo_nd(a, 2, 2)
t2 = (time.perf_counter() - t) / 200 * 1e6
sys.stderr.write("RESULT %%s bound_state %%.3f ms, 2-d update %%.1f us\\\\n" %% (kind, t1, t2))
''' % ROOT
for kind in ("ref", "dropin"):
    p = subprocess.run([sys.executable, "-c", code, kind], capture_output=True, text=True)
    print([l for l in p.stderr.splitlines() if l.startswith("RESULT")] or p.stderr[-400:])
