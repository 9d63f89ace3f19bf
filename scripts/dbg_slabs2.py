import os, sys; sys.path.insert(0, '.')
os.environ["CICE4_AMD_SELF_COMM"] = "1"
import numpy as np
from cice4_amd import lib, synth
c = lib.Context(); c.sync()
dom = c.domain_create_slabs(96, 72, 4, ew=1, ns=0, overlap=4)
c.comm_init(c.comm_unique_id(), 0, 1)
del os.environ["CICE4_AMD_SELF_COMM"]
dp = lib.Context().domain_create_slabs(96, 72, 4, ew=1, ns=0, overlap=4)
print("msgs", dom["nsend_elems"], dom["nrecv_elems"], "plain refresh", len(dp["rsrc"]), "wrap", len(dp["hsrc"]), len(dom["hsrc"]))
s_addr = c.halo_msgs(0)[0][1]; r_addr = c.halo_msgs(1)[0][1]
print("lists equal to plain refresh:", np.array_equal(s_addr, dp["rsrc"]), np.array_equal(r_addr, dp["rdst"]))
rng = np.random.default_rng(0)
for nlev in (1, 2, 4, 5, 14):
    a = rng.uniform(0, 1, (nlev, dom["nblocks"], dom["ny"], dom["nx"]))
    want = a.copy().reshape(nlev, -1)
    want[:, dp["hdst"]] = want[:, dp["hsrc"]]
    want[:, dp["rdst"]] = want[:, dp["rsrc"]]
    got = a.copy(); c.halo_update(got)
    bad = np.argwhere(got.reshape(nlev, -1) != want)
    print("nlev", nlev, "nbad", len(bad), "levels", sorted(set(bad[:, 0]))[:20] if len(bad) else "")
