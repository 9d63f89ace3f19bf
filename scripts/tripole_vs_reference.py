"""evp(dt) of the compiled reference on a one-block 100 x 116 domain with a tripole north boundary and ocean up to the fold
against the GPU paths (one-launch loop with the fold inside; one launch per subcycle + halo update).
usage: python scripts/tripole_vs_reference.py <tripole|tripoleT|open> [ndte]"""
import sys, tempfile; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from __graft_entry__ import REF_CONFIGS
from cice4_amd import lib, synth
from oracle import refapi
from test_oracle_vs_ref import inject, EVP_OUT
ns=sys.argv[1]; BND={"tripole":3,"tripoleT":4,"open":0}
nxg,nyg,bsx,bsy,mxb=REF_CONFIGS["gx3"]
ref=refapi.Ref("gx3"); ref.init_domain(tempfile.mkdtemp(),dt=3600.0,ndte=120,ew="cyclic",ns=ns)
ctx=lib.Context(); dom=ctx.domain_create(nxg,nyg,bsx,bsy,ew=1,ns=BND[ns])
grid=synth.block_fields(synth.global_grid(nxg,nyg,perturb=0.15,land_frac=0.05,seed=4,land_rows=0),dom,ew_cyclic=True,north_ocean=(ns!="open"))
s=synth.evp_state(grid,dom,seed=4,cover="patchy")
ND=int(sys.argv[2]) if len(sys.argv)>2 else 120
ref.set_evp_parameters(3600.0,ND,False); ref.set_strength_parameters(1,0,0,4.0)
inject(ref,grid,s,dom); ref.evp(3600.0)
for opts in (dict(resident=2,resident_fold=1), dict(resident=0,resident_fold=0), dict(resident=0,resident_fold=0,derive_metrics=0)):
    sg={k:v.copy() for k,v in s.items()}
    ctx.evp_init(grid,ndte=ND,krdg_partic=0,krdg_redist=0)
    for k,v in opts.items(): ctx.evp_set_option(k,v)
    ctx.evp(3600.0,sg)
    out=[]
    for k in ("uvel","vvel","stressp_1","strength","strintx","iceumask"):
        w=ref.get(k); bad=np.argwhere(w!=sg[k])
        out.append("%s:%d%s" % (k,len(bad), (" rows %s cols %s max|d| %.3g" % (sorted(set(bad[:,1].tolist()))[:6], sorted(set(bad[:,2].tolist()))[:8], float(np.abs(w-sg[k]).max()))) if len(bad) else ""))
    print(ns, opts, "resident", ctx.evp_get_info("resident"), "waves", ctx.evp_get_info("waves"), ctx.evp_get_info("rows_per_wave"), " | ".join(out))
