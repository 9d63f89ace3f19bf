"""Latency of the reference-signature entry cice_thermo_vertical (host arrays in and out) on a gx1-size block:
what one `call thermo_vertical(...)` of the Fortran drop-in costs, transfers included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cice4_amd import lib, synth
ctx = lib.Context(); ctx.sync(); ctx.thermo_init()
ny, nx = 386, 322
a, icells, ii, jj = synth.thermo_columns(ny, nx, 2, regime="mixed", ice_frac=1.0, coherent=24)
for rep in range(3):
    b = {k: v.copy() for k, v in a.items()}
    t = time.perf_counter()
    st = ctx.thermo_vertical(3600.0, icells, ii, jj, b, yday=150.0)
    dt = time.perf_counter() - t
    print("call", rep, "ms", round(1e3 * dt, 3), "columns", icells, st)
