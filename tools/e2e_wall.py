#!/usr/bin/env python3
"""Where the wall time of one rho2sdf() call goes: the library's own stage clock vs the Python wrapper around it."""
import os, sys, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
from rho2sdf_jl_amd import synthetic
X, IEN, rho_n = synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
rho_e = np.ascontiguousarray(rho_n[IEN - 1].mean(axis=1))
opts = pkg.Rho2sdfOptions(threshold_density=0.5)
L = pkg._lib
real = L.lib().r2s_rho2sdf
for pinned in (False, True):
    for rep in range(3):
        info = {}
        t0 = time.perf_counter()
        r = pkg.rho2sdf("bench", X, IEN, rho_e, options=opts, sdf_grid=g, info=info, pinned_results=pinned)
        t1 = time.perf_counter()
        del r
        t2 = time.perf_counter()
        print(json.dumps({"pinned": pinned, "rep": rep, "wall_ms": round((t1 - t0) * 1e3, 1), "free_ms": round((t2 - t1) * 1e3, 1),
                          "lib_total": round(info["ms_total"], 1), "rbf": round(info["ms_rbf"], 1), "download": round(info["ms_download"], 1),
                          "artifacts": round(info["ms_artifacts"], 1)}))
