#!/usr/bin/env python3
"""Wall time of every stage of rho2sdf() through the host-pointer C ABI (PCIe copies included)
on the synthetic HEX8 family: python tools/pipeline_stages.py --grid 256 --mesh 24 [--interp]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--mesh", type=int, default=24)
ap.add_argument("--interp", action="store_true")
ap.add_argument("--smooth", type=int, default=1)
a = ap.parse_args()
pkg = graft.load_built()
from rho2sdf_jl_amd import synthetic
X, IEN, rn0 = synthetic.hex_mesh(a.mesh)
# element densities from the nodal field (mean of the 8 nodes) so the pre-stage has real work
rho_e = rn0[IEN - 1].mean(axis=1)
mesh = pkg.Mesh(X, IEN)
grid = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(a.grid), 3)
T = {}
def timed(name, f):
    t = time.perf_counter(); r = f(); T[name] = round((time.perf_counter() - t) * 1e3, 2); return r
pkg.calculate_mesh_volume(mesh, rho_e)     # warm-up (context, module load)
vd, vf = timed("calculate_mesh_volume", lambda: pkg.calculate_mesh_volume(mesh, rho_e))
rn = timed("DenseInNodes", lambda: pkg.DenseInNodes(mesh, rho_e))
rt = timed("find_threshold_for_volume", lambda: pkg.find_threshold_for_volume(mesh, rn, vd * vf))
sdf = timed("sdf_fused (evalDistances+Sign_Detection)", lambda: pkg.sdf_fused(mesh, grid, rn, rt))
nf = timed("remove_sdf_artifacts", lambda: pkg.remove_sdf_artifacts(sdf, grid))
info = {}
fine = timed("RBFs_smoothing(%s, smooth=%d)" % ("interp" if a.interp else "approx", a.smooth),
             lambda: pkg.RBFs_smoothing(sdf, grid, a.interp, a.smooth, vd * vf, info=info))
print(json.dumps({"grid": grid.dims, "elements": int(len(IEN)), "rho_t": rt, "flipped": nf, "th": info["th"],
                  "cg_iterations": info["cg_iterations"], "ms": T}))

# whole rho2sdf() in ONE call (r2s_rho2sdf: every stage chained in HBM, results into pinned arrays)
opts = pkg.Rho2sdfOptions(threshold_density=rt, sdf_grid_setup="automatic", rbf_interp=a.interp,
                          rbf_grid="same" if a.smooth == 1 else "fine")
pkg.rho2sdf("bench", X, IEN, rho_e, options=opts, sdf_grid=grid)      # warm-up
for pinned in (True, False):
    info = {}
    t = time.perf_counter()
    pkg.rho2sdf("bench", X, IEN, rho_e, options=opts, sdf_grid=grid, info=info, pinned_results=pinned)
    wall = (time.perf_counter() - t) * 1e3
    print(json.dumps({"r2s_rho2sdf": "pinned results" if pinned else "pageable results", "wall_ms": round(wall, 1),
                      **{k: round(v, 2) for k, v in info.items() if k.startswith("ms_")}, "cg_iters": info["cg_iters"]}))
