#!/usr/bin/env python3
"""Stage timing of the hot path per output mode (dist / sign / fused) on the NS workload (or --tet: config 5)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--mesh", type=int, default=46)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--modes", default="dist,sign,sdf")
ap.add_argument("--tet", action="store_true", help="config 5: 6 x mesh^3 TET4 (defaults 55, 1024^3)")
a = ap.parse_args()
pkg = graft.load_built()
import torch
from rho2sdf_jl_amd import synthetic
if a.tet:
    a.mesh = 55 if a.mesh == 46 else a.mesh
    a.grid = 1024 if a.grid == 512 else a.grid
X, IEN, rn = synthetic.tet_mesh(a.mesh) if a.tet else synthetic.hex_mesh(a.mesh)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(a.grid), 3)
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
out = torch.empty(g.ngp, dtype=torch.float64, device=dev)
plan = pkg.DevicePlan(0)
for mode in a.modes.split(","):
    import time
    st = plan.run(dX, dI, dR, 0.5, g, **{mode: out})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(a.reps):
        st = plan.run(dX, dI, dR, 0.5, g, **{mode: out})
    torch.cuda.synchronize()
    st["ms_wall_avg"] = (time.perf_counter() - t0) / a.reps * 1e3
    print(mode, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
