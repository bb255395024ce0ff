#!/usr/bin/env python3
"""BASELINE config 5 (1M TET4, 1024^3 grid, 8 GPUs) as seen by ONE rank: the interleaved 1/8 share of the
1024^3 grid over the 998 250-tet Schlafli mesh.  Prints stage times and checks sampled planes against the oracle."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--check-planes", type=int, default=2)
a = ap.parse_args()
pkg = graft.load_built()
import torch
from rho2sdf_jl_amd import synthetic, slabs
X, IEN, rn = synthetic.tet_mesh(a.cells)
n_max = synthetic.grid_n_max_for_points(a.grid)
g = pkg.Grid(X.min(0), X.max(0), n_max, 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
owned, per = slabs.interleaved_layers(nz, a.world, a.rank)
out = torch.empty(owned * 4 * nx * ny, dtype=torch.float64, device=dev)
plan = pkg.DevicePlan(0)
for r in range(a.reps):
    t0 = time.perf_counter()
    st = plan.run(dX, dI, dR, 0.5, g, sdf=out, zstride=a.world, zphase=a.rank)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"elements": len(IEN), "grid": [nx, ny, nz], "rank_planes": owned * 4, "ms_call": dt * 1e3,
                  "Mvoxels_per_s_rank": owned * 4 * nx * ny / dt / 1e6,
                  **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}}))
if a.check_planes:
    O = graft.load_oracle()
    og = O.grid_make(X.min(0), X.max(0), n_max, 3)
    # planes of this rank: global k = (layer*world + rank)*4 + p ; sample two
    ks = [(0 * a.world + a.rank) * 4 + 1, ((owned // 2) * a.world + a.rank) * 4 + 2][:a.check_planes]
    got = out.view(owned * 4, ny, nx)
    bad = 0
    for k in ks:
        O.set_k_sampling(nz + 1, k)      # only plane k
        d, _, _ = O.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
        sgn = O.sign_detection(X, IEN, rn, 0.5, og)
        ref = (d * sgn).reshape(nz, ny, nx)[k]
        layer = (k // 4 - a.rank) // a.world
        mine = got[layer * 4 + (k & 3)].cpu().numpy()
        bad += int((mine != ref).sum())
        print("plane", k, "mismatches", int((mine != ref).sum()), "non-sentinel", int((np.abs(ref) < 1e9).sum()))
    O.set_k_sampling(1, 0)
    print("TOTAL mismatches", bad)
