#!/usr/bin/env python3
"""One rank's share under CONTIGUOUS Z-slabs whose cut planes balance the band work, against the interleaved default:
python tools/slab_share.py [--tet] [--world 8].  The cut planes come from a band-work histogram over the tile layers
(active band tiles per 4-plane layer of a full-volume run - the quantity a balanced partition would compute once on the
device); every rank's slab is timed on this one GPU, the slowest is what counts."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
ap = argparse.ArgumentParser()
ap.add_argument("--tet", action="store_true")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
pkg = graft.load_built()
import numpy as np, torch
from rho2sdf_jl_amd import synthetic, slabs
X, IEN, rn = synthetic.tet_mesh(55) if a.tet else synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(1024 if a.tet else 512), 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
plan = pkg.DevicePlan(0)


def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        st = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps * 1e3, st


full = torch.empty(g.ngp, dtype=torch.float64, device=dev)
t_full, st_full = timed(lambda: plan.run(dX, dI, dR, 0.5, g, sdf=full))
# band work per 4-plane tile layer: voxels that are not sentinels (a proxy for projection + gather work)
vol = full.view(nz // 4, 4 * ny * nx)
work = (vol.abs() < 1e9).sum(dim=1).double().cpu().numpy() + 0.02 * 4 * ny * nx   # + a share for the sweep / sign tiles
del full
cum = np.concatenate([[0.0], np.cumsum(work)])
cuts = [0] + [int(np.searchsorted(cum, cum[-1] * r / a.world)) for r in range(1, a.world)] + [nz // 4]
res = {"workload": "tet5" if a.tet else "ns", "world": a.world, "ms_full": round(t_full, 3), "layer_cuts": cuts}
shares = []
for r in range(a.world):
    k0, k1 = cuts[r] * 4, cuts[r + 1] * 4
    out = torch.empty((k1 - k0) * ny * nx, dtype=torch.float64, device=dev)
    t, st = timed(lambda: plan.run(dX, dI, dR, 0.5, g, sdf=out, k_begin=k0, k_end=k1))
    shares.append({"rank": r, "planes": k1 - k0, "ms": round(t, 3), **{k: round(v, 3) for k, v in st.items() if k.startswith("ms_")}})
    del out
res["balanced_contiguous"] = shares
res["balanced_contiguous_slowest_ms"] = max(s["ms"] for s in shares)
owned, _ = slabs.interleaved_layers(nz, a.world, 0)
out = torch.empty(owned * 4 * ny * nx, dtype=torch.float64, device=dev)
t, st = timed(lambda: plan.run(dX, dI, dR, 0.5, g, sdf=out, zstride=a.world, zphase=0))
res["interleaved_rank0"] = {"ms": round(t, 3), **{k: round(v, 3) for k, v in st.items() if k.startswith("ms_")}}
print(json.dumps(res))
