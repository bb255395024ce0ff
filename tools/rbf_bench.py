#!/usr/bin/env python3
"""Time of RBFs_smoothing alone (r2s_rbf_smooth_dev, device-resident in and out) on a synthetic banded SDF:
sphere of radius 0.7 in [-1,1]^3, real distances inside a band of 12 cells, +-1e10 elsewhere (like the raw SDF).
  python tools/rbf_bench.py [--grid 512] [--interp] [--smooth 1] [--reps 3]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--interp", action="store_true")
ap.add_argument("--smooth", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
pkg = graft.load_built()
import torch
L = pkg._lib
g = pkg.Grid(np.full(3, -1.0), np.full(3, 1.0), a.grid - 7, 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
ax = [torch.tensor(g.AABB_min[i] + g.cell_size * np.arange(n), device=dev) for i, n in enumerate((nx, ny, nz))]
r = torch.sqrt(ax[2][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[0][None, None, :] ** 2)
sdf = 0.7 - r
band = 12 * g.cell_size
sdf = torch.where(sdf.abs() < band, sdf, torch.sign(sdf) * 1e10).contiguous().view(-1)
del r
dims = tuple(int(n) * a.smooth + 1 for n in g.c.N)
fine = torch.empty(dims[0] * dims[1] * dims[2], dtype=torch.float32, device=dev)
target = 4.0 / 3.0 * np.pi * 0.7 ** 3
th, its = ctypes.c_float(), ctypes.c_int32()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
times = []
for rep in range(a.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.check(L.lib().r2s_rbf_smooth_dev(ctypes.c_void_p(sdf.data_ptr()), ctypes.byref(g.c), int(a.interp), a.smooth, 1e-3,
                                       float(target), ctypes.c_void_p(fine.data_ptr()), ctypes.byref(th), ctypes.byref(its), stream))
    times.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"grid": [nx, ny, nz], "interp": a.interp, "smooth": a.smooth, "cg_iters": its.value, "th": th.value,
                  "ms_first": round(times[0], 1), "ms": [round(t, 1) for t in times[1:]],
                  "matvec": os.environ.get("R2S_RBF_MATVEC", "lut"), "checksum": float(fine.double().sum().item())}))
