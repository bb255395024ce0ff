#!/usr/bin/env python3
"""Per-phase visit / active-lane counts of iso_project_hex_pl_kernel on the NS workload.
Needs a diagnostic build (adds -DR2S_ISO_STATS to the hipcc line of __graft_entry__.build) loaded through
R2S_LIB_OVERRIDE=build_ab/stats.so python tools/iso_phase_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.build()
import torch
from rho2sdf_jl_amd import synthetic
X, IEN, rn = synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
out = torch.empty(g.ngp, dtype=torch.float64, device=dev)
plan = pkg.DevicePlan(0)
L = pkg._lib.lib()
buf = (ctypes.c_ulonglong * 32)()
L.r2s_debug_iso_stats(buf, 1)
st = plan.run(dX, dI, dR, 0.5, g, sdf=out)
torch.cuda.synchronize()
L.r2s_debug_iso_stats(buf, 1)
names = ["EVAL", "QP", "POST", "LS", "UPD", "trip", "finish"]
for i, n in enumerate(names):
    v, l = buf[2 * i], buf[2 * i + 1]
    print(f"{n:7s} visits {v:10d} lanes {l:12d} util {l / max(1, 64 * v):.3f}")
print("pairs (finished lanes)", buf[13], "chunks", st["n_iso_chunks"])
print("iterations/4 histogram", [buf[16 + i] for i in range(16)])
