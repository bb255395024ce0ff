#!/usr/bin/env python3
"""Per-wavefront duration / iteration counts of iso_straggler_kernel on the NS workload (diagnostic build:
tools/build_diag.sh strag -DR2S_STRAG_DIAG=1; R2S_LIB_OVERRIDE=diag/strag.so python tools/strag_diag.py).
A -DR2S_STRAG_DIAG=2 build also fills the per-pair histograms (iterations, trips) - its times are distorted by the
histogram's same-address atomics: pass `hist` to print them and leave the times out."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_built()
import numpy as np, torch
from rho2sdf_jl_amd import synthetic
HIST = "hist" in sys.argv[1:]
if "chapadlo256" in sys.argv[1:]:      # BASELINE config 4 on the reference's 87 x 166 x 257 grid
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "chapadlo.npz"))
    X, IEN = d["X"], d["IEN"].astype(np.int64)
    rn = pkg.DenseInNodes(pkg.Mesh(X, IEN), d["rho"], device=0)
    g = pkg.Grid(X.min(0), X.max(0), 249, 3)
else:
    X, IEN, rn = synthetic.hex_mesh(46)
    g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
out = torch.empty(g.ngp, dtype=torch.float64, device=dev)
plan = pkg.DevicePlan(0)
L = pkg._lib.lib()
buf = (ctypes.c_ulonglong * (3 * 16384))()
hb = (ctypes.c_ulonglong * 256)()
for _ in range(3):
    L.r2s_debug_strag_diag(buf)
    L.r2s_debug_strag_hist(hb)
    plan.run(dX, dI, dR, 0.5, g, sdf=out)
    torch.cuda.synchronize()
L.r2s_debug_strag_diag(buf)
if not HIST:
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
    act = a[a[:, 2] > 0]
    print("wavefronts with work", len(act), "pairs in all", act[:, 2].sum(), "trips in all", act[:, 1].sum())
    cyc = act[:, 0] / 100.0   # wall_clock64 ticks at 100 MHz -> microseconds
    print("wavefront duration us: mean %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max()))
    print("trips per wavefront: mean %.1f max %d;  us per trip: median %.2f;  pairs per wavefront: mean %.1f max %d" % (
        act[:, 1].mean(), act[:, 1].max(), np.median(cyc / np.maximum(act[:, 1], 1)), act[:, 2].mean(), act[:, 2].max()))
    order = np.argsort(-cyc)[:8]
    print("slowest wavefronts (us, trips, pairs):", [(round(float(cyc[i]), 1), int(act[i, 1]), int(act[i, 2])) for i in order])
else:
    L.r2s_debug_strag_hist(hb)
    h = np.frombuffer(hb, dtype=np.uint64).astype(np.int64)
    def show(name, v):
        nz = np.nonzero(v)[0]
        print(name, "max", int(nz.max()) if len(nz) else 0, "hist by 8:", [int(v[i:i + 8].sum()) for i in range(0, 128, 8)])
    show("SQP iterations of a pair at its end (last 1 run):", h[:128])
    show("trips of a pair:", h[128:])
