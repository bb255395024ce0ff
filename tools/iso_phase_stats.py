#!/usr/bin/env python3
"""Per-phase visit / active-lane counts of iso_project_hex_pl_kernel on the NS workload.
Needs a diagnostic build (adds -DR2S_ISO_STATS to the hipcc line of __graft_entry__.build) loaded through
R2S_LIB_OVERRIDE=diag/stats.so python tools/iso_phase_stats.py  (tools/build_diag.sh builds it)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_built()
import torch
from rho2sdf_jl_amd import synthetic
X, IEN, rn = synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
WORLD = int(os.environ.get("WORLD", "1"))   # WORLD=8: the share of rank 0 of 8 (interleaved tile layers)
out = torch.empty(g.ngp, dtype=torch.float64, device=dev)
plan = pkg.DevicePlan(0)
L = pkg._lib.lib()
buf = (ctypes.c_ulonglong * 56)()
L.r2s_debug_iso_stats(buf, 1)
lng = (ctypes.c_ulonglong * 512)()
for _ in range(3):   # the last run is the one reported (allocations and first touches are behind it)
    L.r2s_debug_iso_stats(buf, 1)
    L.r2s_debug_iso_long(lng)
    st = plan.run(dX, dI, dR, 0.5, g, sdf=out, zstride=WORLD, zphase=0) if WORLD > 1 else plan.run(dX, dI, dR, 0.5, g, sdf=out)
    torch.cuda.synchronize()
L.r2s_debug_iso_stats(buf, 1)
names = ["EVAL", "QP", "FINISH", "bail", "-", "trip", "done", "QP2"]
for i, n in enumerate(names):
    if n == "-":
        continue
    v, l = buf[2 * i], buf[2 * i + 1]
    print(f"{n:7s} visits {v:10d} lanes {l:12d} util {l / max(1, 64 * v):.3f}")
print("handed over to the complete solver: %d of %d pairs (%.2f %%)" % (buf[7], buf[13] + buf[7], 100.0 * buf[7] / max(1, buf[13] + buf[7])))
print("pairs (finished lanes)", buf[13], "chunks", st["n_iso_chunks"], "items", st["n_items"], "stage ms", {k: round(v, 3) for k, v in st.items() if k.startswith("ms_")})
print("iterations/4 histogram", [buf[16 + i] for i in range(16)])
print("log2(trips per pair) histogram", [buf[32 + i] for i in range(16)])
print("pairs with >= 128 trips: trips %d, QP visits %d (second visits of a trip not counted), LS visits %d, SQP iterations %d" % (buf[48], buf[49], buf[50], buf[51]))
import struct
L.r2s_debug_iso_long(lng)
print("pairs with >= 256 trips:", lng[0])
for k in range(min(int(lng[0]), 63)):
    w = lng[8 * (k + 1): 8 * (k + 2)]
    xyz = [struct.unpack("d", struct.pack("Q", v))[0] for v in w[1:4]]
    re = rn[IEN[w[0]] - 1]
    dmin = abs(re - 0.5).min()
    print("LONG el %d x %.6g %.6g %.6g its %d trips %d qp %d ls %d | hard-flag %s dmin/range %.3f" % (
        w[0], xyz[0], xyz[1], xyz[2], w[4], w[5], w[6], w[7], dmin < 0.1 * (re.max() - re.min()), dmin / (re.max() - re.min())))
