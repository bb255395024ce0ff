#!/usr/bin/env python3
"""On which NUMA node do the pages of the result arrays of the e2e legs live (pageable numpy array, array from
r2s_host_alloc), and how fast is a call into each?  (move_pages(2) with a null `nodes` argument only reports.)
  python tools/numa_where.py      (test infrastructure; run on the GPU box)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
from rho2sdf_jl_amd import synthetic
libc = ctypes.CDLL(None, use_errno=True)
SYS_move_pages = 279


def nodes_of(arr, samples=64):
    base = arr.ctypes.data
    step = max(4096, (arr.nbytes // samples) & ~4095) | 4096   # (an odd number of pages: interleaved pages are not all sampled on one node)
    addrs = [(base + i * step) & ~4095 for i in range(samples) if i * step < arr.nbytes]
    pages = (ctypes.c_void_p * len(addrs))(*addrs)
    status = (ctypes.c_int * len(addrs))()
    rc = libc.syscall(SYS_move_pages, 0, len(addrs), pages, None, status, 0)
    if rc != 0:
        return "move_pages failed (errno %d)" % ctypes.get_errno()
    out = {}
    for s in status:
        out[s] = out.get(s, 0) + 1
    return out


X, IEN, rn = synthetic.hex_mesh(46)
grid = pkg.Grid(X.min(0), X.max(0), 505, 3)
mesh = pkg.Mesh(X, IEN)
for kind in ("pageable", "pinned", "pageable", "pinned"):
    out = pkg.host_array(grid.ngp) if kind == "pinned" else np.empty(grid.ngp)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        pkg.sdf_fused(mesh, grid, rn, 0.5, device=0, out=out)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(kind, "ms per call", [round(t, 2) for t in ts], "pages on nodes", nodes_of(out), flush=True)
    del out
print("gpu numa nodes", [open(p).read().strip() for p in sorted(__import__("glob").glob("/sys/class/drm/card*/device/numa_node"))])
