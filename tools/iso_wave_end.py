#!/usr/bin/env python3
"""When the wavefronts of iso_project_hex_pl_kernel leave (tail of the persistent kernel) on one rank's share of the
NS workload.  Needs a diagnostic build with -DR2S_ISO_WAVE_END:
R2S_LIB_OVERRIDE=build_ab/wend.so WORLDS=1,8 python tools/iso_wave_end.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_built()
import numpy as np
import torch
from rho2sdf_jl_amd import synthetic, slabs
X, IEN, rn = synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
plan = pkg.DevicePlan(0)
L = pkg._lib.lib()
for world in [int(w) for w in os.environ.get("WORLDS", "1,8").split(",")]:
    owned, per = slabs.interleaved_layers(nz, world, 0)
    local = torch.empty(4 * owned * nx * ny, dtype=torch.float64, device=dev)
    for _ in range(4):
        st = plan.run(dX, dI, dR, 0.5, g, sdf=local, zstride=world, zphase=0)
    torch.cuda.synchronize()
    we = (ctypes.c_ulonglong * 4096)()
    L.r2s_debug_iso_wave_end(we)
    t = np.array(we[:], dtype=np.float64)
    ends = np.sort(t[:3072][t[:3072] > 0]) - t[4094]
    ends *= 1e-2   # 100 MHz counter -> microseconds
    q = lambda f: ends[min(len(ends) - 1, int(len(ends) * f))]
    if hasattr(L, "r2s_debug_iso_wave_info"):
        wi = (ctypes.c_ulonglong * (4096 * 4))()
        L.r2s_debug_iso_wave_info(wi)
        info = np.array(wi[:], dtype=np.float64).reshape(4096, 4)[:3072]
        tt = t[:3072] - t[4094]
        last = np.argsort(tt)[-12:]
        dry = (info[:, 0] - t[4094]) * 1e-2
        print("work ran dry for the wavefronts (us after the start): first %.0f, median %.0f, last %.0f" % (dry[dry > 0].min(), np.median(dry[dry > 0]), dry.max()))
        for w in last:
            el = None
            print("  wave %4d exits at %.0f us; work dry for it at %.0f us with %d lanes busy, last item %d" % (w, tt[w] * 1e-2, dry[w], info[w, 2], info[w, 1]))
    print("world %d: kernel %.0f us by events; wavefront exits after the start (us): first %.0f, 10%% %.0f, 50%% %.0f, "
          "90%% %.0f, 99%% %.0f, last %.0f" % (world, st["ms_iso_fast"] * 1e3, ends[0], q(0.1), q(0.5), q(0.9), q(0.99), ends[-1]))
