#!/usr/bin/env python3
"""What ONE rank of an N-GPU run does on the NS workload, timed piece by piece on one GPU:
its share of the grid (interleaved tile layers), the packing of its tiles, and the scatter of all ranks' tiles
(every rank's payload is taken to be as large as rank 0's).  The all-gather itself is not measured here."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_built()
import torch
from rho2sdf_jl_amd import synthetic, slabs
X, IEN, rn = synthetic.hex_mesh(46)
g = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(512), 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
plan = pkg.DevicePlan(0)
full = torch.empty(g.ngp, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for world in [int(w) for w in os.environ.get('WORLDS', '1,2,4,8').split(',')]:
    owned, per = slabs.interleaved_layers(nz, world, 0)
    local = torch.empty(4 * owned * nx * ny, dtype=torch.float64, device=dev)
    st = {}
    def run():
        st.update(plan.run(dX, dI, dR, 0.5, g, sdf=local, zstride=world, zphase=0))
    t_run = timed(run)
    nf, nm = int(st["n_active_tiles"]), int(st["n_sign_only_tiles"])
    seglen = nf * 64 + (nf + 1) // 2 + nm + (nm + 1) // 2
    seg = torch.empty(seglen, dtype=torch.float64, device=dev)
    payload, ids, masks, mids = slabs.SlabGather._segment_views(seg, nf, nm)
    t_pack = timed(lambda: plan.pack_tiles2(local, payload, ids, masks, mids))
    t_fill = timed(lambda: plan.fill(full, -1e10))
    def scatter():
        for r in range(world):
            plan.unpack_tiles(payload, ids, nf, g, full)
            plan.unpack_masks(masks, mids, nm, g, full)
    t_unpack = timed(scatter)
    print(json.dumps({"world": world, "ms_run": round(t_run, 3), "stages": {k: round(v, 3) for k, v in st.items() if k.startswith("ms_")},
                      "band_tiles": nf, "sign_only_tiles": nm, "segment_MB": round(seglen * 8 / 1e6, 2),
                      "gathered_MB": round(world * seglen * 8 / 1e6, 2), "ms_pack": round(t_pack, 3),
                      "ms_fill": round(t_fill, 3), "ms_unpack_all": round(t_unpack, 3)}))
