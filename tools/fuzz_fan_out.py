#!/usr/bin/env python3
"""One call over G devices (r2s_options.n_gpus: interleaved tile layers, Z-slab post-processing with interface merge, halo
exchange and plane-wise sums) against the same call on one device, on random meshes, grids, device counts and options -
raw SDF, cleaned field, iteration count, level and smoothed field compared bit for bit.  On a one-GPU box the G logical
devices are mapped onto the one device (R2S_MULTI_OVERSUBSCRIBE): the partition, the threads and the copies are those of a
multi-GPU node.   python tools/fuzz_fan_out.py [first_seed] [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["R2S_MULTI_OVERSUBSCRIBE"] = "1"
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
from rho2sdf_jl_amd import synthetic
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 9))
    X, IEN, rn = synthetic.hex_mesh(n, jitter=float(rng.uniform(0.05, 0.3)), seed=seed)
    tets = seed % 4 == 3
    if tets:
        IEN = synthetic.hex_to_tets(IEN)
    npts = int(rng.integers(36, 90))
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(npts), 3)
    G = int(rng.choice([2, 3, 4, 5, 8]))
    rho = np.clip(rn[IEN - 1].mean(axis=1) + rng.normal(0, float(rng.uniform(0.0, 0.4)), len(IEN)), 0, 1)
    interp = bool(rng.integers(0, 2))
    smooth = int(rng.choice([1, 1, 2]))
    opts = pkg.Rho2sdfOptions(threshold_density=0.5, rbf_interp=interp, rbf_grid="same" if smooth == 1 else "fine",
                              artifact_min_component_ratio=float(rng.choice([0.01, 0.05, 0.3])))
    ia, ib = {}, {}
    a = pkg.rho2sdf("t", X, IEN, rho, options=opts, sdf_grid=pg, info=ia)
    b = pkg.rho2sdf("t", X, IEN, rho, options=opts, sdf_grid=pg, n_gpus=G, info=ib)
    neq = int((a[3] != b[3]).sum()) + int((a[0] != b[0]).sum()) + (ia["n_flipped"] != ib["n_flipped"]) + (ia["cg_iters"] != ib["cg_iters"]) + \
        (ia["level_shift"] != ib["level_shift"])
    bad += neq
    if neq:   # where: which array, how many, the first few
        for name, u, v in (("sdf_dists", a[3], b[3]), ("fine_sdf", a[0], b[0])):
            d = np.flatnonzero(np.asarray(u).ravel() != np.asarray(v).ravel())
            if d.size:
                print(f"   {name}: {d.size} of {np.asarray(u).size} differ; first {d[:4]}, one device {np.asarray(u).ravel()[d[:4]]}, {G} devices {np.asarray(v).ravel()[d[:4]]}")
    print(f"seed {seed}: {'tet4' if tets else 'hex8'} mesh {n}^3, grid {pg.dims}, G = {G}, interp {int(interp)} smooth {smooth}: flipped {ia['n_flipped']}, "
          f"CG {ia['cg_iters']}, level {ia['level_shift']:.9g}: differences {neq}", flush=True)
    if seed % 8 == 7:
        pkg._lib.lib().r2s_release_cache()
print("TOTAL differences:", bad)
