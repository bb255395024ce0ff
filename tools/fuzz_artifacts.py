#!/usr/bin/env python3
"""remove_sdf_artifacts! on random fields (GPU vs CPU oracle): smoothed noise at random lattice sizes - thousands of
components of all sizes, ties in the largest size, components cut by the 64-voxel pieces of the labelling - with random
thresholds and size ratios; the flipped set and its count compared exactly.
  python tools/fuzz_artifacts.py [first_seed] [n_seeds]      (test infrastructure: uses oracle/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
oracle = graft.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    ext = rng.uniform(0.6, 2.0, 3)
    nmax = int(rng.integers(20, 130))
    pg = pkg.Grid(np.zeros(3), ext, nmax, 1)
    og = oracle.grid_make(np.zeros(3), ext, nmax, 1)
    nx, ny, nz = pg.dims
    f = rng.normal(size=(nz, ny, nx))
    for _ in range(int(rng.integers(0, 4))):           # smoothing passes: from voxel noise to blobs
        for ax in range(3):
            f = f + np.roll(f, 1, axis=ax)
    level = float(np.quantile(f, rng.uniform(0.3, 0.97)))
    sdf = (f - level).ravel()
    if seed % 5 == 0:
        sdf[rng.random(sdf.size) < 0.02] = 1e10         # sentinels of either sign
        sdf[rng.random(sdf.size) < 0.02] = -1e10
    ratio = float(rng.choice([0.0, 0.001, 0.01, 0.1, 0.5, 1.0]))
    thr = float(rng.choice([0.0, 0.0, 0.05 * np.abs(f).max()]))
    a, b = sdf.copy(), sdf.copy()
    na = pkg.remove_sdf_artifacts(a, pg, threshold=thr, min_component_ratio=ratio)
    nb = oracle.remove_artifacts(b, og, thr, ratio)
    neq = int((a != b).sum()) + (0 if na == nb else 1)
    bad += neq
    print(f"seed {seed}: lattice {nx}x{ny}x{nz} ratio {ratio} threshold {thr:.3g}: flipped {na} (oracle {nb})  differences: {neq}", flush=True)
print("TOTAL differences:", bad)
