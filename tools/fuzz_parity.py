#!/usr/bin/env python3
"""GPU vs CPU oracle on random distorted HEX8 meshes with random nodal densities (the inputs that send most pairs through
the complete solver: non-convex models, corrections, restorations, pattern searches), every voxel compared bit for bit.
  python tools/fuzz_parity.py [first_seed] [n_seeds] [tet]  (test infrastructure: uses oracle/; `tet`: the same meshes split into
                                                            six TET4 each - the closed-form projection path of BASELINE config 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
oracle = graft.load_oracle()
from rho2sdf_jl_amd import synthetic
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tets = len(sys.argv) > 3 and sys.argv[3] == "tet"
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 9))
    jit = float(rng.uniform(0.1, 0.38))
    X, IEN, _ = synthetic.hex_mesh(n, jitter=jit, seed=seed)
    if tets:
        IEN = synthetic.hex_to_tets(IEN)
    kind = seed % 3
    if kind == 0:
        rn = np.clip(rng.normal(0.5, 0.35, len(X)), 0, 1)
    elif kind == 1:
        rn = (rng.random(len(X)) < 0.5).astype(float)                   # 0/1 densities (iso-surfaces on element faces)
    else:
        rn = np.clip(0.5 + 0.02 * rng.normal(size=len(X)), 0, 1)          # nearly flat field around the threshold
    npts = int(rng.integers(40, 72))
    nmax = synthetic.grid_n_max_for_points(npts)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    st = {}
    sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, stats=st)
    odist, _, ost = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    osign = oracle.sign_detection(X, IEN, rn, 0.5, og)
    want = odist * osign
    neq = int((sdf != want).sum())
    bad += neq
    print(f"seed {seed}{' tet4' if tets else ''}: mesh {n}^3 jitter {jit:.2f} density kind {kind} grid {npts}^3: pairs {ost['n_iso_solves']} "
          f"failed solves {ost['n_iso_fail']}  voxels not bit-equal: {neq}", flush=True)
print("TOTAL voxels not bit-equal:", bad)
sys.exit(1 if bad else 0)
