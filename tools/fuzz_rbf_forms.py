#!/usr/bin/env python3
"""RBFs_smoothing: the table-driven kernels of round 4 (row walk for the CG product and the same-grid evaluations, tables per
parity class for refined output grids) against neighbour-by-neighbour evaluation (R2S_RBF_MATVEC=fly R2S_RBF_APPLY=fly) on
random lattices, fields and options - weights (through the iteration count), level, LSF and output field compared bit for
bit.  GPU only (no oracle: the two forms must agree exactly).
  python tools/fuzz_rbf_forms.py [first_seed] [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_built()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    dims = tuple(int(v) for v in rng.integers(4, 90, 3))
    if seed % 7 == 0:
        dims = (int(rng.integers(120, 330)), int(rng.integers(4, 12)), int(rng.integers(4, 12)))
    smooth = int(rng.choice([1, 1, 2, 2, 3]))
    interp = bool(rng.integers(0, 2))
    h = float(rng.choice([0.125, 0.0625, 0.3, 1.7]))
    lo = rng.uniform(-3, 3, 3)
    g = pkg.Grid(lo, lo + h * (np.array(dims) - 1.0) + 1e-9, max(dims) - 1, 0)
    nx, ny, nz = g.dims
    ax = [g.AABB_min[i] + g.cell_size * np.arange(n) for i, n in enumerate((nx, ny, nz))]
    c = [a[0] + rng.uniform(0.2, 0.8) * (a[-1] - a[0]) for a in ax]
    r = np.sqrt((ax[0][None, None, :] - c[0]) ** 2 + (ax[1][None, :, None] - c[1]) ** 2 + (ax[2][:, None, None] - c[2]) ** 2)
    sdf = rng.uniform(0.15, 0.45) * g.cell_size * max(nx, ny, nz) - r + 0.05 * g.cell_size * rng.normal(size=r.shape)
    if seed % 3 == 0:
        sdf = np.where(np.abs(sdf) < 3 * g.cell_size, sdf, np.sign(sdf) * 1e10)
    sdf = sdf.ravel()
    target = max(float((sdf > 0).sum()), 1.0) * g.cell_size ** 3 * rng.uniform(0.5, 1.5)
    outs = {}
    for mode in ("tables", "fly"):
        for name in ("R2S_RBF_MATVEC", "R2S_RBF_APPLY"):
            if mode == "fly":
                os.environ[name] = "fly"
            else:
                os.environ.pop(name, None)
        info = {}
        fine = pkg.RBFs_smoothing(sdf, g, interp, smooth, target, info=info)
        outs[mode] = (fine, info["th"], info["lsf"], info["cg_iterations"])
    a, b = outs["tables"], outs["fly"]
    neq = int((a[0] != b[0]).sum()) + int((a[2] != b[2]).sum()) + (a[1] != b[1]) + (a[3] != b[3])
    bad += neq
    print(f"seed {seed}: lattice {nx}x{ny}x{nz} smooth {smooth} interp {int(interp)} cell {g.cell_size:.4g}: CG {a[3]} / {b[3]} iterations, "
          f"level {a[1]:.9g}: differences {neq}", flush=True)
    if seed % 16 == 0:
        pkg._lib.lib().r2s_release_cache()
print("TOTAL differences:", bad)
