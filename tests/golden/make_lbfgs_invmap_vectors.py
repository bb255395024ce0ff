#!/usr/bin/env python3
"""Golden vectors for find_local_coordinates (HEX8) from an INDEPENDENT bounded quasi-Newton solver: scipy's
L-BFGS-B with the reference's problem statement (src/SignedDistances/FindLocalCoordinates.jl:27-104): objective
|Xe N(xi) - x|^2, bounds +-1.1, the nine starts (centre + eight at +-0.5), 5 stored corrections, best successful
start wins.  (NLopt's LD_LBFGS is Luksan's code, scipy's is Nocedal's: same family, different line search; both are
run to convergence here - ftol / gtol far below the reference's 1e-8 / 1e-10 - so the vectors are the roots
themselves, not one optimiser's stopping point.)

Cases: strongly distorted hexahedra (node jitter up to 35 % of the edge, the family of
tests/test_parity_gpu.py::test_distorted_hex_random_density) with points in and around the element.

Run in the build container (needs scipy); writes tests/golden/lbfgs_invmap.npz:
  Xe (n,8,3), x (n,3), found (n,), xi (n,3) best minimiser, fmin (n,) objective there."""
import os
import numpy as np
from scipy.optimize import minimize

S = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
STARTS = [(0.0, 0.0, 0.0)] + [tuple(0.5 * s) for s in S]        # FindLocalCoordinates.jl:27-37


def shape(xi):
    return 0.125 * np.prod(1 + S * xi, axis=1)


def dshape(xi):
    t = 1 + S * xi
    d = np.empty((8, 3))
    d[:, 0] = 0.125 * S[:, 0] * t[:, 1] * t[:, 2]
    d[:, 1] = 0.125 * S[:, 1] * t[:, 0] * t[:, 2]
    d[:, 2] = 0.125 * S[:, 2] * t[:, 0] * t[:, 1]
    return d


def solve(Xe, x):
    def fg(xi):
        R = Xe.T @ shape(xi) - x
        return R @ R, 2.0 * (Xe.T @ dshape(xi)).T @ R
    best, bx, found = np.inf, np.full(3, 10.0), False
    for s in STARTS:
        r = minimize(fg, np.array(s), jac=True, method="L-BFGS-B", bounds=[(-1.1, 1.1)] * 3,
                     options={"maxcor": 5, "ftol": 1e-30, "gtol": 1e-14, "maxfun": 2000, "maxiter": 2000})
        if r.status in (0, 1, 2) and r.fun < best:     # converged / limits / abnormal line-search end near the optimum
            best, bx, found = r.fun, r.x.copy(), True
    return found, bx, best


def main():
    rng = np.random.default_rng(20240502)
    out = {k: [] for k in ("Xe", "x", "found", "xi", "fmin")}
    for case in range(3000):
        jit = rng.choice([0.15, 0.25, 0.30, 0.35])
        Xe = 0.5 * S + rng.uniform(-jit, jit, (8, 3))
        mn, mx = Xe.min(0), Xe.max(0)
        if case % 3 == 0:       # points the kernels really test: the element's AABB (SignDetection.jl:30)
            x = rng.uniform(mn, mx)
        elif case % 3 == 1:     # near the faces: images of points with one |xi| close to 1
            xi = rng.uniform(-1, 1, 3)
            xi[rng.integers(3)] = rng.choice([-1, 1]) * rng.uniform(0.9, 1.12)
            x = Xe.T @ shape(xi)
        else:                   # a shell around the element
            x = rng.uniform(mn - 0.2, mx + 0.2)
        found, xi, f = solve(Xe, x)
        out["Xe"].append(Xe); out["x"].append(x); out["found"].append(found); out["xi"].append(xi); out["fmin"].append(f)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lbfgs_invmap.npz"),
                        **{k: np.array(v) for k, v in out.items()})
    print("wrote", len(out["x"]), "cases")


if __name__ == "__main__":
    main()
