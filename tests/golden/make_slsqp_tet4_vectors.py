#!/usr/bin/env python3
"""Golden vectors for compute_coords_on_iso, TET4 (src/SignedDistances/ComputeCoordsOnIso.jl:90-181) from scipy's
SLSQP with the reference's problem statement: variables lambda_1..3 in [0,1], equality N(lambda).rho_e = rho_t,
inequality sum(lambda) <= 1, start (0.25, 0.25, 0.25), objective |x - Xe N(lambda)|^2 with N = (l1, l2, l3, 1 - sum).
The map is affine and the density linear, so the problem is a strictly convex QP with a unique minimiser (the
closest point of the planar polygon {rho = rho_t} in the tetrahedron) - the library's closed form must reproduce it.

Writes tests/golden/slsqp_tet4_projection.npz: x (n,3), Xe (n,4,3), re (n,4), rt (n,), lam (n,3), dist (n,).
Cases: jittered Schlafli tetrahedra of a unit cube (the config-5 family) with points inside and around them."""
import os
import numpy as np
from scipy.optimize import minimize

SCHLAFLI = np.array([[0, 1, 2, 6], [0, 5, 1, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 4, 5, 6], [0, 7, 4, 6]])
S = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)


def N(l):
    return np.array([l[0], l[1], l[2], 1.0 - l.sum()])


DN = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1]], float)


def solve(x, Xe, re, rt):
    f = lambda l: np.sum((x - Xe.T @ N(l)) ** 2)
    g = lambda l: -2 * (Xe.T @ DN).T @ (x - Xe.T @ N(l))
    cons = [{"type": "eq", "fun": lambda l: re @ N(l) - rt, "jac": lambda l: DN.T @ re},
            {"type": "ineq", "fun": lambda l: 1.0 - l.sum(), "jac": lambda l: -np.ones(3)}]
    return minimize(f, np.full(3, 0.25), jac=g, bounds=[(0, 1)] * 3, constraints=cons, method="SLSQP",
                    options={"ftol": 1e-16, "maxiter": 500})


def main():
    rng = np.random.default_rng(20240503)
    out = {k: [] for k in ("x", "Xe", "re", "rt", "lam", "dist")}
    while len(out["x"]) < 600:
        H = 0.5 * S + rng.uniform(-0.15, 0.15, (8, 3))
        Xe = H[SCHLAFLI[rng.integers(6)]]
        re = np.clip(rng.normal(0.5, 0.3, 4), 0, 1)
        rt = 0.5
        if not (re.min() < rt < re.max()):
            continue
        x = rng.uniform(Xe.min(0) - 0.3, Xe.max(0) + 0.3)
        r = solve(x, Xe, re, rt)
        lam = r.x
        if r.status != 0 or abs(re @ N(lam) - rt) > 1e-12 or lam.min() < -1e-12 or lam.sum() > 1 + 1e-12:
            continue
        out["x"].append(x); out["Xe"].append(Xe); out["re"].append(re); out["rt"].append(rt)
        out["lam"].append(lam); out["dist"].append(np.linalg.norm(x - Xe.T @ N(lam)))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "slsqp_tet4_projection.npz"),
                        **{k: np.array(v) for k, v in out.items()})
    print("wrote", len(out["x"]), "cases")


if __name__ == "__main__":
    main()
