#!/usr/bin/env python3
"""Golden vectors for the iso-surface projection (compute_coords_on_iso, HEX8) from an INDEPENDENT
implementation of the reference's optimiser family: scipy.optimize SLSQP (Kraft's SLSQP, the code NLopt's
LD_SLSQP is translated from), same objective / constraint / bounds / start as
src/SignedDistances/ComputeCoordsOnIso.jl:19-86, converged tightly (ftol 1e-15).

Run in the build container (needs scipy); writes tests/golden/slsqp_iso_projection.npz:
  x (n,3), Xe (n,8,3), re (n,8), rt (n,), xi (n,3) SLSQP minimiser, dist (n,) = |x - Xe N(xi)|
Cases: jittered unit hexes (+-0.15 h) with smooth radial density fields (the north-star family)."""
import os
import numpy as np
from scipy.optimize import minimize

S = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)


def shape(xi):
    return 0.125 * np.prod(1 + S * xi, axis=1)


def dshape(xi):
    t = 1 + S * xi
    d = np.empty((8, 3))
    d[:, 0] = 0.125 * S[:, 0] * t[:, 1] * t[:, 2]
    d[:, 1] = 0.125 * S[:, 1] * t[:, 0] * t[:, 2]
    d[:, 2] = 0.125 * S[:, 2] * t[:, 0] * t[:, 1]
    return d


def slsqp(x, Xe, re, rt):
    f = lambda xi: np.sum((x - Xe.T @ shape(xi)) ** 2)
    g = lambda xi: -2 * (Xe.T @ dshape(xi)).T @ (x - Xe.T @ shape(xi))
    c = lambda xi: re @ shape(xi) - rt
    cg = lambda xi: dshape(xi).T @ re
    r = minimize(f, np.zeros(3), jac=g, bounds=[(-1, 1)] * 3, constraints=[{"type": "eq", "fun": c, "jac": cg}],
                 method="SLSQP", options={"ftol": 1e-15, "maxiter": 1000})
    return r


def main():
    rng = np.random.default_rng(20240501)
    out = {k: [] for k in ("x", "Xe", "re", "rt", "xi", "dist")}
    while len(out["x"]) < 400:
        Xe = 0.5 * S + rng.uniform(-0.15, 0.15, (8, 3))
        ctr = rng.uniform(-3, 3, 3)
        R = np.linalg.norm(ctr) + rng.uniform(-0.4, 0.4)
        re = np.clip(0.5 + (R - np.linalg.norm(Xe - ctr, axis=1)) / 2.0, 0, 1)
        rt = 0.5
        if not (re.min() < rt < re.max()):
            continue
        x = rng.uniform(-0.8, 0.8, 3)
        r = slsqp(x, Xe, re, rt)
        if r.status != 0 or abs(re @ shape(r.x) - rt) > 1e-12:
            continue
        out["x"].append(x); out["Xe"].append(Xe); out["re"].append(re); out["rt"].append(rt)
        out["xi"].append(r.x); out["dist"].append(np.linalg.norm(x - Xe.T @ shape(r.x)))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "slsqp_iso_projection.npz"),
                        **{k: np.array(v) for k, v in out.items()})
    print("wrote", len(out["x"]), "cases")


if __name__ == "__main__":
    main()
