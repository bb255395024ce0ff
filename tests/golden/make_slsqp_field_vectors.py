#!/usr/bin/env python3
"""Golden FIELDS for evalDistances on the reference's own fixtures with the iso-surface projections done by an
INDEPENDENT SLSQP (scipy's Kraft SLSQP - the code NLopt's LD_SLSQP is a translation of) instead of the oracle's
SQP restatement: every (iso element, band voxel) pair the oracle visits (its pair log, processing order) is solved
with the reference's problem statement (src/SignedDistances/ComputeCoordsOnIso.jl:19-86: objective, equality
constraint, bounds +-1, start 0) and fed back through the oracle's override hook, so that triangles, update rules
and order stay the reference's.  Two solver settings:
  ref   acc = 1e-5   the reference's own tolerances (ftol_rel = ftol_abs = xtol_rel = 1e-5)
  tight acc = 1e-15  converged
Like the reference (ComputeCoordsOnIso.jl:79-86) the result of a solve is used whatever its status.

Fixtures: the reference's 1hex_el input (test/runtests.jl:51-86, N = 15), cantilever_beam_vfrac_03, chapadlo.
Writes tests/golden/slsqp_fields.npz: <fixture>_{ref,tight} = dist[ngp] (1e10 = untouched), <fixture>_pairs = number
of pairs, <fixture>_fail_{ref,tight} = solves that did not report success, <fixture>_{ref,tight}_offsurface = voxels whose
value is the distance to an SLSQP result that does not satisfy the constraint.  Needs scipy; ~15 min on 8 cores."""
import multiprocessing as mp
import os
import sys

import numpy as np
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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


def slsqp(x, Xe, re, rt, acc):
    f = lambda xi: np.sum((x - Xe.T @ shape(xi)) ** 2)
    g = lambda xi: -2 * (Xe.T @ dshape(xi)).T @ (x - Xe.T @ shape(xi))
    c = lambda xi: re @ shape(xi) - rt
    cg = lambda xi: dshape(xi).T @ re
    r = minimize(f, np.zeros(3), jac=g, bounds=[(-1, 1)] * 3, constraints=[{"type": "eq", "fun": c, "jac": cg}],
                 method="SLSQP", options={"ftol": acc, "maxiter": 1000})
    return np.clip(r.x, -1.0, 1.0), r.status == 0


def work(args):
    pts, Xes, res, rt, acc = args
    out = np.empty((len(pts), 3))
    ok = 0
    for k in range(len(pts)):
        out[k], s = slsqp(pts[k], Xes[k], res[k], rt, acc)
        ok += s
    return out, ok


def one_hex():
    """the reference's 1hex_el input (test/runtests.jl:51-86)"""
    X = 1.0 * S
    IEN = np.arange(1, 9, dtype=np.int64)[None, :]
    return X, IEN, np.array([1.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 1.0])


def main():
    import __graft_entry__ as graft
    from conftest import load_fixture
    O = graft.load_oracle()
    res = {}
    for name, rt in (("1hex_el", 0.5), ("beam_vfrac_03", 0.5), ("chapadlo", 0.5)):
        if name == "1hex_el":
            X, IEN, rn = one_hex()
            g = O.grid_make(X.min(0), X.max(0), 15, 3)     # runtests.jl:54
        else:
            X, IEN, rho = load_fixture(name)
            rn = O.dense_in_nodes(X, IEN, rho)
            g, _ = O.auto_grid(X, IEN)
        with O.iso_pair_log(2_000_000) as log:
            O.eval_distances(X, IEN, rn, rt, g, 1.1, want_xp=False)
        n = log.n
        el, v = log.el[:n], log.v[:n]
        nx, ny, nz = g.dims
        amin, cell = np.array(g.amin[:]), g.cell
        pts = np.stack([amin[0] + cell * (v % nx), amin[1] + cell * ((v // nx) % ny), amin[2] + cell * (v // (nx * ny))], 1)
        Xes, rs = X[IEN[el] - 1], rn[IEN[el] - 1]
        res[name + "_pairs"] = n
        for tag, acc in (("ref", 1e-5), ("tight", 1e-15)):
            chunks = np.array_split(np.arange(n), 64)
            with mp.Pool(8) as pool:
                parts = pool.map(work, [(pts[c], Xes[c], rs[c], rt, acc) for c in chunks])
            xi = np.concatenate([p[0] for p in parts])
            with O.iso_override(xi):
                d, _, _ = O.eval_distances(X, IEN, rn, rt, g, 1.1, want_xp=False)
            res[f"{name}_{tag}"] = d
            res[f"{name}_fail_{tag}"] = n - sum(p[1] for p in parts)
            # voxels whose value comes from an SLSQP result that is NOT on the iso-surface (|rho - rho_t| > 1e-9): the
            # reference uses whatever NLopt returns (ComputeCoordsOnIso.jl:79-86), but such a "distance" is to a point
            # that does not belong to the surface - comparisons may want to set them apart
            N = 0.125 * np.prod(1 + S[None] * xi[:, None, :], axis=2)
            infeas = np.abs(np.einsum("nk,nk->n", N, rs) - rt) > 1e-9
            dp = np.linalg.norm(pts - np.einsum("nk,nki->ni", N, Xes), axis=1)
            bad = np.zeros(g.ngp, bool)
            hit = infeas & (np.abs(dp - d[v]) <= 1e-12 * np.maximum(d[v], 1e-300))
            bad[v[hit]] = True
            res[f"{name}_{tag}_offsurface"] = np.flatnonzero(bad)
            print(name, tag, "pairs", n, "not converged", res[f"{name}_fail_{tag}"], "results off the surface", int(infeas.sum()),
                  "voxels decided by them", int(bad.sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, "slsqp_fields.npz"), **res)


if __name__ == "__main__":
    main()
