#!/usr/bin/env python3
"""The oracle's SQP (= the kernel's, operation for operation) against scipy's Kraft SLSQP on random elements:
jittered unit hexes with (a) smooth radial density fields, (b) independent random nodal densities - the worst case
for a local method (several disconnected pieces of the iso-surface per element).  CPU only.
  python tools/sqp_vs_slsqp.py [--n 2000] [--seed 7]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import __graft_entry__ as graft
import make_slsqp_vectors as M

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--seed", type=int, default=7)
a = ap.parse_args()
O = graft.load_oracle()
rng = np.random.default_rng(a.seed)
for kind in ("smooth", "random"):
    cat = dict(agree=0, other_min_nearer=0, other_min_farther=0, only_sqp=0, only_slsqp=0, both_fail=0)
    its = []
    n = 0
    while n < a.n:
        Xe = 0.5 * M.S + rng.uniform(-0.15, 0.15, size=(8, 3))
        if kind == "smooth":
            c = rng.uniform(-1.5, 1.5, 3); r0 = rng.uniform(0.6, 1.6)
            re = np.clip(1.0 - (np.linalg.norm(Xe - c, axis=1) - r0), 0.0, 1.0)
        else:
            re = rng.uniform(0.0, 1.0, 8)
        if not (re.min() < 0.5 < re.max()):
            continue
        x = rng.uniform(-0.9, 0.9, 3)
        xi, it = O.iso_project_hex8(x, Xe, re, 0.5)
        N = M.shape(xi); dm = np.linalg.norm(x - Xe.T @ N); mok = abs(re @ N - 0.5) < 1e-9
        rs = M.slsqp(x, Xe, re, 0.5); Ns = M.shape(rs.x); ds = np.linalg.norm(x - Xe.T @ Ns)
        sok = bool(rs.success) and abs(re @ Ns - 0.5) < 1e-9
        its.append(min(it, 61)); n += 1
        if mok and sok:
            cat["agree" if abs(dm - ds) <= 1e-6 * max(ds, 1e-300) else ("other_min_nearer" if dm < ds else "other_min_farther")] += 1
        elif mok: cat["only_sqp"] += 1
        elif sok: cat["only_slsqp"] += 1
        else: cat["both_fail"] += 1
    print(kind, {k: f"{100.0 * v / a.n:.2f}%" for k, v in cat.items()}, "mean iterations %.2f" % np.mean(its))
