"""What stands between the oracle and the Julia reference, as numbers (VERDICT r1 "pin what can be pinned").

1. the production oracle (step tolerances 1e-6 / 1e-7) against its tight build (-DORC_TIGHT: 1e-12 / 1e-13) on every
   fixture, the reference's 1hex_el input and distorted elements: sentinel sets and signs identical, and the LARGEST
   relative change of a band distance is asserted (<= 1e-7; north_star asks 1e-6);
2. find_local_coordinates: the single-start clamped Newton against 9-start L-BFGS-B vectors (the reference's problem
   statement, FindLocalCoordinates.jl:27-104) on strongly distorted hexahedra - the booleans the callers use;
3. compute_coords_on_iso, TET4: the closed form against SLSQP vectors (ComputeCoordsOnIso.jl:90-181);
4. evalDistances on the reference's 1hex_el input and two of its fixtures with every iso projection done by scipy's
   SLSQP (golden FIELDS, tests/golden/make_slsqp_field_vectors.py): converged, and at the reference's own 1e-5 tolerances.
The GPU path is bit-identical to the production oracle on these fixtures (tests/test_parity_gpu.py), so every count
below is also the product's.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, load_fixture

GOLD = os.path.join(ROOT, "tests", "golden")


def one_hex():
    """the reference's `1hex_el` input (test/runtests.jl:51-86): ONE HEX8 on [-1,1]^3 whose iso-surface clips two
    opposite corners of a face (two sheets in one element)"""
    X = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
    IEN = np.arange(1, 9, dtype=np.int64)[None, :]
    rho_n = np.array([1.0, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 1.0])
    return X, IEN, rho_n


def _cases(oracle):
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = one_hex()
    for N in (15, 20, 31):                                        # runtests.jl:54 uses N = 15
        yield f"1hex_el N{N}", X, IEN, rn, 0.5, oracle.grid_make(X.min(0), X.max(0), N, 3), 1.1
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    yield "sphere N25", X, IEN, rn, 0.5, oracle.grid_make(X.min(0), X.max(0), 25, 3), 1.1
    yield "sphere N10 band 2.5", X, IEN, rn, 0.5, oracle.grid_make(X.min(0), X.max(0), 10, 3), 2.5
    for name, rt in (("beam_vfrac_03", 0.5), ("beam_vfrac_04", 0.518555), ("chapadlo", 0.5)):
        X, IEN, rho = load_fixture(name)
        rn = oracle.dense_in_nodes(X, IEN, rho)
        yield name, X, IEN, rn, rt, oracle.auto_grid(X, IEN)[0], 1.1
    X, IEN, rn = synthetic.hex_mesh(12)
    yield "synthetic 12^3 / 64^3", X, IEN, rn, 0.5, oracle.grid_make(X.min(0), X.max(0), synthetic.grid_n_max_for_points(64), 3), 1.1
    for seed, jit in ((1, .30), (2, .35), (3, .25)):
        X, IEN, _ = synthetic.hex_mesh(7, jitter=jit, seed=20240501 + seed)
        rn = np.clip(np.random.default_rng(seed).normal(0.5, 0.35, len(X)), 0, 1)
        yield f"distorted hex, random density, seed {seed}", X, IEN, rn, 0.5, \
            oracle.grid_make(X.min(0), X.max(0), synthetic.grid_n_max_for_points(48), 3), 1.1


def test_production_tolerances_against_the_frozen_tight_oracle(pkg, oracle):
    """production step tolerance (1e-6) against the tight build (1e-12) on every fixture, the reference's 1hex_el input
    and strongly distorted elements with random densities: NO band voxel may move by more than 1e-6 (north_star's
    bar; round 2 froze COUNTS of voxels that were up to 28 % off - pairs at the iteration cap, on Gauss-Newton steps
    creeping towards saddle points, or stuck off the iso-surface).  Achieved: <= 1e-8 everywhere."""
    rows = []
    for name, X, IEN, rn, rt, g, bf in _cases(oracle):
        d, _, st = oracle.eval_distances(X, IEN, rn, rt, g, bf, want_xp=False)
        s = oracle.sign_detection(X, IEN, rn, rt, g)
        with oracle.tight():
            d2, _, _ = oracle.eval_distances(X, IEN, rn, rt, g, bf, want_xp=False)
            s2 = oracle.sign_detection(X, IEN, rn, rt, g)
        assert np.array_equal(d == 1e10, d2 == 1e10), f"{name}: sentinel set moves with the tolerances"
        assert np.array_equal(s, s2), f"{name}: {int((s != s2).sum())} signs move with the tolerances"
        real = d < 1e9
        rel = np.abs(d[real] - d2[real]) / np.maximum(d2[real], 1e-300)
        rows.append((name, int(real.sum()), float(rel.max()), int((rel > 1e-9).sum()), st["n_iso_fail"], st["n_iso_solves"]))
        assert rel.max() <= 1e-7, f"{name}: a band voxel moves by {rel.max():.2e} (relative) with the solver tolerance"
        assert st["n_iso_fail"] <= 6e-4 * st["n_iso_solves"], (name, st)   # runs that end without a KKT point
    for r in rows:
        print("drift vs tight oracle: %-42s band voxels %6d  max rel %.1e  > 1e-9: %3d  failed solves %d / %d" % r)


def test_inverse_map_against_nine_start_lbfgs(oracle):
    """3 000 (distorted element, point) cases: where the bounded minimiser is a root of the map inside the +-1.1 box
    the Newton restatement must find the same root; the three comparisons the callers make (max|xi| < 0.95 / 1.001 /
    1.01: SignDetection.jl:56-62, sdfOnDensityField.jl:92-101) must agree everywhere except within 1e-6 of a threshold"""
    d = np.load(os.path.join(GOLD, "lbfgs_invmap.npz"))
    n_root = n_bool = n_amb = n_fail = 0
    worst = 0.0
    for k in range(len(d["x"])):
        ok, xi = oracle.inv_map_hex8(d["x"][k], d["Xe"][k])
        m_g, m = np.abs(d["xi"][k]).max(), np.abs(xi).max()
        has_root = d["fmin"][k] < 1e-20
        n_fail += not ok
        amb = any(abs(m_g - t) < 1e-6 for t in (0.95, 1.001, 1.01))
        n_amb += amb
        if not amb and (m_g < 0.95, m_g < 1.001, m_g < 1.01) != (m < 0.95, m < 1.001, m < 1.01):
            n_bool += 1
        if has_root and m_g < 1.1 - 1e-9:
            n_root += 1
            assert ok, f"case {k}: Newton from 0 fails although the map has a root in the box"
            worst = max(worst, float(np.abs(xi - d["xi"][k]).max()))
    assert n_bool == 0, f"{n_bool} boolean decisions differ from the multi-start minimiser"
    assert n_root > 1500 and worst < 1e-9
    print(f"inverse map: {len(d['x'])} cases, {n_root} with a root in the box (max |dxi| {worst:.1e}), "
          f"{n_fail} Newton failures (all without a root in the box), {n_amb} within 1e-6 of a threshold, 0 boolean differences")


def test_tet4_projection_against_slsqp(oracle):
    d = np.load(os.path.join(GOLD, "slsqp_tet4_projection.npz"))
    worst = 0.0
    for k in range(len(d["x"])):
        lam = oracle.iso_project_tet4(d["x"][k], d["Xe"][k], d["re"][k], float(d["rt"][k]))
        N = np.array([lam[0], lam[1], lam[2], 1.0 - lam.sum()])
        assert lam.min() >= -1e-12 and lam.sum() <= 1 + 1e-12 and abs(d["re"][k] @ N - d["rt"][k]) < 1e-12
        dist = np.linalg.norm(d["x"][k] - d["Xe"][k].T @ N)
        worst = max(worst, abs(dist - d["dist"][k]) / max(d["dist"][k], 1e-300))
    assert worst < 1e-9, worst                    # 600 cases; achieved 7e-15
    print(f"TET4 projection: {len(d['x'])} SLSQP vectors, max rel distance error {worst:.1e}")


# fixture -> (band voxels; against the CONVERGED SLSQP field, voxels SLSQP decides with a point ON the iso-surface only:
#             product nearer by more than 1e-6 / farther by more than 1e-6; voxels beyond 1e-6 of the field at the
#             reference's own 1e-5 tolerances (all voxels); voxels on which the two SLSQP fields differ by more than 1e-6)
SLSQP_FIELD_COUNTS = {"1hex_el": (5832, 99, 0, 2253, 2162), "beam_vfrac_03": (9298, 3, 0, 931, 929),
                      "chapadlo": (17875, 58, 11, 1226, 1163)}


# (measured with this oracle: the product never lies farther from the iso-surface than converged SLSQP on 1hex_el / beam,
#  on 11 voxels of chapadlo by up to 9.3 % = 0.24 cell (another local minimum from the same start); it lies NEARER by up to
#  0.81 / 0.0006 / 0.77 cell.  Against SLSQP stopped at the reference's 1e-5 tolerances the largest difference is the same
#  0.81 / 0.0006 / 0.77 cell: that field is the one that has not converged.)
SLSQP_FIELD_SIZES = {"1hex_el": (0.0, 0.8074, 290, 86, 0.8074), "beam_vfrac_03": (0.0, 0.000618, 55, 4, 0.000618),
                     "chapadlo": (0.0927, 0.7720, 161, 67, 0.7720)}


@pytest.mark.parametrize("name", sorted(SLSQP_FIELD_COUNTS))
def test_fields_against_independent_slsqp(oracle, name):
    """the whole evalDistances field of the reference's 1hex_el input (runtests.jl:51-86) and of two fixtures
    (automatic grid) with every iso projection by scipy's SLSQP (tests/golden/make_slsqp_field_vectors.py).
    Converged SLSQP: wherever the product differs it has found a NEARER point of the iso-surface (1hex_el 99 voxels -
    SLSQP stops on the saddle points the symmetric lattice points start towards -, beam 3) except 11 voxels of chapadlo
    (other local minima, up to 9 %); round 2: 12 / 89 differing voxels on beam / chapadlo, up to 0.57 cell farther.
    SLSQP results that are not on the surface (|rho - rho_t| > 1e-9: 47 voxels of chapadlo) are set apart - the
    reference uses whatever NLopt returns (ComputeCoordsOnIso.jl:79-86), but a distance to a point off the surface is
    no target.  At the reference's own tolerances (1e-5) SLSQP itself is 1e-6 away from its converged answer on
    2 162 / 929 / 1 163 voxels: that, not the restatement, bounds parity with a Julia run."""
    from test_oracle_drift import one_hex
    F = np.load(os.path.join(GOLD, "slsqp_fields.npz"))
    if name == "1hex_el":
        X, IEN, rn = one_hex()
        g = oracle.grid_make(X.min(0), X.max(0), 15, 3)
    else:
        X, IEN, rho = load_fixture(name)
        rn = oracle.dense_in_nodes(X, IEN, rho)
        g, _ = oracle.auto_grid(X, IEN)
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, g, 1.1, want_xp=False)
    band, n_near, n_far, n_ref, n_self = SLSQP_FIELD_COUNTS[name]
    real = d < 1e9
    assert int(real.sum()) == band
    r = F[f"{name}_tight"]
    assert np.array_equal(r == 1e10, d == 1e10) and np.array_equal(F[f"{name}_ref"] == 1e10, d == 1e10)
    on = real.copy()
    on[F[f"{name}_tight_offsurface"]] = False
    rel = (d[on] - r[on]) / np.maximum(r[on], 1e-300)
    near, far = int((rel < -1e-6).sum()), int((rel > 1e-6).sum())
    rr = F[f"{name}_ref"]
    ref = int((np.abs(d[real] - rr[real]) / np.maximum(rr[real], 1e-300) > 1e-6).sum())
    self_ = int((np.abs(rr[real] - r[real]) / np.maximum(r[real], 1e-300) > 1e-6).sum())
    print(f"{name}: {band} band voxels; vs converged SLSQP on the surface ({int(on.sum())} voxels): nearer {near}, farther {far}"
          f" (max {rel.max():.2e}); beyond 1e-6 of SLSQP at 1e-5: {ref}; SLSQP 1e-5 vs converged: {self_}")
    assert far <= n_far and abs(near - n_near) <= 3 and ref <= n_ref + 5 and abs(self_ - n_self) <= 5
    # ... and the SIZES of those differences (round-3 verdict: counts alone hide them).  fixture -> against converged SLSQP:
    # largest relative "farther", largest "nearer" in cells; against SLSQP at the reference's 1e-5 tolerances (the closest
    # stand-in for what NLopt returns): voxels beyond 1e-4 / 1e-2 relative, largest difference in cells
    far_rel_max, near_cells_max, n_1e4, n_1e2, ref_cells_max = SLSQP_FIELD_SIZES[name]
    cells = (d[on] - r[on]) / g.cell
    relref = np.abs(d[real] - rr[real]) / np.maximum(rr[real], 1e-300)
    refcells = np.abs(d[real] - rr[real]) / g.cell
    print(f"{name}: farther by at most {max(rel.max(), 0.0):.3e} relative / {max(cells.max(), 0.0):.3e} cell, nearer by at most "
          f"{-cells.min():.4f} cell; vs SLSQP@1e-5: {int((relref > 1e-4).sum())} voxels beyond 1e-4, {int((relref > 1e-2).sum())} beyond 1e-2, "
          f"max {refcells.max():.4f} cell")
    assert rel.max() <= far_rel_max * 1.01 + 1e-8
    assert -cells.min() <= near_cells_max * 1.01
    assert int((relref > 1e-4).sum()) <= n_1e4 + 3 and int((relref > 1e-2).sum()) <= n_1e2 + 2
    assert refcells.max() <= ref_cells_max * 1.01
