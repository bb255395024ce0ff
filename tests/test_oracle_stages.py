"""Oracle pins for the pre-stage / post-processing rows."""
import numpy as np
import pytest

from conftest import load_fixture


def test_gauss_tables(oracle):
    for n in (3, 9, 15, 20):
        x, w = oracle.gauss_legendre(n)
        xr, wr = np.polynomial.legendre.leggauss(n)
        assert np.abs(x - xr).max() < 5e-16 and np.abs(w - wr).max() < 5e-15


def test_threshold_bisection_reproduces_reference_literal(oracle):
    """runtests.jl:198 passes threshold_density = 0.518555 for cantilever_beam_vfrac_04.mat, which is
    round(best_threshold, sigdigits=6) as printed by find_threshold_for_volume (Isocontour_volume.jl:149):
    volume -> nodal densities -> bisection must land on it."""
    X, IEN, rho = load_fixture("beam_vfrac_04")
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    assert vd == pytest.approx(4800.0, rel=1e-12)            # 60x20x4 unit cubes
    assert vf == pytest.approx(rho.mean(), rel=1e-12)        # equal element volumes
    rn = oracle.dense_in_nodes(X, IEN, rho)
    rt, it = oracle.find_threshold(X, IEN, rn, vd * vf)
    assert float(f"{rt:.6g}") == 0.518555
    assert rt == 0.5185546875 and it == 9


def _sphere_sdf(n, r=0.5):
    """analytic sphere SDF sampled in Float32 on [-1,1]^3 (ConvergenceTests/SphereConvergenceTest.jl:13-90)"""
    ax = np.linspace(-1.0, 1.0, n + 1).astype(np.float32)
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    return (np.float32(r) - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32), np.float32(ax[1] - ax[0])


def _cube_sdf(n, side=1.0):
    ax = np.linspace(-1.0, 1.0, n + 1).astype(np.float32)
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    h = np.float32(side / 2)
    return (h - np.maximum(np.maximum(np.abs(X), np.abs(Y)), np.abs(Z))).astype(np.float32), np.float32(ax[1] - ax[0])


def test_volume_convergence_sphere(oracle):
    """SphereConvergenceTest.jl:355-378: rel. error < 10/5/2 % at N >= 16/32/64, decreasing"""
    errs = []
    for n, bound in ((16, 0.10), (32, 0.05), (64, 0.02)):
        sdf, edge = _sphere_sdf(n)
        v = oracle.volume_from_sdf(sdf, edge, order=20)
        e = abs(v - 4.0 / 3.0 * np.pi * 0.125) / (4.0 / 3.0 * np.pi * 0.125)
        assert e < bound
        errs.append(e)
    assert errs[0] > errs[1] > errs[2]


def test_volume_convergence_cube(oracle):
    """CubeConvergenceTest.jl:383-405: rel. error < 5/2/1 % at N >= 16/32/64"""
    for n, bound in ((16, 0.05), (32, 0.02), (64, 0.01)):
        sdf, edge = _cube_sdf(n)
        v = oracle.volume_from_sdf(sdf, edge, order=20)
        assert abs(v - 1.0) < bound


def test_remove_artifacts_semantics(oracle):
    g = oracle.grid_make([0, 0, 0], [1, 1, 1], 10, 0)     # 11^3 points
    nx, ny, nz = g.dims
    sdf = -np.ones((nz, ny, nx))
    sdf[1:8, 1:8, 1:8] = 2.0          # 343 nodes: main body
    sdf[9, 9, 9] = 3.0                # 1-node speck
    sdf[9:11, 0:2, 0] = 1.0           # 4-node component
    a = sdf.reshape(-1).copy()
    n = oracle.remove_artifacts(a, g, 0.0, 0.01)          # min size = max(1, round(3.43)) = 3
    assert n == 1 and a.reshape(sdf.shape)[9, 9, 9] == -3.0
    assert (a.reshape(sdf.shape)[9:11, 0:2, 0] == 1.0).all()
    b = sdf.reshape(-1).copy()
    assert oracle.remove_artifacts(b, g, 0.0, 0.02) == 5  # min size = round(6.86) = 7 -> both small ones go
    c = -np.ones(g.ngp)
    assert oracle.remove_artifacts(c, g, 0.0, 0.01) == 0  # no interior nodes (:150-153)


def test_rbf_pipeline_volume_preserved(oracle):
    """RBFs_smoothing: the level shift makes the coarse LSF enclose the target volume to 1e-4 (:280-296)"""
    X, IEN, rho = load_fixture("beam_vfrac_04")
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    rn = oracle.dense_in_nodes(X, IEN, rho)
    g, _ = oracle.auto_grid(X, IEN)
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.518555, g, 1.1, want_xp=False)
    sdf = d * oracle.sign_detection(X, IEN, rn, 0.518555, g)
    oracle.remove_artifacts(sdf, g)
    fine, th, its, lsf = oracle.rbf_smoothing(sdf, g, False, 1, vd * vf)
    edge = np.float32((np.float32(g.amax[0]) - np.float32(g.amin[0])) / np.float32(g.dims[0] - 1))
    assert its == 0 and fine.shape == (g.dims[2], g.dims[1], g.dims[0])
    assert abs(oracle.volume_from_sdf(fine, edge) - vd * vf) < 2e-2     # f32 sums; bisection stops at 1e-4
    # same field on the :same grid; coarse and fine coordinates are generated differently in the
    # reference (range vs explicit arithmetic), hence Float32 round-off differences
    assert np.allclose(fine, lsf + np.float32(th), rtol=0, atol=1e-5 * np.abs(lsf).max())
