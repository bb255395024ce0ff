"""GPU parity of the pre-stage and post-processing kernels against the oracle (through the C ABI)."""
import numpy as np
import pytest

from conftest import load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["sphere", "beam_vfrac_03", "chapadlo"])
def test_mesh_volume_and_nodal_densities(pkg, oracle, name):
    X, IEN, rho = load_fixture(name)
    mesh = pkg.Mesh(X, IEN)
    vd, vf = pkg.calculate_mesh_volume(mesh, rho)
    ovd, ovf = oracle.mesh_volume(X, IEN, rho)
    assert vd == pytest.approx(ovd, rel=1e-12) and vf == pytest.approx(ovf, rel=1e-12)   # summation order only
    rn = pkg.DenseInNodes(mesh, rho)
    orn = oracle.dense_in_nodes(X, IEN, rho)
    assert np.abs(rn - orn).max() <= 1e-12, np.abs(rn - orn).max()
    print(name, "V", vd, vf, "rho_n max|diff|", np.abs(rn - orn).max(), "bit-equal", int((rn == orn).sum()), "/", rn.size)
    if name == "sphere":   # reference known answers, HexSphereSdfTest.jl:26-27
        assert rn.max() == pytest.approx(1.0000000000000022, rel=1e-10, abs=1e-12)
        assert rn.mean() == pytest.approx(0.29490556408887564, rel=1e-10, abs=1e-12)


def test_nodal_densities_node_lists_from_the_device(pkg, oracle):
    """the node -> element lists of DenseInNodes are built on the device (counts, scan, scatter, per-node sort): the result
    is the same number for number whatever order the scatter's atomics ran in, also on a shuffled connectivity, and node
    ids outside 1..nnp are an error, not a fault"""
    from rho2sdf_jl_amd import synthetic
    X, IT, _ = synthetic.tet_mesh(9)
    rng = np.random.default_rng(5)
    rho = rng.random(len(IT))
    a = pkg.DenseInNodes(pkg.Mesh(X, IT), rho)
    for _ in range(3):
        assert np.array_equal(pkg.DenseInNodes(pkg.Mesh(X, IT), rho), a)
    assert np.abs(a - oracle.dense_in_nodes(X, IT, rho)).max() <= 1e-12
    bad = IT.copy()
    bad[7, 2] = len(X) + 1
    with pytest.raises(pkg._lib.R2SError, match="outside 1..nnp"):
        pkg.DenseInNodes(pkg.Mesh(X, bad), rho)
    bad[7, 2] = 0
    with pytest.raises(pkg._lib.R2SError, match="outside 1..nnp"):
        pkg.DenseInNodes(pkg.Mesh(X, bad), rho)


@pytest.mark.parametrize("name", ["beam_vfrac_03", "beam_vfrac_04"])
def test_find_threshold(pkg, oracle, name):
    X, IEN, rho = load_fixture(name)
    mesh = pkg.Mesh(X, IEN)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    rn = oracle.dense_in_nodes(X, IEN, rho)
    rt = pkg.find_threshold_for_volume(mesh, rn, vd * vf)
    ort, _ = oracle.find_threshold(X, IEN, rn, vd * vf)
    assert rt == ort
    if name == "beam_vfrac_04":
        assert float(f"{rt:.6g}") == 0.518555          # reference literal, runtests.jl:198
    with pytest.raises(pkg._lib.R2SError, match="outside the possible range"):
        pkg.find_threshold_for_volume(mesh, rn, vd * 2.0)


def _analytic(kind, n):
    ax = np.linspace(-1.0, 1.0, n + 1).astype(np.float32)
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    if kind == "sphere":
        s = np.float32(0.5) - np.sqrt(X * X + Y * Y + Z * Z)
    else:
        s = np.float32(0.5) - np.maximum(np.maximum(np.abs(X), np.abs(Y)), np.abs(Z))
    return s.astype(np.float32), np.float32(ax[1] - ax[0])


@pytest.mark.parametrize("kind,exact", [("sphere", 4.0 / 3.0 * np.pi * 0.125), ("cube", 1.0)])
def test_volume_from_sdf(pkg, oracle, kind, exact):
    """reference convergence tests (ConvergenceTests/*.jl) on the HIP kernel + oracle parity"""
    bounds = {"sphere": (0.10, 0.05, 0.02), "cube": (0.05, 0.02, 0.01)}[kind]
    for n, b in zip((16, 32, 64), bounds):
        sdf, edge = _analytic(kind, n)
        v = pkg.calculate_volume_from_sdf(sdf, edge, detailed_quad_order=20)
        assert abs(v - exact) / exact < b
        ov = oracle.volume_from_sdf(sdf, edge, order=20)
        assert v == pytest.approx(ov, rel=2e-5)        # Float32 summation order
    sdf, edge = _analytic(kind, 32)
    assert pkg.calculate_volume_from_sdf(sdf, edge) == pytest.approx(oracle.volume_from_sdf(sdf, edge, order=9), rel=2e-5)


def _raw_sdf(oracle, name, rt):
    X, IEN, rho = load_fixture(name)
    rn = oracle.dense_in_nodes(X, IEN, rho)
    g, _ = oracle.auto_grid(X, IEN)
    d, _, _ = oracle.eval_distances(X, IEN, rn, rt, g, 1.1, want_xp=False)
    return X, IEN, rho, g, d * oracle.sign_detection(X, IEN, rn, rt, g)


def test_remove_artifacts(pkg, oracle):
    X, IEN, rho, og, sdf = _raw_sdf(oracle, "chapadlo", 0.5)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    rng = np.random.default_rng(5)
    noisy = sdf.copy()
    idx = rng.choice(sdf.size, 400, replace=False)       # sprinkle interior specks
    noisy[idx] = np.abs(noisy[idx])
    for ratio in (0.01, 0.2):
        a, b = noisy.copy(), noisy.copy()
        na = pkg.remove_sdf_artifacts(a, pg, min_component_ratio=ratio)
        nb = oracle.remove_artifacts(b, og, 0.0, ratio)
        assert na == nb and np.array_equal(a, b)
        print("artifacts ratio", ratio, "flipped", na)
    c = -np.abs(sdf)
    assert pkg.remove_sdf_artifacts(c, pg) == 0
    with pytest.raises(pkg._lib.R2SError, match="doesn't match grid points"):
        pkg.remove_sdf_artifacts(np.zeros(7), pg)


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [None, "16"])
def test_remove_artifacts_many_components(pkg, oracle, monkeypatch, cap):
    """a noisy field: thousands of components of all sizes, ties in the largest size, components cut by the 64-voxel
    pieces of the labelling; with a short list of roots (R2S_CCL_ROOTS_CAP) the sweeps over all counters take over"""
    if cap:
        monkeypatch.setenv("R2S_CCL_ROOTS_CAP", cap)
    rng = np.random.default_rng(23)
    pg = pkg.Grid(np.zeros(3), np.array([2.0, 1.3, 0.9]), 70, 1)
    og = oracle.grid_make(np.zeros(3), np.array([2.0, 1.3, 0.9]), 70, 1)
    nx, ny, nz = pg.dims
    f = rng.normal(size=(nz, ny, nx))
    for ax in range(3):                                   # a little smoothing: components of many sizes
        f = f + np.roll(f, 1, axis=ax)
    sdf = (f - 0.8).ravel()
    for ratio in (0.01, 0.5):
        a, b = sdf.copy(), sdf.copy()
        na = pkg.remove_sdf_artifacts(a, pg, min_component_ratio=ratio)
        nb = oracle.remove_artifacts(b, og, 0.0, ratio)
        assert na == nb > 0 and np.array_equal(a, b)
    two = -np.ones((nz, ny, nx))                          # two components of the same (largest) size: the first one stays
    two[1:3, 1:3, 1:3] = 1.0
    two[5:7, 5:7, 5:7] = 1.0
    a, b = two.ravel().copy(), two.ravel().copy()
    assert pkg.remove_sdf_artifacts(a, pg, min_component_ratio=2.0) == oracle.remove_artifacts(b, og, 0.0, 2.0)
    assert np.array_equal(a, b)


def _set_rbf_mode(monkeypatch, mode, names=("R2S_RBF_MATVEC", "R2S_RBF_APPLY")):
    """mode "walk" = the default kernels (row walk, r2s_rbf_walk.hpp): no environment override"""
    for name in names:
        if mode == "walk":
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, mode)


@pytest.mark.parametrize("interp,smooth", [(False, 1), (True, 1), (True, 2)])
def test_rbf_smoothing(pkg, oracle, interp, smooth):
    """BASELINE configs 2/3: beam, approximation and interpolation, :same and :fine grids"""
    X, IEN, rho, og, sdf = _raw_sdf(oracle, "beam_vfrac_04", 0.518555)
    oracle.remove_artifacts(sdf, og)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    info = {}
    fine = pkg.RBFs_smoothing(sdf, pg, interp, smooth, vd * vf, info=info)
    ofine, oth, oits, olsf = oracle.rbf_smoothing(sdf, og, interp, smooth, vd * vf)
    assert fine.shape == ofine.shape
    scale = np.abs(olsf).max()
    # Float32 pipeline: exp() differs by <= 1 ulp(f64) between libm and the device, CG dot products are
    # reduced in a different order -> compare at Float32 round-off of the field magnitude
    tol = 5e-5 if interp else 2e-6
    assert np.abs(info["lsf"] - olsf).max() <= tol * scale, np.abs(info["lsf"] - olsf).max() / scale
    assert abs(info["cg_iterations"] - oits) <= 1
    # the level shift is only defined to the 1e-4 volume tolerance of the bisection
    assert abs(info["th"] - oth) <= 1e-3 * scale
    assert np.abs((fine - np.float32(info["th"])) - (ofine - np.float32(oth))).max() <= tol * scale
    print("rbf", interp, smooth, "th", info["th"], oth, "cg", info["cg_iterations"], oits,
          "max|dLSF|/scale", np.abs(info["lsf"] - olsf).max() / scale)


def test_rho2sdf_end_to_end(pkg, oracle):
    """the reference's default smoke test (runtests.jl:186-207): beam, threshold 0.518555, automatic grid,
    rbf_interp = true, rbf_grid = :same - all stages on the GPU, compared stage by stage with the oracle"""
    X, IEN, rho = load_fixture("beam_vfrac_04")
    opts = pkg.Rho2sdfOptions(threshold_density=0.518555, sdf_grid_setup="automatic", rbf_interp=True, rbf_grid="same")
    fine_sdf, fine_grid, sdf_grid, sdf_dists = pkg.rho2sdf("beam", X, IEN, rho, options=opts)
    og, _ = oracle.auto_grid(X, IEN)
    rn = oracle.dense_in_nodes(X, IEN, rho)
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.518555, og, 1.1, want_xp=False)
    ref = d * oracle.sign_detection(X, IEN, rn, 0.518555, og)
    oracle.remove_artifacts(ref, og)
    assert np.array_equal(np.abs(ref) == 1e10, np.abs(sdf_dists) == 1e10)
    assert np.array_equal(np.sign(ref), np.sign(sdf_dists))
    real = np.abs(ref) < 1e9
    assert np.allclose(sdf_dists[real], ref[real], rtol=1e-6, atol=1e-12)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    ofine, oth, _, olsf = oracle.rbf_smoothing(ref, og, True, 1, vd * vf)
    assert fine_sdf.shape == ofine.shape == (11, 27, 67)
    assert np.abs(fine_sdf - ofine).max() <= 2e-3 * np.abs(olsf).max()
    # automatic threshold path (options.threshold_density = nothing)
    opts2 = pkg.Rho2sdfOptions(sdf_grid_setup="automatic", rbf_interp=False)
    out = pkg.rho2sdf("beam", X, IEN, rho, options=opts2)
    assert out[0].shape == (11, 27, 67)


def test_mesh_volume_tet4(pkg, oracle):
    """calculate_element_volume TET4 (MeshVolume.jl:75-117): restated with the reference's Jacobian (25 % low)"""
    from rho2sdf_jl_amd import synthetic
    X, IT, rn = synthetic.tet_mesh(5)
    rho = np.linspace(0.1, 0.9, len(IT))
    vd, vf = pkg.calculate_mesh_volume(pkg.Mesh(X, IT), rho)
    ovd, ovf = oracle.mesh_volume_tet4(X, IT, rho)
    assert vd == pytest.approx(ovd, rel=1e-12) and vf == pytest.approx(ovf, rel=1e-12)
    assert vd == pytest.approx(0.75 * 8.0, rel=1e-12)      # [-1,1]^3, see the note in r2s_pre.hip
    rn_gpu = pkg.DenseInNodes(pkg.Mesh(X, IT), rho)
    assert np.abs(rn_gpu - oracle.dense_in_nodes(X, IT, rho)).max() <= 1e-12


@pytest.mark.parametrize("badval", [0, -3, 10**9])
def test_connectivity_outside_node_range_is_an_error(pkg, oracle, badval):
    """IEN ids outside 1..nnp: the call fails with a message (the reference would throw a BoundsError)
    and the device survives to run the next call."""
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.Grid(X.min(0), X.max(0), 10, 3)
    bad = IEN.copy()
    bad[len(bad) // 2, 3] = badval
    with pytest.raises(pkg._lib.R2SError, match="outside 1..nnp"):
        pkg.sdf_fused(pkg.Mesh(X, bad), pg, rn, 0.5)
    sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=2.5)
    assert (np.abs(sdf) == 1e10).sum() == 1836


def test_rbf_matvec_variants_are_bit_identical(pkg, oracle, monkeypatch):
    """the CG's matrix-vector product has three implementations: the table of distinct matrix entries (default; two kernels), the
    materialised matrix (R2S_RBF_MATVEC=k) and on-the-fly evaluation (=fly).  All three form every row sum from the
    same Float32 values in the same order, so weights, iteration counts and the smoothed field must be identical"""
    X, IEN, rho, og, sdf = _raw_sdf(oracle, "beam_vfrac_04", 0.518555)
    oracle.remove_artifacts(sdf, og)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    outs = {}
    for mode in ("walk", "lut", "lutg", "k", "fly"):   # walk: default; lut: table rows staged in LDS; lutg: entries gathered from L1 / L2
        _set_rbf_mode(monkeypatch, mode, ("R2S_RBF_MATVEC",))
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, pg, True, 1, vd * vf, info=info), info["cg_iterations"], info["th"], info["lsf"])
    for mode in ("walk", "lutg", "k", "fly"):
        assert outs[mode][1] == outs["lut"][1] and outs[mode][2] == outs["lut"][2]
        assert np.array_equal(outs[mode][0], outs["lut"][0]) and np.array_equal(outs[mode][3], outs["lut"][3])
    pkg._lib.lib().r2s_release_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("threshold", [1e-2, 0.1])
def test_rbf_tables_other_kernel_thresholds(pkg, oracle, monkeypatch, threshold):
    """other supports (threshold 1e-2: radius 2 with fewer neighbours; 0.1: radius 1) run the generic instantiations of
    the table kernels: CG with table / on the fly products and table / neighbour-by-neighbour evaluation must agree bit
    for bit there too"""
    X, IEN, rho, og, sdf = _raw_sdf(oracle, "beam_vfrac_04", 0.518555)
    oracle.remove_artifacts(sdf, og)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    outs = {}
    for mode in ("walk", "lut", "fly"):
        _set_rbf_mode(monkeypatch, mode)
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, pg, True, 1, vd * vf, threshold, info=info), info["cg_iterations"], info["th"], info["lsf"])
    for mode in ("walk", "fly"):
        assert outs[mode][1] == outs["lut"][1] and outs[mode][2] == outs["lut"][2]
        assert np.array_equal(outs[mode][0], outs["lut"][0]) and np.array_equal(outs[mode][3], outs["lut"][3])
    assert np.isfinite(outs["lut"][0]).all() and outs["lut"][1] > 0
    pkg._lib.lib().r2s_release_cache()


@pytest.mark.gpu
def test_rbf_lds_kernels_on_a_wide_grid(pkg, monkeypatch):
    """the kernels that stage their table rows in LDS need rows of >= 256 points (a workgroup of 256 consecutive rows then
    spans at most two lattice rows): a 261 x 41 x 37 lattice with a banded synthetic SDF, CG + evaluation, against the
    gathered-table kernels and neighbour-by-neighbour evaluation - bit for bit"""
    g = pkg.Grid(np.array([0.013, -0.2, 0.07]), np.array([26.013, 3.8, 3.67]), 260, 0)
    nx, ny, nz = g.dims
    assert nx >= 256
    ax = [g.AABB_min[i] + g.cell_size * np.arange(n) for i, n in enumerate((nx, ny, nz))]
    r = np.sqrt(((ax[0][None, None, :] - 13.0) / 6.0) ** 2 + (ax[1][None, :, None] - 1.8) ** 2 + (ax[2][:, None, None] - 1.9) ** 2)
    sdf = 1.3 - r
    sdf = np.where(np.abs(sdf) < 6 * g.cell_size, sdf, np.sign(sdf) * 1e10).ravel()
    target = float((sdf > 0).sum()) * g.cell_size ** 3
    outs = {}
    for mode in ("walk", "lut", "lutg", "fly"):
        _set_rbf_mode(monkeypatch, mode)
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, g, True, 1, target, info=info), info["cg_iterations"], info["th"], info["lsf"])
    for mode in ("walk", "lutg", "fly"):
        assert outs[mode][1] == outs["lut"][1] and outs[mode][2] == outs["lut"][2]
        assert np.array_equal(outs[mode][0], outs["lut"][0]) and np.array_equal(outs[mode][3], outs["lut"][3])
    assert outs["lut"][1] > 0 and np.isfinite(outs["lut"][0]).all()
    pkg._lib.lib().r2s_release_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(4, 3, 5), (61, 7, 5), (5, 67, 3), (130, 9, 8), (311, 6, 6), (64, 64, 5)])
def test_rbf_walk_kernels_on_odd_lattices(pkg, monkeypatch, dims):
    """the row-walk kernels on lattices that are smaller than their tiles in one direction or another (fewer columns than a
    wavefront, fewer rows than a walk, fewer planes than the stencil reaches, widths just beyond a multiple of the 60 / 64
    outputs of a wavefront): CG + both evaluations against neighbour-by-neighbour evaluation, bit for bit"""
    nx, ny, nz = dims
    lo = np.array([0.375, -0.25, 0.125])   # (dyadic numbers: the cell count of `Grid` comes out exact)
    g = pkg.Grid(lo, lo + 0.125 * (np.array(dims) - 1.0), max(dims) - 1, 0)
    assert g.dims == dims
    ax = [g.AABB_min[i] + g.cell_size * np.arange(n) for i, n in enumerate(dims)]
    c = [0.5 * (a[0] + a[-1]) for a in ax]
    r = np.sqrt((ax[0][None, None, :] - c[0]) ** 2 + (ax[1][None, :, None] - c[1]) ** 2 + (ax[2][:, None, None] - c[2]) ** 2)
    sdf = 0.3 * g.cell_size * max(dims) - r
    sdf = np.where(np.abs(sdf) < 3 * g.cell_size, sdf, np.sign(sdf) * 1e10).ravel()
    target = max(float((sdf > 0).sum()), 1.0) * g.cell_size ** 3
    outs = {}
    for mode in ("walk", "fly"):
        _set_rbf_mode(monkeypatch, mode)
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, g, True, 1, target, info=info), info["cg_iterations"], info["th"], info["lsf"])
    assert outs["fly"][1] == outs["walk"][1] and outs["fly"][2] == outs["walk"][2]
    assert np.array_equal(outs["fly"][0], outs["walk"][0]) and np.array_equal(outs["fly"][3], outs["walk"][3])
    assert np.isfinite(outs["walk"][0]).all()
    pkg._lib.lib().r2s_release_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [False, True])
def test_rbf_evaluation_table_is_bit_identical(pkg, oracle, monkeypatch, interp):
    """same-grid evaluation (the LSF of the level bisection and the output field at smooth = 1) through the table of
    distinct kernel values vs neighbour-by-neighbour evaluation (R2S_RBF_APPLY=fly): same values, same order"""
    X, IEN, rho, og, sdf = _raw_sdf(oracle, "beam_vfrac_04", 0.518555)
    oracle.remove_artifacts(sdf, og)
    vd, vf = oracle.mesh_volume(X, IEN, rho)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    outs = {}
    for mode in ("walk", "lut", "lutg", "fly"):   # walk: default; lut: table rows staged in LDS where a workgroup spans <= 2 rows; lutg: gathered
        _set_rbf_mode(monkeypatch, mode, ("R2S_RBF_APPLY",))
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, pg, interp, 1, vd * vf, info=info), info["th"], info["lsf"])
    for mode in ("walk", "lutg", "fly"):
        assert outs[mode][1] == outs["lut"][1]
        assert np.array_equal(outs[mode][0], outs["lut"][0]) and np.array_equal(outs[mode][2], outs["lut"][2])
    assert np.isfinite(outs["lut"][0]).all() and np.ptp(outs["lut"][0]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("smooth,dims", [(2, (40, 23, 17)), (3, (21, 30, 13)), (2, (131, 9, 70)), (4, (12, 11, 15))])
def test_rbf_refined_grid_tables_are_bit_identical(pkg, monkeypatch, smooth, dims):
    """rbf_grid = :fine (smooth >= 2): the output field through the tables of its parity classes (one wavefront = 64
    targets of one parity of a row) vs neighbour-by-neighbour evaluation (R2S_RBF_APPLY=fly) - same values, same order,
    on lattices whose sizes are no multiples of anything, with sentinels in the input"""
    rng = np.random.default_rng(sum(dims) + smooth)
    lo = np.array([0.375, -0.25, 0.125])   # (dyadic numbers: the cell count of `Grid` comes out exact)
    pg = pkg.Grid(lo, lo + 0.125 * (np.array(dims) - 1.0), max(dims) - 1, 0)
    assert pg.dims == dims
    ax = [pg.AABB_min[i] + pg.cell_size * np.arange(n) for i, n in enumerate(dims)]
    c = [0.45 * (a[0] + a[-1]) for a in ax]
    r = np.sqrt((ax[0][None, None, :] - c[0]) ** 2 + (ax[1][None, :, None] - c[1]) ** 2 + (ax[2][:, None, None] - c[2]) ** 2)
    sdf = (0.3 * pg.cell_size * max(dims) - r + 0.01 * rng.normal(size=r.shape)).ravel()
    sdf[rng.random(sdf.size) < 0.05] = 1e10
    vol = max(float((sdf > 0).sum()), 1.0) * pg.cell_size ** 3
    outs = {}
    for mode in ("walk", "fly"):
        _set_rbf_mode(monkeypatch, mode, ("R2S_RBF_APPLY",))
        info = {}
        outs[mode] = (pkg.RBFs_smoothing(sdf, pg, True, smooth, vol, info=info), info["th"])
    assert outs["walk"][0].size == np.prod([(d - 1) * smooth + 1 for d in dims])
    assert outs["walk"][1] == outs["fly"][1]
    assert np.array_equal(outs["walk"][0], outs["fly"][0])
    assert np.isfinite(outs["walk"][0]).all() and np.ptp(outs["walk"][0]) > 0
