"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): sentinel set and sign bit-exact, Float64 distances within
1e-6 relative.  The kernels keep the oracle's operation order, so distances are also
expected to agree far tighter than the bar; the achieved maximum is printed.
"""
import numpy as np
import pytest

from conftest import ROOT, block_mesh, load_fixture

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _compare(pkg, oracle, X, IEN, rn, rt, pg, og, bf, label):
    mesh = pkg.Mesh(X, IEN)
    st = {}
    dist, xp = pkg.evalDistances(mesh, pg, rn, rt, band_factor=bf, stats=st)
    sign = pkg.Sign_Detection(mesh, pg, rn, rt)
    sdf = pkg.sdf_fused(mesh, pg, rn, rt, band_factor=bf)
    odist, oxp, ost = oracle.eval_distances(X, IEN, rn, rt, og, bf)
    osign = oracle.sign_detection(X, IEN, rn, rt, og)
    sent = odist == 1e10
    assert np.array_equal(dist == 1e10, sent), f"{label}: sentinel set differs"
    assert np.array_equal(sign, osign), f"{label}: sign differs at {np.flatnonzero(sign != osign)[:10]}"
    real = ~sent
    rel = np.abs(dist[real] - odist[real]) / np.maximum(odist[real], 1e-300)
    # |d| ~ 0 voxels: compare absolutely against the cell size
    bad = (rel > RTOL) & (np.abs(dist[real] - odist[real]) > 1e-12 * og.cell)
    assert not bad.any(), f"{label}: {bad.sum()} distances beyond {RTOL}, max rel {rel.max()}"
    assert np.array_equal(sdf, dist * sign), f"{label}: fused sdf != dist*sign"
    assert np.allclose(xp[real], oxp[real], rtol=0, atol=1e-9 * max(1.0, np.abs(X).max()))
    assert np.array_equal(xp[sent], np.zeros_like(xp[sent]))
    if IEN.shape[1] == 8:
        # projections that end without a KKT point are counted, not hidden - the device's complete solver
        # (stragglers + overflow sweep) fails on exactly the pairs the oracle's does
        assert st["n_iso_fail"] == ost["n_iso_fail"], (label, st["n_iso_fail"], ost["n_iso_fail"])
        assert ost["n_iso_fail"] <= st["n_iso_straggler"] <= ost["n_iso_solves"]
    print(f"{label}: ngp {og.ngp} sentinels {int(sent.sum())} +1 signs {int((sign > 0).sum())} "
          f"max rel err {rel.max() if rel.size else 0:.3e} bit-equal {int((dist[real] == odist[real]).sum())}/{int(real.sum())} "
          f"items {st.get('n_items')} active tiles {st.get('n_active_tiles')} iso solves {ost.get('n_iso_solves')} "
          f"stragglers {st.get('n_iso_straggler')} without KKT point {st.get('n_iso_fail')}")


@pytest.mark.parametrize("bf", [1.1, 2.5])
@pytest.mark.parametrize("nmax", [10, 25])
def test_sphere(pkg, oracle, bf, nmax):
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, bf, f"sphere N{nmax} bf{bf}")


def test_sphere_known_answers_on_gpu(pkg, oracle):
    """reference test/HexSphereSdfTest.jl:28-29 evaluated with the HIP path"""
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.Grid(X.min(0), X.max(0), 10, 3)
    sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=2.5)
    assert (np.abs(sdf) == 1e10).sum() == 1836
    assert sdf.max() == pytest.approx(0.8669785608800439, rel=1e-10, abs=1e-12)
    assert sdf.mean() == pytest.approx(-3.7370242217627172e9, abs=1e5)


@pytest.mark.parametrize("rt", [0.1, 0.9])
def test_sphere_edge_thresholds(pkg, oracle, rt):
    """HexSphereSdfTest.jl:182-195: rho_t in {0.1, 0.9} on a 5-cell grid"""
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.Grid(X.min(0), X.max(0), 5, 3)
    og = oracle.grid_make(X.min(0), X.max(0), 5, 3)
    _compare(pkg, oracle, X, IEN, rn, rt, pg, og, 1.1, f"sphere rt{rt}")


@pytest.mark.parametrize("bf", [1.1, 2.5])
def test_block(pkg, oracle, bf):
    X, IEN = block_mesh([2, 1, 1])
    rn = np.array([0.0, 0.0, 0.5, 0.5, 0.5, 0.5, 1.0, 1.0, 0.0, 0.0, 0.5, 0.5])
    pg = pkg.Grid(X.min(0), X.max(0), 20, 3)
    og = oracle.grid_make(X.min(0), X.max(0), 20, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, bf, f"block bf{bf}")
    if bf == 2.5:
        sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=2.5)
        assert sdf.max() == pytest.approx(0.4242640687119285, rel=1e-10, abs=1e-12)   # HexBlockSdfTest.jl:25
        assert sdf.mean() == pytest.approx(-1.4699474563515213e9, abs=1e5)            # :26


@pytest.mark.parametrize("name,rt", [("beam_vfrac_03", 0.5), ("beam_vfrac_04", 0.518555), ("chapadlo", 0.5)])
def test_fixture_meshes_auto_grid(pkg, oracle, name, rt):
    """BASELINE configs 2-4 at the reference's automatic grid (solid elements + boundary faces)"""
    X, IEN, rho = load_fixture(name)
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
    og, _ = oracle.auto_grid(X, IEN)
    _compare(pkg, oracle, X, IEN, rn, rt, pg, og, 1.1, name)


@pytest.mark.parametrize("N", [15, 20, 31])
def test_one_hex_el(pkg, oracle, N):
    """the reference's `1hex_el` input (test/runtests.jl:51-86, N = 15): one HEX8 whose iso-surface clips two opposite
    corners of a face.  Symmetric lattice points start the SQP on the stable manifold of saddle points: 17-26 % of the
    pairs see a non-convex model, a third need the second-order correction - the straggler path carries this case"""
    from test_oracle_drift import one_hex
    X, IEN, rn = one_hex()
    pg = pkg.Grid(X.min(0), X.max(0), N, 3)
    og = oracle.grid_make(X.min(0), X.max(0), N, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, 1.1, f"1hex_el N{N}")
    # against the tight-tolerance oracle: the magnitude the round-2 verdict measured (6.3 % at N = 15) is gone
    dist, _ = pkg.evalDistances(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=1.1)
    with oracle.tight():
        d2, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    real = d2 < 1e9
    assert (np.abs(dist[real] - d2[real]) / np.maximum(d2[real], 1e-300)).max() <= 1e-7


@pytest.mark.parametrize("seed,jit", [(1, .30), (2, .35)])
def test_distorted_hexes_random_density(pkg, oracle, seed, jit):
    """strongly distorted elements with independent random nodal densities (several pieces of iso-surface per element):
    half of the pairs leave the fast path (non-convex models, corrections, restorations)"""
    from rho2sdf_jl_amd import synthetic
    X, IEN, _ = synthetic.hex_mesh(7, jitter=jit, seed=20240501 + seed)
    rn = np.clip(np.random.default_rng(seed).normal(0.5, 0.35, len(X)), 0, 1)
    nmax = synthetic.grid_n_max_for_points(48)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, 1.1, f"distorted hexes seed {seed}")


@pytest.mark.parametrize("seed", [100, 101, 102])
def test_random_meshes_and_density_kinds(pkg, oracle, seed):
    """three cases of tools/fuzz_parity.py (160 of them ran bit-equal, profiles/r03_fuzz_parity.txt): random mesh size,
    jitter and grid; densities 0/1 (iso-surfaces on element faces, seed 100), nearly flat around the threshold (101),
    random normal (102) - every voxel of the fused field bit-equal to the oracle"""
    from rho2sdf_jl_amd import synthetic
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 9))
    jit = float(rng.uniform(0.1, 0.38))
    X, IEN, _ = synthetic.hex_mesh(n, jitter=jit, seed=seed)
    kind = seed % 3
    if kind == 0:
        rn = np.clip(rng.normal(0.5, 0.35, len(X)), 0, 1)
    elif kind == 1:
        rn = (rng.random(len(X)) < 0.5).astype(float)
    else:
        rn = np.clip(0.5 + 0.02 * rng.normal(size=len(X)), 0, 1)
    nmax = synthetic.grid_n_max_for_points(int(rng.integers(40, 72)))
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5)
    odist, _, ost = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    want = odist * oracle.sign_detection(X, IEN, rn, 0.5, og)
    assert ost["n_iso_solves"] > 100000
    assert np.array_equal(sdf, want), f"seed {seed}: {int((sdf != want).sum())} voxels differ"


def test_straggler_list_overflow(pkg, oracle):
    """R2S_ISO_STRAGGLER_CAP=64 (read once per process: run in a child): the list of handed-over pairs overflows and
    iso_sweep_kernel finds the unsolved result slots - same field"""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import __graft_entry__ as graft
        from test_oracle_drift import one_hex
        pkg = graft.load_built(); O = graft.load_oracle()
        X, IEN, rn = one_hex()
        pg = pkg.Grid(X.min(0), X.max(0), 20, 3); og = O.grid_make(X.min(0), X.max(0), 20, 3)
        d, _ = pkg.evalDistances(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=1.1)
        od, _, _ = O.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
        assert np.array_equal(d, od), int((d != od).sum())
        print("overflow path OK", int((od < 1e9).sum()))
        # many items: the sweep's chunk -> item search, tile slots and store offsets, and the projection points it writes
        from rho2sdf_jl_amd import synthetic
        X, IEN, _ = synthetic.hex_mesh(7, jitter=.30, seed=20240502)
        rn = np.clip(np.random.default_rng(1).normal(0.5, 0.35, len(X)), 0, 1)
        nmax = synthetic.grid_n_max_for_points(48)
        pg = pkg.Grid(X.min(0), X.max(0), nmax, 3); og = O.grid_make(X.min(0), X.max(0), nmax, 3)
        st = {}
        d, xp = pkg.evalDistances(pkg.Mesh(X, IEN), pg, rn, 0.5, band_factor=1.1, stats=st)
        od, oxp, ost = O.eval_distances(X, IEN, rn, 0.5, og, 1.1)
        real = od < 1e9
        assert np.array_equal(d, od), int((d != od).sum())
        assert np.allclose(xp[real], oxp[real], rtol=0, atol=1e-9) and not xp[~real].any()
        assert st["n_iso_straggler"] > 64 and st["n_iso_fail"] == ost["n_iso_fail"], (st["n_iso_straggler"], st["n_iso_fail"], ost["n_iso_fail"])
        print("overflow path, distorted mesh OK", int(real.sum()), "handed over", st["n_iso_straggler"], "without KKT point", st["n_iso_fail"])
    ''') % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, R2S_ISO_STRAGGLER_CAP="64")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())


def test_overlapping_elements_without_the_inner_region_shortcut(pkg, oracle):
    """two copies of a 3^3 block shifted by 0.37 of an element against each other, the second with other densities: a
    NON-conforming mesh of overlapping elements.  The sign pass's inner-region shortcut assumes a conforming mesh
    (include/rho2sdf_hip.h); with R2S_SIGN_NO_INNER=1 every candidate pair runs its inverse map and the ordered walk of
    SignDetection.jl:41-68 must come out exactly as the oracle's (run in a child: the switch is read once per process)"""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import __graft_entry__ as graft
        from conftest import block_mesh
        pkg = graft.load_built(); O = graft.load_oracle()
        X1, I1 = block_mesh([3, 3, 3])
        X2 = X1 + 0.37 * (2.0 / 3.0) * np.array([1.0, 0.6, -0.8])
        X = np.vstack([X1, X2]); IEN = np.vstack([I1, I1 + len(X1)])
        rng = np.random.default_rng(5)
        rn = np.concatenate([np.clip(1.2 - np.linalg.norm(X1, axis=1), 0, 1), rng.uniform(0.0, 1.0, len(X2))])
        pg = pkg.Grid(X.min(0), X.max(0), 30, 3); og = O.grid_make(X.min(0), X.max(0), 30, 3)
        s = pkg.Sign_Detection(pkg.Mesh(X, IEN), pg, rn, 0.5)
        so = O.sign_detection(X, IEN, rn, 0.5, og)
        assert np.array_equal(s, so), int((s != so).sum())
        print("overlapping mesh: signs equal,", int((so > 0).sum()), "of", so.size, "positive")
    ''') % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, R2S_SIGN_NO_INNER="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    # the same switch per call (r2s_params.sign_no_inner), in this process
    X1, I1 = block_mesh([3, 3, 3])
    X2 = X1 + 0.37 * (2.0 / 3.0) * np.array([1.0, 0.6, -0.8])
    X = np.vstack([X1, X2]); IEN = np.vstack([I1, I1 + len(X1)])
    rng = np.random.default_rng(5)
    rn = np.concatenate([np.clip(1.2 - np.linalg.norm(X1, axis=1), 0, 1), rng.uniform(0.0, 1.0, len(X2))])
    pg = pkg.Grid(X.min(0), X.max(0), 30, 3); og = oracle.grid_make(X.min(0), X.max(0), 30, 3)
    so = oracle.sign_detection(X, IEN, rn, 0.5, og)
    assert np.array_equal(pkg.Sign_Detection(pkg.Mesh(X, IEN), pg, rn, 0.5, sign_no_inner=True), so)
    od, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    assert np.array_equal(pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, sign_no_inner=True), od * so)


def test_synthetic_jittered_hex(pkg, oracle):
    """north-star mesh family at a size the oracle finishes in seconds: 12^3 HEX8, 64^3 grid"""
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = synthetic.hex_mesh(12)
    nmax = synthetic.grid_n_max_for_points(64)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    assert pg.dims == (64, 64, 64)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, 1.1, "synthetic 12^3 / 64^3")


def test_slabs_equal_full_volume(pkg, oracle):
    """Z-slab runs (the multi-GPU partition) stitched together == one full-volume run"""
    import torch
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = synthetic.hex_mesh(8)
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(40), 3)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    nx, ny, nz = pg.dims
    full = torch.empty(nz * ny * nx, dtype=torch.float64, device=dev)
    plan.run(dX, dI, dR, 0.5, pg, sdf=full)
    for parts in (2, 3, 8):
        bounds = [round(nz * p / parts) for p in range(parts + 1)]
        pieces = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            out = torch.empty((b - a) * ny * nx, dtype=torch.float64, device=dev)
            plan.run(dX, dI, dR, 0.5, pg, k_begin=a, k_end=b, sdf=out)
            pieces.append(out)
        assert torch.equal(torch.cat(pieces), full), f"{parts} slabs differ from the full volume"
    plan.close()


# ---------------------------------------------------------------------------------------
# TET4 (the reference's only TET4 coverage is PrimitiveGeometriesTest/SimpleCubeWithSchlafli.jl)
# ---------------------------------------------------------------------------------------
def test_tet4_radial_cube(pkg, oracle):
    """SimpleCubeWithSchlafli.jl:19-143: 10^3 cube of side 10 split 6-way, rho_n = 1 - r/(5 sqrt 3), rho_t = 0.5"""
    from rho2sdf_jl_amd import synthetic
    X, IH, rn = synthetic.radial_cube(10, 10.0)
    IT = synthetic.hex_to_tets(IH)
    pg = pkg.Grid(X.min(0), X.max(0), 40, 3)
    og = oracle.grid_make(X.min(0), X.max(0), 40, 3)
    _compare(pkg, oracle, X, IT, rn, 0.5, pg, og, 1.1, "tet4 radial cube")
    # analytic cross-check: HEX8 and TET4 discretisations of the same field agree to O(h^2)
    dh, _ = pkg.evalDistances(pkg.Mesh(X, IH), pg, rn, 0.5, want_xp=False)
    dt, _ = pkg.evalDistances(pkg.Mesh(X, IT), pg, rn, 0.5, want_xp=False)
    assert np.array_equal(dh == 1e10, dt == 1e10)
    both = dh < 1e9
    assert np.abs(dh[both] - dt[both]).max() < 0.1          # h = 1


@pytest.mark.parametrize("bf", [1.1, 2.5])
def test_tet4_jittered(pkg, oracle, bf):
    """BASELINE config 5 family (jittered Schlafli tets) at oracle size: 6^3*6 tets, 48^3 grid"""
    from rho2sdf_jl_amd import synthetic
    X, IT, rn = synthetic.tet_mesh(6)
    nmax = synthetic.grid_n_max_for_points(48)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IT, rn, 0.5, pg, og, bf, f"tet4 jittered bf{bf}")


def test_tet4_solid_boundary(pkg, oracle):
    """solid tets with boundary faces + iso tets with validated boundary triangles (rho_t low)"""
    from rho2sdf_jl_amd import synthetic
    X, IH, rn = synthetic.radial_cube(6, 6.0)
    IT = synthetic.hex_to_tets(IH)
    pg = pkg.Grid(X.min(0), X.max(0), 24, 3)
    og = oracle.grid_make(X.min(0), X.max(0), 24, 3)
    _compare(pkg, oracle, X, IT, rn, 0.05, pg, og, 1.1, "tet4 solid boundary")


@pytest.mark.parametrize("squash,seed", [(0.05, 1), (0.02, 2), (5e-4, 3)])
def test_tet4_flat_elements_random_density(pkg, oracle, squash, seed):
    """strongly jittered Schlafli tets whose upper half is squashed in z (flat elements: height / edge down to 5e-4, below
    the 1e-3 limit under which the face-plane shortcuts of the sign pass switch themselves off) with RANDOM nodal
    densities: the tile culling, the accept / reject planes and the unordered candidate lists must leave every sign,
    sentinel and distance as the oracle has them"""
    from rho2sdf_jl_amd import synthetic
    X, IT, _ = synthetic.tet_mesh(5, jitter=0.3, seed=seed)
    X = X.copy()
    up = X[:, 2] > 0
    X[up, 2] *= squash
    rn = np.random.default_rng(seed).uniform(0.0, 1.0, len(X))
    nmax = synthetic.grid_n_max_for_points(44)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IT, rn, 0.5, pg, og, 1.1, f"tet4 flat {squash}")


@pytest.mark.parametrize("dims_n,tets", [(40, False), (37, False), (37, True)])
def test_interleaved_layers_equal_full_volume(pkg, oracle, dims_n, tets):
    """the balanced multi-GPU partition (4-plane tile layers dealt round-robin, r2s_params.zstride/zphase):
    all ranks' parts, gathered and reordered exactly as bench.py does, == one full-volume run (HEX8 and the
    TET4 family of BASELINE config 5)"""
    import torch
    from rho2sdf_jl_amd import slabs, synthetic
    X, IEN, rn = synthetic.tet_mesh(6) if tets else synthetic.hex_mesh(8)
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(dims_n), 3)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    nx, ny, nz = pg.dims
    full = torch.empty(nz * ny * nx, dtype=torch.float64, device=dev)
    plan.run(dX, dI, dR, 0.5, pg, sdf=full)
    for world in (2, 3, 8):
        sgs = [slabs.SlabGather((nx, ny, nz), r, world, dev, interleaved=True) for r in range(world)]
        for r, sg in enumerate(sgs):
            if sg.my_planes:
                plan.run(dX, dI, dR, 0.5, pg, sdf=sg.my_slab, zstride=world, zphase=r)
        # emulate the all-gather: every rank's `mine` into rank 0's buffer
        for r, sg in enumerate(sgs[1:], start=1):
            sgs[0].gathered[r * sgs[0].per * sgs[0].plane:(r + 1) * sgs[0].per * sgs[0].plane].copy_(sg.mine)
        assert torch.equal(sgs[0].volume().reshape(-1), full), f"{world} interleaved parts differ from the full volume"
    plan.close()


def test_sparse_tile_stitching_equals_full_volume(pkg, oracle):
    """sparse multi-GPU stitching on one GPU: every rank's interleaved part is packed into tiles
    (r2s_plan_pack_tiles_dev), scattered into a sentinel-filled volume (r2s_fill_dev / r2s_unpack_tiles_dev),
    and the result must equal the full-volume run bit for bit"""
    import torch
    from rho2sdf_jl_amd import slabs, synthetic
    X, IEN, rn = synthetic.hex_mesh(8)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    for npts in (40, 37):
        pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(npts), 3)
        nx, ny, nz = pg.dims
        full = torch.empty(nz * ny * nx, dtype=torch.float64, device=dev)
        plan.run(dX, dI, dR, 0.5, pg, sdf=full)
        for world in (2, 3, 8):
            vol = torch.empty_like(full)
            plan.fill(vol, -1.0e10)
            moved = 0
            for r in range(world):
                owned, _ = slabs.interleaved_layers(nz, world, r)
                if not owned:
                    continue
                local = torch.empty(4 * owned * ny * nx, dtype=torch.float64, device=dev)
                st = plan.run(dX, dI, dR, 0.5, pg, sdf=local, zstride=world, zphase=r)
                n = st["n_any_tiles"]
                payload = torch.empty(max(n, 1) * 64, dtype=torch.float64, device=dev)
                ids = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
                assert plan.pack_tiles(local, payload, ids) == n
                plan.unpack_tiles(payload, ids, n, pg, vol)
                moved += n
            torch.cuda.synchronize()
            assert torch.equal(vol, full), f"{world} ranks, {npts}^3: sparse stitching differs"
            ntiles = ((nx + 3) // 4) * ((ny + 3) // 4) * ((nz + 3) // 4)
            assert moved < ntiles       # fewer tiles than the dense volume
    plan.close()


@pytest.mark.parametrize("tets", [False, True])
def test_compressed_tile_stitching_equals_full_volume(pkg, oracle, tets):
    """compressed stitching (r2s_plan_pack_tiles2_dev / r2s_unpack_masks_dev): band tiles travel as 64 doubles,
    sign-only tiles as one 64-bit mask; scattering every rank's share must reproduce the full-volume run bit for bit"""
    import torch
    from rho2sdf_jl_amd import slabs, synthetic
    X, IEN, rn = synthetic.tet_mesh(8) if tets else synthetic.hex_mesh(8)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    for npts in (120, 37):
        pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(npts), 3)
        nx, ny, nz = pg.dims
        full = torch.empty(nz * ny * nx, dtype=torch.float64, device=dev)
        plan.run(dX, dI, dR, 0.5, pg, sdf=full)
        for world in (2, 3, 8):
            vol = torch.empty_like(full)
            plan.fill(vol, -1.0e10)
            n_full = n_mask = 0
            segs = []
            for r in range(world):
                owned, _ = slabs.interleaved_layers(nz, world, r)
                if not owned:
                    segs.append((0, 0, None, None, None, None))
                    continue
                local = torch.empty(4 * owned * ny * nx, dtype=torch.float64, device=dev)
                st = plan.run(dX, dI, dR, 0.5, pg, sdf=local, zstride=world, zphase=r)
                nf, nm = st["n_active_tiles"], st["n_sign_only_tiles"]
                assert nf + nm == st["n_any_tiles"]
                payload = torch.empty(max(nf, 1) * 64, dtype=torch.float64, device=dev)
                ids = torch.zeros(max(nf, 1), dtype=torch.int32, device=dev)
                masks = torch.zeros(max(nm, 1), dtype=torch.int64, device=dev)
                mids = torch.zeros(max(nm, 1), dtype=torch.int32, device=dev)
                assert plan.pack_tiles2(local, payload, ids, masks, mids) == (nf, nm)
                plan.unpack_tiles(payload, ids, nf, pg, vol)
                plan.unpack_masks(masks, mids, nm, pg, vol)
                segs.append((nf, nm, payload, ids, masks, mids))
                n_full += nf
                n_mask += nm
            torch.cuda.synchronize()
            assert torch.equal(vol, full), f"{world} ranks, {npts}^3: compressed stitching differs"
            assert n_mask > 0 or npts < 100     # the finer grid has tiles deep inside the solid
            # the same exchange as ONE buffer of `world` segments scattered by r2s_unpack_segments_dev (counts read from
            # the segment headers on the device), laid out exactly as slabs.SlabGather lays it out
            mf, mm = max(segs, key=lambda c: c[0])[0] + 3, max(segs, key=lambda c: c[1])[1] + 2
            seglen = 2 + mf * 64 + (mf + 1) // 2 + mm + (mm + 1) // 2
            buf = torch.zeros(world * seglen, dtype=torch.float64, device=dev)
            for r, (nf, nm, payload, ids, masks, mids) in enumerate(segs):
                seg = buf[r * seglen:(r + 1) * seglen]
                seg[:2].view(torch.int64).copy_(torch.tensor([nf, nm]))
                if payload is None:
                    continue
                p2, i2, m2, mi2 = slabs.SlabGather._segment_views(seg[2:], mf, mm)
                p2[:nf * 64].copy_(payload[:nf * 64]); i2[:nf].copy_(ids[:nf]); m2[:nm].copy_(masks[:nm]); mi2[:nm].copy_(mids[:nm])
            vol2 = torch.empty_like(full)
            plan.fill(vol2, -1.0e10)
            plan.unpack_segments(buf, world, seglen, mf, mm, pg, vol2)
            torch.cuda.synchronize()
            assert torch.equal(vol2, full), f"{world} ranks, {npts}^3: segment-wide scatter differs"
    plan.close()


@pytest.mark.parametrize("seed,jitter,bf", [(1, 0.30, 1.1), (2, 0.35, 2.5), (3, 0.25, 1.1)])
def test_distorted_hex_random_density(pkg, oracle, seed, jitter, bf):
    """stress case for the restated solvers and the exact pruning of the sign pass: strongly distorted HEX8
    (node jitter up to 35 % of the cell, warped faces) with RANDOM nodal densities - iso-surfaces cut elements
    at arbitrary angles, many projections end on faces / edges / corners of the reference cube, the
    active-set walk and the line search take their rare paths.  Distances, signs and the sentinel set must
    still equal the oracle's bit for bit."""
    from rho2sdf_jl_amd import synthetic
    X, IEN, _ = synthetic.hex_mesh(7, jitter=jitter, seed=20240501 + seed)
    rng = np.random.default_rng(seed)
    rn = np.clip(rng.normal(0.5, 0.35, len(X)), 0.0, 1.0)
    nmax = synthetic.grid_n_max_for_points(48)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, bf, f"distorted hex, seed {seed}")


@pytest.mark.parametrize("fillv", [0.0, 1.0])
def test_uniform_density(pkg, oracle, fillv):
    """degenerate work lists: an all-void mesh (no items, no hot tiles: the output is the sentinel everywhere) and an
    all-solid one (boundary triangles only, no iso items; every tile of the mesh is hot)"""
    from rho2sdf_jl_amd import synthetic
    X, IEN, _ = synthetic.hex_mesh(5, jitter=0.1)
    rn = np.full(len(X), fillv)
    nmax = synthetic.grid_n_max_for_points(36)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, 1.1, f"uniform density {fillv}")
    if fillv == 0.0:
        assert np.all(pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5) == -1.0e10)


def test_chapadlo_config4_grid(pkg, oracle):
    """BASELINE config 4 (robot gripper, "256^3"): N_max = 249 gives the reference's 87 x 166 x 257 grid
    (SURVEY 8(d)); the four contiguous Z-slabs of the 4-GPU partition (257 planes -> 65 per rank) are computed one
    by one and must equal the oracle's full volume bit for bit"""
    import torch
    X, IEN, rho = load_fixture("chapadlo")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    pg = pkg.Grid(X.min(0), X.max(0), 249, 3)
    og = oracle.grid_make(X.min(0), X.max(0), 249, 3)
    assert pg.dims == (87, 166, 257)
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    ref = d * oracle.sign_detection(X, IEN, rn, 0.5, og)
    from rho2sdf_jl_amd import slabs
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    nx, ny, nz = pg.dims
    per, bounds = slabs.slab_bounds(nz, 4)
    assert per == 65
    pieces = []
    for k0, k1 in bounds:
        out = torch.empty((k1 - k0) * ny * nx, dtype=torch.float64, device=dev)
        plan.run(dX, dI, dR, 0.5, pg, k_begin=k0, k_end=k1, sdf=out)
        pieces.append(out)
    got = torch.cat(pieces).cpu().numpy()
    plan.close()
    assert np.array_equal(np.abs(got) > 1e9, np.abs(ref) > 1e9)
    assert np.array_equal(np.sign(got), np.sign(ref))
    real = np.abs(ref) < 1e9
    rel = np.abs(got[real] - ref[real]) / np.maximum(np.abs(ref[real]), 1e-300)
    assert not ((rel > RTOL) & (np.abs(got[real] - ref[real]) > 1e-12 * og.cell)).any()
    print(f"chapadlo 87x166x257: bit-equal {int((got == ref).sum())}/{got.size}")


def test_mesh_finer_than_grid(pkg, oracle):
    """a grid much coarser than the mesh: tile lists hold hundreds of items / candidates (the > 64-entry paths of
    the list sort and of the lane-parallel list walk in the gather)"""
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = synthetic.hex_mesh(20)
    nmax = synthetic.grid_n_max_for_points(24)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    st = {}
    pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5, stats=st)
    assert st["n_sign_entries"] / max(st["n_active_sign_tiles"], 1) > 64
    _compare(pkg, oracle, X, IEN, rn, 0.5, pg, og, 1.1, "mesh 20^3 on a 24^3 grid")


def test_elements_much_larger_than_cells(pkg, oracle):
    """8 elements on a 340^3 grid: every item box holds > 2^22 lattice points (the integer-division path of the
    box enumeration, thousands of 64-voxel chunks per item).  Checked on every 48th Z plane (the oracle's plane
    sampling, as in bench.py)."""
    from rho2sdf_jl_amd import synthetic
    X, IEN, _ = synthetic.hex_mesh(2, jitter=0.1)
    rn = np.clip(1.2 - np.linalg.norm(X, axis=1), 0.0, 1.0)
    nmax = synthetic.grid_n_max_for_points(340)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    nx, ny, nz = pg.dims
    sdf = pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5).reshape(nz, ny, nx)
    stride = 48
    oracle.set_k_sampling(stride, 0)
    try:
        d, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
        ref = (d * oracle.sign_detection(X, IEN, rn, 0.5, og)).reshape(nz, ny, nx)[::stride]
    finally:
        oracle.set_k_sampling(1, 0)
    got = sdf[::stride]
    assert np.array_equal(np.abs(got) > 1e9, np.abs(ref) > 1e9) and np.array_equal(np.sign(got), np.sign(ref))
    real = np.abs(ref) < 1e9
    assert real.sum() > 1000
    rel = np.abs(got[real] - ref[real]) / np.maximum(np.abs(ref[real]), 1e-300)
    assert not ((rel > RTOL) & (np.abs(got[real] - ref[real]) > 1e-12 * og.cell)).any()
    print(f"large elements: {int(real.sum())} band voxels on the sampled planes, bit-equal {int((got == ref).sum())}/{got.size}")


def test_speculated_sizes_fall_back_when_the_data_changes(pkg, oracle):
    """a plan remembers the list / work-array sizes of its last call and launches the next call of the same shapes
    without waiting for them (run_impl, read_back_kernel); when the data behind the same shapes changes (another
    threshold, another density field, a connectivity error) the device-side check must stop the call before anything
    is written outside its buffers and the call is repeated the slow way - results as from a fresh plan"""
    import torch
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = synthetic.hex_mesh(9)
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(72), 3)
    dev = torch.device("cuda:0")
    dX, dI = torch.from_numpy(X).to(dev), torch.from_numpy(IEN).to(dev)
    rng = np.random.default_rng(11)
    fields = [rn, rn, np.clip(rn + rng.normal(0, 0.2, len(rn)), 0, 1), rn * 0.0, np.clip(rng.normal(0.5, 0.3, len(rn)), 0, 1), rn]
    thresholds = [0.5, 0.5, 0.5, 0.5, 0.35, 0.62]
    plan = pkg.DevicePlan(0)
    out = torch.empty(pg.ngp, dtype=torch.float64, device=dev)
    for f, rt in zip(fields, thresholds):
        dR = torch.from_numpy(np.ascontiguousarray(f)).to(dev)
        plan.run(dX, dI, dR, rt, pg, sdf=out)
        fresh = pkg.DevicePlan(0)
        want = torch.empty_like(out)
        fresh.run(dX, dI, dR, rt, pg, sdf=want)
        fresh.close()
        assert torch.equal(out, want)
    # a connectivity error behind remembered sizes is still reported, and the plan keeps working afterwards
    bad = IEN.copy()
    bad[5, 2] = 10 ** 7
    with pytest.raises(pkg._lib.R2SError, match="outside 1..nnp"):
        plan.run(dX, torch.from_numpy(bad).to(dev), torch.from_numpy(rn).to(dev), 0.62, pg, sdf=out)
    plan.run(dX, dI, torch.from_numpy(rn).to(dev), 0.62, pg, sdf=out)
    assert torch.equal(out, want)
    plan.close()
