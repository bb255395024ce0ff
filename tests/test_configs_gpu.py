"""BASELINE.json configs 2, 3 and 5 as compositions, at their real sizes, through the C ABI, against the oracle.

config 2: cantilever_beam_vfrac_03, sdf_grid_setup = :automatic, threshold from the bisection, rbf_interp = false
config 3: cantilever_beam_vfrac_04, threshold 0.518555, rbf_interp = true, rbf_grid = :fine
config 5: 998 250 jittered Schlafli TET4, 1024^3 grid, the eight interleaved rank shares of the 8-GPU partition
          computed one after the other on ONE GPU, reassembled, compared with the oracle on sampled planes
(config 1 = test_parity_gpu.py::test_sphere, config 4 = test_parity_gpu.py::test_chapadlo_config4_grid, north star =
bench.py's default check.)
"""
import numpy as np
import pytest

from conftest import load_fixture

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _assert_field(got, ref, cell, label):
    assert np.array_equal(np.abs(got) > 1e9, np.abs(ref) > 1e9), f"{label}: sentinel set differs"
    assert np.array_equal(np.sign(got), np.sign(ref)), f"{label}: sign differs"
    real = np.abs(ref) < 1e9
    rel = np.abs(got[real] - ref[real]) / np.maximum(np.abs(ref[real]), 1e-300)
    bad = (rel > RTOL) & (np.abs(got[real] - ref[real]) > 1e-12 * cell)
    assert not bad.any(), f"{label}: {int(bad.sum())} distances beyond {RTOL} (max rel {rel.max()})"
    return int((got == ref).sum()), int(real.sum()), float(rel.max()) if rel.size else 0.0


def test_config2_end_to_end(pkg, oracle):
    """runtests.jl:186-207 with the options of BASELINE config 2, stage by stage against the oracle"""
    X, IEN, rho = load_fixture("beam_vfrac_03")
    opts = pkg.Rho2sdfOptions(sdf_grid_setup="automatic", rbf_interp=False)      # threshold_density = nothing
    info = {}
    fine_sdf, fine_grid, sdf_grid, sdf_dists = pkg.rho2sdf("beam", X, IEN, rho, options=opts, info=info)
    # mesh volume, nodal densities, threshold
    ovd, ovf = oracle.mesh_volume(X, IEN, rho)
    assert info["V_domain"] == pytest.approx(ovd, rel=1e-12) and info["V_frac"] == pytest.approx(ovf, rel=1e-12)
    orn = oracle.dense_in_nodes(X, IEN, rho)
    assert np.abs(info["rho_n"] - orn).max() <= 1e-12
    ort, oit = oracle.find_threshold(X, IEN, orn, ovd * ovf)
    assert info["rho_t"] == ort and info["threshold_iters"] == oit
    # raw SDF + artifact removal
    og, _ = oracle.auto_grid(X, IEN)
    assert sdf_grid.dims == og.dims == (67, 27, 11)
    d, _, _ = oracle.eval_distances(X, IEN, orn, ort, og, 1.1, want_xp=False)
    ref = d * oracle.sign_detection(X, IEN, orn, ort, og)
    nflip = oracle.remove_artifacts(ref, og)
    assert info["n_flipped"] == nflip
    eq, nreal, mx = _assert_field(sdf_dists, ref, og.cell, "config 2 sdf_dists")
    # RBF approximation on the same grid
    ofine, oth, _, olsf = oracle.rbf_smoothing(ref, og, False, 1, ovd * ovf)
    assert fine_sdf.shape == ofine.shape == (11, 27, 67)
    scale = np.abs(olsf).max()
    assert abs(info["level_shift"] - oth) <= 1e-3 * scale
    assert np.abs((fine_sdf - np.float32(info["level_shift"])) - (ofine - np.float32(oth))).max() <= 2e-6 * scale
    print(f"config 2: rho_t {info['rho_t']} ({oit} its), flipped {nflip}, sdf bit-equal {eq}/{sdf_dists.size} "
          f"({nreal} band voxels, max rel {mx:.2e}), th {info['level_shift']} vs {oth}")


def test_config3_end_to_end(pkg, oracle):
    """BASELINE config 3: beam_vfrac_04, threshold 0.518555 (runtests.jl:198), CG interpolation on the :fine grid"""
    X, IEN, rho = load_fixture("beam_vfrac_04")
    opts = pkg.Rho2sdfOptions(threshold_density=0.518555, sdf_grid_setup="automatic", rbf_interp=True, rbf_grid="fine")
    info = {}
    fine_sdf, fine_grid, sdf_grid, sdf_dists = pkg.rho2sdf("beam", X, IEN, rho, options=opts, info=info)
    og, _ = oracle.auto_grid(X, IEN)
    orn = oracle.dense_in_nodes(X, IEN, rho)
    d, _, _ = oracle.eval_distances(X, IEN, orn, 0.518555, og, 1.1, want_xp=False)
    ref = d * oracle.sign_detection(X, IEN, orn, 0.518555, og)
    oracle.remove_artifacts(ref, og)
    _assert_field(sdf_dists, ref, og.cell, "config 3 sdf_dists")
    ovd, ovf = oracle.mesh_volume(X, IEN, rho)
    ofine, oth, oits, olsf = oracle.rbf_smoothing(ref, og, True, 2, ovd * ovf)
    assert fine_sdf.shape == ofine.shape == (21, 53, 133)
    scale = np.abs(olsf).max()
    assert abs(info["cg_iters"] - oits) <= 1
    assert abs(info["level_shift"] - oth) <= 1e-3 * scale
    assert np.abs((fine_sdf - np.float32(info["level_shift"])) - (ofine - np.float32(oth))).max() <= 5e-5 * scale
    assert fine_grid[2] == (133, 53, 21)


def test_config5_full_size(pkg, oracle):
    """BASELINE config 5 at its real size on one GPU: 998 250 TET4 (55^3 cells split 6-way), 1024^3 grid, the eight
    rank shares of the interleaved Z partition (zstride = 8) reassembled into the whole 8.6 GB volume; it must equal
    a single full-volume run bit for bit, and the oracle on 16 sampled planes (every rank's share and every position
    inside a 4-plane tile layer are hit)."""
    import torch
    from rho2sdf_jl_amd import slabs, synthetic
    X, IT, rn = synthetic.tet_mesh(55)
    assert len(IT) == 998250
    nmax = synthetic.grid_n_max_for_points(1024)
    pg = pkg.Grid(X.min(0), X.max(0), nmax, 3)
    og = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
    nx, ny, nz = pg.dims
    assert (nx, ny, nz) == (1024, 1024, 1024)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IT, rn))
    plan = pkg.DevicePlan(0)
    world = 8
    vol = torch.empty(nz, ny, nx, dtype=torch.float64, device=dev)
    layers = nz // 4
    for r in range(world):
        owned, _ = slabs.interleaved_layers(nz, world, r)
        local = torch.empty(owned * 4 * ny * nx, dtype=torch.float64, device=dev)
        plan.run(dX, dI, dR, 0.5, pg, sdf=local, zstride=world, zphase=r)
        # layer i of rank r is global layer i*world + r
        vol.view(layers // world, world, 4 * ny * nx)[:, r].copy_(local.view(owned, 4 * ny * nx))
        del local
    full = torch.empty(nz * ny * nx, dtype=torch.float64, device=dev)
    st = plan.run(dX, dI, dR, 0.5, pg, sdf=full)
    assert torch.equal(full, vol.view(-1)), "eight interleaved shares differ from the single full-volume run"
    del full
    stride, phase = 67, 5
    ks = list(range(phase, nz, stride))
    assert {(k // 4) % world for k in ks} == set(range(world)) and {k % 4 for k in ks} == {0, 1, 2, 3}
    got = vol[phase::stride].cpu().numpy()
    plan.close()
    del vol
    oracle.set_k_sampling(stride, phase)
    try:
        d, _, _ = oracle.eval_distances(X, IT, rn, 0.5, og, 1.1, want_xp=False)
        s = oracle.sign_detection(X, IT, rn, 0.5, og)
        ref = d.reshape(nz, ny, nx)[phase::stride] * s.reshape(nz, ny, nx)[phase::stride]
    finally:
        oracle.set_k_sampling(1, 0)
    eq, nreal, mx = _assert_field(got, ref, og.cell, "config 5")
    assert nreal > 100000
    print(f"config 5: {len(IT)} TET4, {nx}x{ny}x{nz}; 8 shares == full volume; {len(ks)} planes vs oracle: bit-equal {eq}/{got.size}, "
          f"{nreal} band voxels, max rel {mx:.2e}; items {st['n_items']}, active tiles {st['n_active_tiles']}")


def test_tet4_threshold_and_iso_volume(pkg, oracle):
    """TET4 iso-volume + volume-preserving threshold (the reference's search is HEX8-only)"""
    from rho2sdf_jl_amd import synthetic
    X, IT, rn = synthetic.tet_mesh(8)
    mesh = pkg.Mesh(X, IT)
    rho = np.clip(rn[IT - 1].mean(axis=1), 0, 1)
    vd, vf = pkg.calculate_mesh_volume(mesh, rho)
    assert pkg.calculate_isocontour_volume(mesh, rn, 0.0) == pytest.approx(vd, rel=1e-12)
    for thr in (0.25, 0.5, 0.9):
        assert pkg.calculate_isocontour_volume(mesh, rn, thr) == pytest.approx(oracle.isocontour_volume(X, IT, rn, thr), rel=1e-12)
    rt = pkg.find_threshold_for_volume(mesh, rn, vd * vf)
    ort, _ = oracle.find_threshold(X, IT, rn, vd * vf)
    assert rt == ort
    with pytest.raises(pkg._lib.R2SError, match="outside the possible range"):
        pkg.find_threshold_for_volume(mesh, rn, vd * 2.0)
    # and the whole chain on tetrahedra with the automatic threshold
    opts = pkg.Rho2sdfOptions(rbf_interp=False, element_type=pkg._lib.TET4)
    g = pkg.Grid(X.min(0), X.max(0), 30, 3)
    info = {}
    fine, _, _, sd = pkg.rho2sdf("tets", X, IT, rho, options=opts, sdf_grid=g, info=info)
    orn = oracle.dense_in_nodes(X, IT, rho)
    ort2, _ = oracle.find_threshold(X, IT, orn, vd * vf)
    assert info["rho_t"] == ort2
    og = oracle.grid_make(X.min(0), X.max(0), 30, 3)
    d, _, _ = oracle.eval_distances(X, IT, orn, ort2, og, 1.1, want_xp=False)
    ref = d * oracle.sign_detection(X, IT, orn, ort2, og)
    oracle.remove_artifacts(ref, og)
    _assert_field(sd, ref, og.cell, "tet chain")


@pytest.mark.parametrize("pinned", [True, False])
def test_host_entry_points_reuse_their_session(pkg, oracle, pinned):
    """r2s_sdf / r2s_eval_distances / r2s_sign_detection on host pointers keep plan + buffers between calls
    (r2s_release_cache frees them): big -> small -> big sequences, pinned (r2s_host_alloc) and pageable
    destinations, and a destination larger than one staging chunk must all give the device-path result."""
    import torch
    from rho2sdf_jl_amd import synthetic
    dev = torch.device("cuda:0")
    plan = pkg.DevicePlan(0)
    cases = [(synthetic.hex_mesh(10), 200), (synthetic.hex_mesh(4), 24), (synthetic.tet_mesh(5), 90), (synthetic.hex_mesh(10), 200)]
    for (X, IEN, rn), npts in cases:
        pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(npts), 3)
        dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
        want = torch.empty(pg.ngp, dtype=torch.float64, device=dev)
        wd = torch.empty_like(want)
        ws = torch.empty_like(want)
        plan.run(dX, dI, dR, 0.5, pg, sdf=want)
        plan.run(dX, dI, dR, 0.5, pg, dist=wd, sign=ws)
        mesh = pkg.Mesh(X, IEN)
        alloc = pkg.host_array if pinned else np.empty
        out = alloc(pg.ngp)
        got = pkg.sdf_fused(mesh, pg, rn, 0.5, out=out)
        assert got is out and np.array_equal(got, want.cpu().numpy())
        d, _ = pkg.evalDistances(mesh, pg, rn, 0.5, want_xp=False, out=alloc(pg.ngp))
        s = pkg.Sign_Detection(mesh, pg, rn, 0.5, out=alloc(pg.ngp))
        assert np.array_equal(d, wd.cpu().numpy()) and np.array_equal(s, ws.cpu().numpy())
    plan.close()
    pkg._lib.lib().r2s_release_cache()
    X, IEN, rn = synthetic.hex_mesh(4)
    pg = pkg.Grid(X.min(0), X.max(0), 17, 3)
    assert np.isfinite(pkg.sdf_fused(pkg.Mesh(X, IEN), pg, rn, 0.5)).all()     # sessions come back after a release


@pytest.mark.parametrize("tets,npts", [(False, 173), (True, 170)])
def test_host_pointer_sparse_download(pkg, tets, npts):
    """r2s_sdf on grids of more than 4 M voxels brings the fused field into the caller's array by the sparse download
    (host threads write the sentinel while the device works, only the 4x4x4 tiles that can differ from it are
    transferred and scattered: run_host_device in r2s_host.hip).  Bit-equal to the device path, into pinned and into
    ordinary memory, on grids whose sizes are no multiples of the tile edge, twice in a row (the second call reuses the
    landing zone and speculates the sizes)."""
    import torch
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = (synthetic.tet_mesh if tets else synthetic.hex_mesh)(9)
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(npts), 3)
    assert pg.ngp > (1 << 22) and any(d % 4 for d in pg.dims)
    dev = torch.device("cuda:0")
    dX, dI, dR = (torch.from_numpy(a).to(dev) for a in (X, IEN, rn))
    plan = pkg.DevicePlan(0)
    want = torch.empty(pg.ngp, dtype=torch.float64, device=dev)
    plan.run(dX, dI, dR, 0.5, pg, sdf=want)
    want = want.cpu().numpy()
    plan.close()
    assert (np.abs(want) < 1e9).sum() > 10000 and (want == 1.0e10).any() and (want == -1.0e10).any()
    mesh = pkg.Mesh(X, IEN)
    for alloc in (pkg.host_array, np.empty):
        out = alloc(pg.ngp)
        for _ in range(2):
            out[:] = 7.0                                     # nothing of the previous call may survive
            got = pkg.sdf_fused(mesh, pg, rn, 0.5, out=out)
            assert got is out and np.array_equal(got, want)


@pytest.mark.parametrize("G", [2, 3, 8])
def test_single_process_fan_out(pkg, oracle, G, monkeypatch):
    """r2s_params.n_gpus / r2s_options.n_gpus: ONE call fans out over G devices (one host thread each, interleaved
    tile layers, every device's layers sent straight to their place in the caller's array; r2s_rho2sdf moves them to
    contiguous slabs by peer copies and post-processes slab-distributed).  A one-GPU box has one device, so the test hook R2S_MULTI_OVERSUBSCRIBE maps the G
    logical devices onto it (separate sessions, plans and buffers): the partition, the threads and the copies are the
    ones an 8-GPU node runs.  Results must equal the single-device call bit for bit."""
    from rho2sdf_jl_amd import synthetic
    X, IEN, rn = synthetic.hex_mesh(8)
    pg = pkg.Grid(X.min(0), X.max(0), synthetic.grid_n_max_for_points(70), 3)     # 70 planes: a partial last layer
    mesh = pkg.Mesh(X, IEN)
    want = pkg.sdf_fused(mesh, pg, rn, 0.5)
    wd, _ = pkg.evalDistances(mesh, pg, rn, 0.5, want_xp=False)
    with pytest.raises(pkg._lib.R2SError, match="visible"):
        pkg.sdf_fused(mesh, pg, rn, 0.5, n_gpus=G + 100)
    monkeypatch.setenv("R2S_MULTI_OVERSUBSCRIBE", "1")
    for out in (None, pkg.host_array(pg.ngp)):
        got = pkg.sdf_fused(mesh, pg, rn, 0.5, n_gpus=G, out=out)
        assert np.array_equal(got, want)
    d, xp = pkg.evalDistances(mesh, pg, rn, 0.5, n_gpus=G)
    d1, xp1 = pkg.evalDistances(mesh, pg, rn, 0.5)
    assert np.array_equal(d, wd) and np.array_equal(xp, xp1)
    # the chained call: raw SDF on interleaved tile layers, then connected components (labels merged over the slab
    # interfaces), CG / RBF evaluation with halo exchanges and the level bisection on Z-slabs over the G devices.
    # A noisy density field gives dozens of small components (some of them cut by slab interfaces) to remove.
    rng = np.random.default_rng(17)
    rho = np.clip(rn[IEN - 1].mean(axis=1) + rng.normal(0, 0.35, len(IEN)), 0, 1)
    for interp, smooth in ((False, 1), (True, 1), (True, 2)):
        opts = pkg.Rho2sdfOptions(threshold_density=0.5, rbf_interp=interp, rbf_grid="same" if smooth == 1 else "fine",
                                  artifact_min_component_ratio=0.05)
        ia, ib = {}, {}
        a = pkg.rho2sdf("t", X, IEN, rho, options=opts, sdf_grid=pg, info=ia)
        b = pkg.rho2sdf("t", X, IEN, rho, options=opts, sdf_grid=pg, n_gpus=G, info=ib)
        assert ia["n_flipped"] == ib["n_flipped"] > 0 and ia["cg_iters"] == ib["cg_iters"] and ia["level_shift"] == ib["level_shift"]
        assert np.array_equal(a[3], b[3]), (G, interp, smooth, "sdf_dists")
        assert np.array_equal(a[0], b[0]), (G, interp, smooth, "fine_sdf")
    pkg._lib.lib().r2s_release_cache()


def test_fan_out_after_another_mesh_on_other_shares(pkg, monkeypatch):
    """the plans of the logical devices outlive a call: a TET4 call on 8 devices after a TET4 call on 5 (another mesh, other
    shares of the grid) must not see anything of the earlier one - elements outside a device's share keep a CLEARED record
    that names no sign candidate (tools/fuzz_fan_out.py found 145 wrong signs here before elem_prep_kernel cleared it)"""
    from rho2sdf_jl_amd import synthetic
    monkeypatch.setenv("R2S_MULTI_OVERSUBSCRIBE", "1")
    Xa, Ia, ra = synthetic.tet_mesh(7, jitter=0.25, seed=59)
    ga = pkg.Grid(Xa.min(0), Xa.max(0), synthetic.grid_n_max_for_points(67), 3)
    Xb, Ib, rb = synthetic.tet_mesh(6, jitter=0.12, seed=63)
    gb = pkg.Grid(Xb.min(0), Xb.max(0), synthetic.grid_n_max_for_points(66), 3)
    rng = np.random.default_rng(63)
    rb = np.clip(rb + rng.normal(0, 0.2, len(rb)), 0, 1)
    want = pkg.sdf_fused(pkg.Mesh(Xb, Ib), gb, rb, 0.5)
    for Ga, Gb in ((5, 8), (3, 8), (8, 5)):
        pkg.sdf_fused(pkg.Mesh(Xa, Ia), ga, ra, 0.5, n_gpus=Ga)
        got = pkg.sdf_fused(pkg.Mesh(Xb, Ib), gb, rb, 0.5, n_gpus=Gb)
        assert np.array_equal(got, want), (Ga, Gb, int((got != want).sum()))
    pkg._lib.lib().r2s_release_cache()

