"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the
header declares, its host logic (Grid, automatic grid) matches the oracle, and compute
entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_fixture


def test_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "rho2sdf_hip.h")).read()
    declared = set(re.findall(r"\b(r2s_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"r2s_plan"}
    bound = {name for name, _, _ in pkg._lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name)
    assert pkg._lib.lib().r2s_version() == 100


def test_grid_matches_oracle(pkg, oracle):
    for name, nmax in (("sphere", 10), ("sphere", 25), ("chapadlo", 249), ("beam_vfrac_03", 59)):
        X, IEN, _ = load_fixture(name)
        g = pkg.Grid(X.min(0), X.max(0), nmax, 3)
        o = oracle.grid_make(X.min(0), X.max(0), nmax, 3)
        assert list(g.N) == list(o.N) and g.ngp == o.ngp
        assert g.cell_size == o.cell
        assert np.array_equal(g.AABB_min, np.array(o.amin[:])) and np.array_equal(g.AABB_max, np.array(o.amax[:]))
    X, IEN, _ = load_fixture("chapadlo")
    assert list(pkg.Grid(X.min(0), X.max(0), 249, 3).N + 1) == [87, 166, 257]


def test_auto_grid_matches_oracle(pkg, oracle):
    for name in ("beam_vfrac_03", "chapadlo", "sphere"):
        X, IEN, _ = load_fixture(name)
        g = pkg.noninteractive_sdf_grid_setup(pkg.Mesh(X, IEN))
        o, _ = oracle.auto_grid(X, IEN)
        assert list(g.N) == list(o.N) and g.cell_size == o.cell and g.ngp == o.ngp


def test_argument_errors(pkg):
    with pytest.raises(pkg._lib.R2SError):
        pkg.Mesh(np.zeros((4, 3)), np.ones((1, 5), np.int64))
    with pytest.raises(pkg._lib.R2SError):
        pkg.Grid([0, 0, 0], [1, 1, 1], 0)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback(pkg):
    """without a GPU the compute entry points must fail loudly"""
    X, IEN, rho = load_fixture("sphere")
    mesh = pkg.Mesh(X, IEN)
    grid = pkg.Grid(X.min(0), X.max(0), 5, 3)
    with pytest.raises(pkg._lib.R2SError, match="no HIP device|CPU fallback"):
        pkg.sdf_fused(mesh, grid, np.zeros(mesh.nnp), 0.5)
