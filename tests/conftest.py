import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def pkg():
    """the product package (C-ABI library loaded; raises if the .so is missing)"""
    return graft.build()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle - test infrastructure only"""
    graft.build()
    return graft.load_oracle()


def load_fixture(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return d["X"], d["IEN"].astype(np.int64), d["rho"]


def block_mesh(N):
    """TestGeometryBlock (reference src/PrimitiveGeometries/PrimitiveGeometries.jl:157-214),
    restated: nodes numbered z-fastest, HEX8 corner order (i,j,k),(i+1,j,k),(i+1,j+1,k),(i,j+1,k),+k."""
    N = np.array(N)
    delta = 2.0 / N.max()
    Lx = delta * N
    X = np.zeros((int(np.prod(N + 1)), 3))
    idm = {}
    for i in range(N[0] + 1):
        for j in range(N[1] + 1):
            for k in range(N[2] + 1):
                nid = i * (N[2] + 1) * (N[1] + 1) + j * (N[2] + 1) + k
                X[nid] = [-Lx[0] / 2 + i * delta, -Lx[1] / 2 + j * delta, -Lx[2] / 2 + k * delta]
                idm[(i, j, k)] = nid + 1
    IEN = np.zeros((int(np.prod(N)), 8), np.int64)
    for i in range(N[0]):
        for j in range(N[1]):
            for k in range(N[2]):
                e = i * N[2] * N[1] + j * N[2] + k
                c = [(i, j, k), (i + 1, j, k), (i + 1, j + 1, k), (i, j + 1, k),
                     (i, j, k + 1), (i + 1, j, k + 1), (i + 1, j + 1, k + 1), (i, j + 1, k + 1)]
                IEN[e] = [idm[t] for t in c]
    return X, IEN
