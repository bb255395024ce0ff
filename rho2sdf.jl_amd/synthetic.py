"""Synthetic workloads named in BASELINE.md section 4 (configs 5 and NS).

HEX8: n^3 cells on [-1,1]^3, interior nodes jittered by U(-0.15h, 0.15h) (seed 20240501),
nodal density rho_n = clamp(0.5 + (0.7 - |x|)/(4h), 0, 1): the rho_t = 0.5 iso-surface is
(close to) a sphere of radius 0.7.  Node numbering x-fastest; element node order is the
reference's HEX8 order (bottom face CCW, then top; hex8_shape.jl:28-35).
"""
import numpy as np


def hex_mesh(n, jitter=0.15, seed=20240501):
    h = 2.0 / n
    ax = -1.0 + h * np.arange(n + 1)
    Z, Y, Xc = np.meshgrid(ax, ax, ax, indexing="ij")           # x fastest when flattened
    X = np.stack([Xc.ravel(), Y.ravel(), Z.ravel()], axis=1)
    rng = np.random.default_rng(seed)
    jit = rng.uniform(-jitter * h, jitter * h, X.shape)
    idx = np.arange(n + 1)
    interior = np.ones((n + 1,) * 3, dtype=bool)
    interior[[0, -1], :, :] = False
    interior[:, [0, -1], :] = False
    interior[:, :, [0, -1]] = False
    X = X + jit * interior.ravel()[:, None]
    m = n + 1
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = (k * m * m + j * m + i).ravel()
    off = np.array([0, 1, m + 1, m, m * m, m * m + 1, m * m + m + 1, m * m + m])
    IEN = (base[:, None] + off[None, :] + 1).astype(np.int64)   # 1-based
    r = np.linalg.norm(X, axis=1)
    rho_n = np.clip(0.5 + (0.7 - r) / (4 * h), 0.0, 1.0)
    del idx
    return np.ascontiguousarray(X), IEN, rho_n


def grid_n_max_for_points(npts):
    """N_max such that Grid(..., N_max, 3) on [-1,1]^3 has `npts` points per axis
    (N = N_max + 6 cells, +1 points; BASELINE.md: 505 -> 512)."""
    return npts - 7


# Schlafli split of a HEX8 into 6 TET4 sharing the 1-7 diagonal
# (reference test/PrimitiveGeometriesTest/SimpleCubeWithSchlafli.jl:22-29), 0-based local nodes
SCHLAFLI = np.array([[0, 1, 2, 6], [0, 5, 1, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 4, 5, 6], [0, 7, 4, 6]])


def hex_to_tets(IEN_hex):
    """(nel, 8) 1-based HEX8 connectivity -> (6 nel, 4) TET4 connectivity, element-major"""
    return np.ascontiguousarray(IEN_hex[:, SCHLAFLI].reshape(-1, 4))


def tet_mesh(n, jitter=0.15, seed=20240501):
    """BASELINE.md config 5 family: the jittered hex mesh split 6-way"""
    X, IEN, rho_n = hex_mesh(n, jitter, seed)
    return X, hex_to_tets(IEN), rho_n


def radial_cube(n=10, side=10.0):
    """SimpleCube.jl / SimpleCubeWithSchlafli.jl: cube of side `side` centred at 0, n^3 cells, nodal density
    1 - r/(sqrt(3) side/2): the 0.5 iso-surface of the interpolant is close to a sphere of radius
    sqrt(3) side/4.  Returns (X, IEN_hex, rho_n)."""
    X, IEN, _ = hex_mesh(n, jitter=0.0)
    X = X * (side / 2.0)
    rho_n = 1.0 - np.linalg.norm(X, axis=1) / (np.sqrt(3.0) * side / 2.0)
    return X, IEN, rho_n
