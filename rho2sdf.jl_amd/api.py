"""Host-side mirror of the reference's Julia interface for the hot path.

Names, argument meaning and error behaviour follow the reference functions
that `rho2sdf()` calls (src/RhoToSDF.jl:148-224); every function is a thin
marshalling layer over the C ABI in include/rho2sdf_hip.h - exactly what the
Julia `ccall` wrapper (julia/Rho2sdfHIP.jl) does.  No numerics happen here.

Array conventions (numpy): X is (nnp, 3) float64 = Julia's 3 x nnp column-major
matrix; IEN is (nel, nen) 1-based = Julia's nen x nel matrix; grid vectors are
flat in the reference's x-fastest order (Grid.jl:84-92).
"""
import ctypes

import numpy as np

from . import _lib as L


class Grid:
    """Grid(AABB_min, AABB_max, N_max, margineCells=3)  - src/MeshGrid/Grid.jl:2-35"""

    def __init__(self, AABB_min, AABB_max, N_max, margineCells=3, _raw=None):
        if _raw is not None:
            self.c = _raw
            return
        self.c = L.R2SGrid()
        a = (ctypes.c_double * 3)(*[float(v) for v in AABB_min])
        b = (ctypes.c_double * 3)(*[float(v) for v in AABB_max])
        L.check(L.lib().r2s_grid_make(a, b, int(N_max), int(margineCells), ctypes.byref(self.c)))

    AABB_min = property(lambda s: np.array(s.c.aabb_min[:]))
    AABB_max = property(lambda s: np.array(s.c.aabb_max[:]))
    N = property(lambda s: np.array(s.c.N[:], dtype=np.int64))
    cell_size = property(lambda s: float(s.c.cell_size))
    ngp = property(lambda s: int(s.c.ngp))
    dims = property(lambda s: tuple(int(n) + 1 for n in s.c.N))


class Mesh:
    """Flat view of `Mesh{T}` (src/MeshGrid/MeshInformations.jl:16-67): X, IEN (1-based)."""

    def __init__(self, X, IEN, element_type=None):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.IEN = np.ascontiguousarray(IEN, dtype=np.int64)
        if self.X.ndim != 2 or self.X.shape[1] != 3:
            raise L.R2SError("X must be (nnp, 3)")
        nen = self.IEN.shape[1]
        if nen not in (8, 4):
            # MeshInformations.jl:58-60
            raise L.R2SError(f"Element connectivity size ({nen}) doesn't match element type nodes")
        self.element_type = {8: L.HEX8, 4: L.TET4}[nen] if element_type is None else element_type
        self.nnp, self.nel, self.nen = len(self.X), len(self.IEN), nen


def getMesh_AABB(X):
    """src/MeshGrid/Grid.jl:73-77"""
    X = np.asarray(X)
    return X.min(axis=0), X.max(axis=0)


def noninteractive_sdf_grid_setup(mesh):
    """src/MeshGrid/Grid_setup.jl:94-108 -> Grid"""
    g = L.R2SGrid()
    med = ctypes.c_double()
    L.check(L.lib().r2s_auto_grid(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.element_type,
                                  ctypes.byref(g), ctypes.byref(med)))
    return Grid(None, None, None, _raw=g)


def _d(a):
    return a.ctypes.data_as(L.c_double_p)


def _i(a):
    return a.ctypes.data_as(L.c_int64_p)


def _params(mesh, band_factor, device):
    p = L.R2SParams()
    L.lib().r2s_default_params(ctypes.byref(p))
    p.band_factor = float(band_factor)
    p.elem_type = mesh.element_type
    p.device = int(device)
    return p


def _rho(mesh, rho_n):
    r = np.ascontiguousarray(rho_n, dtype=np.float64)
    if r.shape != (mesh.nnp,):
        raise L.R2SError("length of nodal densities does not match number of nodes")
    return r


def evalDistances(mesh, grid, rho_n, rho_t, *, band_factor=1.1, want_xp=True, device=-1, stats=None):
    """evalDistances(mesh, grid, points, rho_n, rho_t) -> (dist, xp)
    src/SignedDistances/sdfOnDensityField.jl:139-486 (`points` is implied by `grid`)."""
    r = _rho(mesh, rho_n)
    dist = np.empty(grid.ngp)
    xp = np.empty((grid.ngp, 3)) if want_xp else None
    st = L.R2SStats()
    p = _params(mesh, band_factor, device)
    L.check(L.lib().r2s_eval_distances(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(r), float(rho_t),
                                       ctypes.byref(grid.c), ctypes.byref(p), _d(dist),
                                       _d(xp) if want_xp else None, ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return dist, xp


def Sign_Detection(mesh, grid, rho_n, rho_t, *, device=-1, stats=None):
    """Sign_Detection(mesh, grid, points, rho_n, rho_t) -> signs  (SignDetection.jl:275-283)"""
    r = _rho(mesh, rho_n)
    s = np.empty(grid.ngp)
    st = L.R2SStats()
    p = _params(mesh, 1.1, device)
    L.check(L.lib().r2s_sign_detection(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(r), float(rho_t),
                                       ctypes.byref(grid.c), ctypes.byref(p), _d(s), ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return s


def sdf_fused(mesh, grid, rho_n, rho_t, *, band_factor=1.1, device=-1, stats=None):
    """`dists .* signs` in one pass (RhoToSDF.jl:169-171)."""
    r = _rho(mesh, rho_n)
    out = np.empty(grid.ngp)
    st = L.R2SStats()
    p = _params(mesh, band_factor, device)
    L.check(L.lib().r2s_sdf(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(r), float(rho_t),
                            ctypes.byref(grid.c), ctypes.byref(p), _d(out), ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return out


class DevicePlan:
    """Device-resident path (inputs/outputs are torch CUDA tensors = HBM buffers).

    torch is used only as the owner of device memory and streams; the pointers
    go straight into r2s_plan_run_dev().
    """

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        L.check(L.lib().r2s_plan_create(int(device), ctypes.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            L.lib().r2s_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, dX, dIEN, d_rho_n, rho_t, grid, *, k_begin=0, k_end=None, band_factor=1.1,
            elem_type=L.HEX8, dist=None, sign=None, sdf=None, xp=None, stream=None):
        import torch
        k_end = int(grid.c.N[2]) + 1 if k_end is None else int(k_end)
        for t, dt in ((dX, torch.float64), (dIEN, torch.int64), (d_rho_n, torch.float64)):
            assert t.is_cuda and t.is_contiguous() and t.dtype == dt
        mode = 0
        ptr = []
        for t, bit in ((dist, L.OUT_DIST), (sign, L.OUT_SIGN), (sdf, L.OUT_SDF), (xp, L.OUT_XP)):
            if t is not None:
                assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
                mode |= bit
                ptr.append(ctypes.c_void_p(t.data_ptr()))
            else:
                ptr.append(None)
        p = L.R2SParams()
        L.lib().r2s_default_params(ctypes.byref(p))
        p.band_factor = float(band_factor)
        p.elem_type = int(elem_type)
        st = L.R2SStats()
        s = ctypes.c_void_p(stream.cuda_stream if stream is not None
                            else torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().r2s_plan_run_dev(self._h, ctypes.c_void_p(dX.data_ptr()), dX.shape[0],
                                         ctypes.c_void_p(dIEN.data_ptr()), dIEN.shape[0],
                                         ctypes.c_void_p(d_rho_n.data_ptr()), float(rho_t),
                                         ctypes.byref(grid.c), ctypes.byref(p), int(k_begin), k_end, mode,
                                         ptr[0], ptr[1], ptr[2], ptr[3], s, ctypes.byref(st)))
        return st.as_dict()
