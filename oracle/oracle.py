"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libr2s_oracle.so")


class OrcGrid(ctypes.Structure):
    _fields_ = [("amin", ctypes.c_double * 3), ("amax", ctypes.c_double * 3),
                ("N", ctypes.c_int64 * 3), ("cell", ctypes.c_double), ("ngp", ctypes.c_int64)]

    @property
    def dims(self):
        return tuple(int(n) + 1 for n in self.N)


class OrcStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in
                ("n_solid", "n_iso", "n_iso_solves", "n_iso_fail", "n_tri_tests", "n_invmap")]


def build(force=False):
    tight_lib = os.path.join(_HERE, "libr2s_oracle_tight.so")
    src_t = os.path.getmtime(os.path.join(_HERE, "r2s_oracle.c"))
    if force or not os.path.exists(_LIB) or not os.path.exists(tight_lib) or \
            os.path.getmtime(_LIB) < src_t or os.path.getmtime(tight_lib) < src_t:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None
_lib_tight = None
_use_tight = False
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib, _lib_tight
    if _use_tight:
        if _lib_tight is None:
            build()
            _lib_tight = ctypes.CDLL(os.path.join(_HERE, "libr2s_oracle_tight.so"))
        return _lib_tight
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


class tight:
    """context manager: route every call to the frozen tight-tolerance build (-DORC_TIGHT)"""

    def __enter__(self):
        global _use_tight
        self.prev, _use_tight = _use_tight, True

    def __exit__(self, *a):
        global _use_tight
        _use_tight = self.prev


def inv_map_hex8(x, Xe):
    """find_local_coordinates restatement for one (element, point) -> (ok, xi)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    Xe = np.ascontiguousarray(Xe, dtype=np.float64)
    xi = np.zeros(3)
    ok = lib().orc_inv_map_hex8(_d(x), _d(Xe), _d(xi))
    return bool(ok), xi


def iso_project_tet4(x, Xe4, re4, rt):
    """compute_coords_on_iso (TET4) restatement -> lambda[3]"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    Xe4 = np.ascontiguousarray(Xe4, dtype=np.float64)
    re4 = np.ascontiguousarray(re4, dtype=np.float64)
    lam = np.zeros(3)
    lib().orc_iso_project_tet4(_d(x), _d(Xe4), _d(re4), ctypes.c_double(rt), _d(lam))
    return lam


class iso_pair_log:
    """records every (element, voxel, xi) iso pair of the HEX8 eval_distances calls made inside the block"""

    def __init__(self, cap):
        self.el = np.zeros(cap, np.int64)
        self.v = np.zeros(cap, np.int64)
        self.xi = np.zeros((cap, 3))
        self.n = 0

    def __enter__(self):
        lib().orc_iso_log(ctypes.c_int64(len(self.el)), _i(self.el), _i(self.v), _d(self.xi))
        return self

    def __exit__(self, *a):
        f = lib().orc_iso_log_count
        f.restype = ctypes.c_int64
        self.n = int(f())
        lib().orc_iso_log(ctypes.c_int64(0), None, None, None)


class iso_override:
    """replaces the iso-surface solver of HEX8 eval_distances by a table of local coordinates (pair-log order)"""

    def __init__(self, xi):
        self.xi = np.ascontiguousarray(xi, dtype=np.float64)

    def __enter__(self):
        lib().orc_iso_override(_d(self.xi), ctypes.c_int64(len(self.xi)))

    def __exit__(self, *a):
        lib().orc_iso_override(None, ctypes.c_int64(0))


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _mesh(X, IEN):
    X = np.ascontiguousarray(X, dtype=np.float64)
    IEN = np.ascontiguousarray(IEN, dtype=np.int64)
    assert X.ndim == 2 and X.shape[1] == 3 and IEN.ndim == 2
    et = {8: 0, 4: 1}[IEN.shape[1]]
    return X, IEN, et


def grid_make(xmin, xmax, n_max, margin=3):
    g = OrcGrid()
    a = (ctypes.c_double * 3)(*xmin)
    b = (ctypes.c_double * 3)(*xmax)
    lib().orc_grid_make(a, b, ctypes.c_int64(n_max), ctypes.c_int64(margin), ctypes.byref(g))
    return g


def auto_grid(X, IEN):
    X, IEN, et = _mesh(X, IEN)
    g = OrcGrid()
    med = ctypes.c_double()
    rc = lib().orc_auto_grid(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)), et,
                             ctypes.byref(g), ctypes.byref(med))
    assert rc == 0
    return g, med.value


def grid_points(g):
    pts = np.empty((g.ngp, 3))
    lib().orc_grid_points(ctypes.byref(g), _d(pts))
    return pts


def dense_in_nodes(X, IEN, rho_e):
    X, IEN, et = _mesh(X, IEN)
    rho_e = np.ascontiguousarray(rho_e, dtype=np.float64)
    out = np.empty(len(X))
    rc = lib().orc_dense_in_nodes(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)),
                                  et, _d(rho_e), _d(out))
    assert rc == 0
    return out


def eval_distances(X, IEN, rho_n, rho_t, g, band_factor=1.1, want_xp=True):
    X, IEN, et = _mesh(X, IEN)
    rho_n = np.ascontiguousarray(rho_n, dtype=np.float64)
    dist = np.empty(g.ngp)
    xp = np.empty((g.ngp, 3)) if want_xp else None
    st = OrcStats()
    rc = lib().orc_eval_distances(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)), et,
                                  _d(rho_n), ctypes.c_double(rho_t), ctypes.byref(g),
                                  ctypes.c_double(band_factor), _d(dist),
                                  _d(xp) if want_xp else None, ctypes.byref(st))
    assert rc == 0, rc
    return dist, xp, {k: getattr(st, k) for k, _ in OrcStats._fields_}


def sign_detection(X, IEN, rho_n, rho_t, g, bruteforce=False):
    X, IEN, et = _mesh(X, IEN)
    rho_n = np.ascontiguousarray(rho_n, dtype=np.float64)
    s = np.empty(g.ngp)
    if bruteforce:
        rc = lib().orc_sign_detection_bruteforce(_d(X), ctypes.c_int64(len(X)), _i(IEN),
                                                 ctypes.c_int64(len(IEN)), _d(rho_n),
                                                 ctypes.c_double(rho_t), ctypes.byref(g), _d(s))
    else:
        rc = lib().orc_sign_detection(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)),
                                      et, _d(rho_n), ctypes.c_double(rho_t), ctypes.byref(g), _d(s))
    assert rc == 0, rc
    return s


def iso_project_hex8(x, Xe, rho_e, rho_t):
    x = np.ascontiguousarray(x, dtype=np.float64)
    Xe = np.ascontiguousarray(Xe, dtype=np.float64)
    rho_e = np.ascontiguousarray(rho_e, dtype=np.float64)
    xi = np.zeros(3)
    f = lib().orc_iso_project_hex8
    f.restype = ctypes.c_int
    it = f(_d(x), _d(Xe), _d(rho_e), ctypes.c_double(rho_t), _d(xi))
    return xi, it


class true_min:
    """context manager: order-independent "true minimum" semantics for the calls inside the block"""

    def __enter__(self):
        lib().orc_set_true_min(ctypes.c_int(1))

    def __exit__(self, *a):
        lib().orc_set_true_min(ctypes.c_int(0))


def set_k_sampling(stride=1, phase=0):
    """bench.py cpu_baseline: evaluate only planes k % stride == phase"""
    lib().orc_set_k_sampling(ctypes.c_int64(stride), ctypes.c_int64(phase))


_fp = ctypes.POINTER(ctypes.c_float)


def gauss_legendre(n):
    x = np.zeros(n)
    w = np.zeros(n)
    lib().orc_gauss_legendre(ctypes.c_int(n), _d(x), _d(w))
    return x, w


def mesh_volume(X, IEN, rho_e):
    """calculate_mesh_volume -> (V_domain, V_frac)"""
    X, IEN, et = _mesh(X, IEN)
    assert et == 0
    rho_e = np.ascontiguousarray(rho_e, dtype=np.float64)
    vd, vf = ctypes.c_double(), ctypes.c_double()
    lib().orc_mesh_volume(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)), _d(rho_e),
                          ctypes.byref(vd), ctypes.byref(vf))
    return vd.value, vf.value


def isocontour_volume(X, IEN, rho_n, thr):
    X, IEN, et = _mesh(X, IEN)
    rho_n = np.ascontiguousarray(rho_n, dtype=np.float64)
    f = lib().orc_isocontour_volume if et == 0 else lib().orc_isocontour_volume_tet4
    f.restype = ctypes.c_double
    return f(_d(X), _i(IEN), ctypes.c_int64(len(IEN)), _d(rho_n), ctypes.c_double(thr))


def find_threshold(X, IEN, rho_n, target_volume, tol=1e-4, maxit=60):
    X, IEN, et = _mesh(X, IEN)
    rho_n = np.ascontiguousarray(rho_n, dtype=np.float64)
    rt, it = ctypes.c_double(), ctypes.c_int()
    rc = (lib().orc_find_threshold if et == 0 else lib().orc_find_threshold_tet4)(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)), _d(rho_n),
                                  ctypes.c_double(target_volume), ctypes.c_double(tol), ctypes.c_int(maxit),
                                  ctypes.byref(rt), ctypes.byref(it))
    if rc:
        raise ValueError("Requested volume is outside the possible range")   # Isocontour_volume.jl:94
    return rt.value, it.value


def remove_artifacts(sdf, g, threshold=0.0, min_component_ratio=0.01):
    """in place; returns the number of flipped nodes"""
    assert sdf.dtype == np.float64 and sdf.flags.c_contiguous and sdf.size == g.ngp
    f = lib().orc_remove_artifacts
    f.restype = ctypes.c_int64
    return int(f(_d(sdf), ctypes.byref(g), ctypes.c_double(threshold), ctypes.c_double(min_component_ratio)))


def volume_from_sdf(sdf_f32, edge, iso=0.0, order=9):
    """sdf_f32: (nz, ny, nx) float32 (x fastest) -> Float32 volume"""
    a = np.ascontiguousarray(sdf_f32, dtype=np.float32)
    nz, ny, nx = a.shape
    f = lib().orc_volume_from_sdf
    f.restype = ctypes.c_float
    return float(f(a.ctypes.data_as(_fp), ctypes.c_int64(nx), ctypes.c_int64(ny), ctypes.c_int64(nz),
                   ctypes.c_float(edge), ctypes.c_float(iso), ctypes.c_int(order)))


def rbf_smoothing(sdf, g, is_interp, smooth, target_volume, kthr=1e-3):
    """RBFs_smoothing -> (fine_sdf (nz',ny',nx') float32, th, cg_iterations, LSF on the coarse grid)"""
    sdf = np.ascontiguousarray(sdf, dtype=np.float64)
    dims = tuple(int(n) * smooth + 1 for n in g.N)
    fine = np.empty(dims[2] * dims[1] * dims[0], dtype=np.float32)
    lsf = np.empty(g.ngp, dtype=np.float32)
    th, its = ctypes.c_float(), ctypes.c_int()
    rc = lib().orc_rbf_smoothing(_d(sdf), ctypes.byref(g), ctypes.c_int(int(is_interp)), ctypes.c_int(smooth),
                                 ctypes.c_double(kthr), ctypes.c_double(target_volume),
                                 fine.ctypes.data_as(_fp), ctypes.byref(th), ctypes.byref(its),
                                 lsf.ctypes.data_as(_fp))
    if rc:
        raise ValueError("every SDF value is a sentinel")
    return fine.reshape(dims[2], dims[1], dims[0]), th.value, its.value, lsf.reshape(g.dims[2], g.dims[1], g.dims[0])


def mesh_volume_tet4(X, IEN, rho_e):
    """calculate_mesh_volume for TET4 (MeshVolume.jl:75-117, including its 25 % low Jacobian)"""
    X, IEN, et = _mesh(X, IEN)
    assert et == 1
    rho_e = np.ascontiguousarray(rho_e, dtype=np.float64)
    vd, vf = ctypes.c_double(), ctypes.c_double()
    lib().orc_mesh_volume_tet4(_d(X), ctypes.c_int64(len(X)), _i(IEN), ctypes.c_int64(len(IEN)), _d(rho_e),
                               ctypes.byref(vd), ctypes.byref(vf))
    return vd.value, vf.value
