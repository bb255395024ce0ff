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


def _params(mesh, band_factor, device, n_gpus=1, true_min=False, sign_no_inner=False):
    p = L.R2SParams()
    L.lib().r2s_default_params(ctypes.byref(p))
    p.band_factor = float(band_factor)
    p.elem_type = mesh.element_type
    p.device = int(device)
    p.n_gpus = int(n_gpus)
    p.true_min = int(bool(true_min))
    p.sign_no_inner = int(bool(sign_no_inner))
    return p


def host_array(n, dtype=np.float64):
    """numpy array over pinned host memory from r2s_host_alloc (what the Julia wrapper `unsafe_wrap`s its result
    arrays from): device -> host copies into it are plain DMA.  Freed with r2s_host_free when the array dies."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    ptr = L.lib().r2s_host_alloc(max(nbytes, 1))
    if not ptr:
        raise L.R2SError("r2s_host_alloc failed: " + L.lib().r2s_last_error().decode())
    buf = (ctypes.c_char * max(nbytes, 1)).from_address(ptr)
    a = np.frombuffer(buf, dtype=dtype, count=int(n))
    weakref.finalize(buf, L.lib().r2s_host_free, ctypes.c_void_p(ptr))
    return a


def _out(out, n, dtype=np.float64):
    if out is None:
        return np.empty(n, dtype=dtype)
    if out.dtype != dtype or not out.flags.c_contiguous or out.size != n:
        raise L.R2SError("`out` must be a contiguous %s array of %d values" % (np.dtype(dtype).name, n))
    return out


def _rho(mesh, rho_n):
    r = np.ascontiguousarray(rho_n, dtype=np.float64)
    if r.shape != (mesh.nnp,):
        raise L.R2SError("length of nodal densities does not match number of nodes")
    return r


def evalDistances(mesh, grid, rho_n, rho_t, *, band_factor=1.1, want_xp=True, device=-1, stats=None, n_gpus=1, out=None, true_min=False):
    """evalDistances(mesh, grid, points, rho_n, rho_t) -> (dist, xp)
    src/SignedDistances/sdfOnDensityField.jl:139-486 (`points` is implied by `grid`)."""
    r = _rho(mesh, rho_n)
    dist = _out(out, grid.ngp)
    xp = np.empty((grid.ngp, 3)) if want_xp else None
    st = L.R2SStats()
    p = _params(mesh, band_factor, device, n_gpus, true_min)
    L.check(L.lib().r2s_eval_distances(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(r), float(rho_t),
                                       ctypes.byref(grid.c), ctypes.byref(p), _d(dist),
                                       _d(xp) if want_xp else None, ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return dist, xp


def Sign_Detection(mesh, grid, rho_n, rho_t, *, device=-1, stats=None, n_gpus=1, out=None, true_min=False, sign_no_inner=False):
    """Sign_Detection(mesh, grid, points, rho_n, rho_t) -> signs  (SignDetection.jl:275-283)"""
    r = _rho(mesh, rho_n)
    s = _out(out, grid.ngp)
    st = L.R2SStats()
    p = _params(mesh, 1.1, device, n_gpus, true_min, sign_no_inner)
    L.check(L.lib().r2s_sign_detection(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(r), float(rho_t),
                                       ctypes.byref(grid.c), ctypes.byref(p), _d(s), ctypes.byref(st)))
    if stats is not None:
        stats.update(st.as_dict())
    return s


def sdf_fused(mesh, grid, rho_n, rho_t, *, band_factor=1.1, device=-1, stats=None, n_gpus=1, out=None, true_min=False, sign_no_inner=False):
    """`dists .* signs` in one pass (RhoToSDF.jl:169-171).  `out`: result array to fill (e.g. from host_array)."""
    r = _rho(mesh, rho_n)
    out = _out(out, grid.ngp)
    st = L.R2SStats()
    p = _params(mesh, band_factor, device, n_gpus, true_min, sign_no_inner)
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
            elem_type=None, dist=None, sign=None, sdf=None, xp=None, stream=None, zstride=1, zphase=0, true_min=False):
        """one pass over planes [k_begin, k_end) (zstride <= 1) or over the interleaved tile layers
        t % zstride == zphase of the whole grid (see r2s_params in include/rho2sdf_hip.h)"""
        import torch
        if elem_type is None:
            elem_type = L.HEX8 if dIEN.shape[1] == 8 else L.TET4
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
        p.zstride = int(zstride)
        p.zphase = int(zphase)
        p.true_min = int(bool(true_min))
        st = L.R2SStats()
        s = ctypes.c_void_p(stream.cuda_stream if stream is not None
                            else torch.cuda.current_stream().cuda_stream)
        L.check(L.lib().r2s_plan_run_dev(self._h, ctypes.c_void_p(dX.data_ptr()), dX.shape[0],
                                         ctypes.c_void_p(dIEN.data_ptr()), dIEN.shape[0],
                                         ctypes.c_void_p(d_rho_n.data_ptr()), float(rho_t),
                                         ctypes.byref(grid.c), ctypes.byref(p), int(k_begin), k_end, mode,
                                         ptr[0], ptr[1], ptr[2], ptr[3], s, ctypes.byref(st)))
        return st.as_dict()

    # ---- sparse stitching (see include/rho2sdf_hip.h) ----
    @staticmethod
    def _stream(stream):
        import torch
        return ctypes.c_void_p(stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream)

    def pack_tiles(self, local_sdf, payload, ids, stream=None):
        """pack the non-sentinel tiles of the last run's output into payload (n,64) f64 / ids (n,) i32"""
        n = ctypes.c_int64()
        L.check(L.lib().r2s_plan_pack_tiles_dev(self._h, ctypes.c_void_p(local_sdf.data_ptr()),
                                                ctypes.c_void_p(payload.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                                int(ids.numel()), ctypes.byref(n), self._stream(stream)))
        return int(n.value)

    @staticmethod
    def unpack_tiles(payload, ids, n, grid, volume, stream=None):
        L.check(L.lib().r2s_unpack_tiles_dev(ctypes.c_void_p(payload.data_ptr()), ctypes.c_void_p(ids.data_ptr()), int(n),
                                             ctypes.byref(grid.c), ctypes.c_void_p(volume.data_ptr()),
                                             DevicePlan._stream(stream)))

    def pack_tiles2(self, local_sdf, payload, ids, masks, mask_ids, stream=None):
        """compressed packing: band tiles -> payload (n,64) f64 / ids i32; sign-only tiles -> masks i64 / mask_ids i32.
        Returns (n_full, n_mask)."""
        nf, nm = ctypes.c_int64(), ctypes.c_int64()
        L.check(L.lib().r2s_plan_pack_tiles2_dev(
            self._h, ctypes.c_void_p(local_sdf.data_ptr()), ctypes.c_void_p(payload.data_ptr()),
            ctypes.c_void_p(ids.data_ptr()), int(ids.numel()), ctypes.c_void_p(masks.data_ptr()),
            ctypes.c_void_p(mask_ids.data_ptr()), int(mask_ids.numel()), ctypes.byref(nf), ctypes.byref(nm),
            self._stream(stream)))
        return int(nf.value), int(nm.value)

    @staticmethod
    def unpack_masks(masks, mask_ids, n, grid, volume, magnitude=1.0e10, stream=None):
        L.check(L.lib().r2s_unpack_masks_dev(ctypes.c_void_p(masks.data_ptr()), ctypes.c_void_p(mask_ids.data_ptr()), int(n),
                                             ctypes.byref(grid.c), float(magnitude), ctypes.c_void_p(volume.data_ptr()),
                                             DevicePlan._stream(stream)))

    @staticmethod
    def unpack_segments(buf, world, seglen, cap_full, cap_mask, grid, volume, magnitude=1.0e10, stream=None):
        """scatter every rank's segment of the exchange buffer in two launches; the tile counts are read from the segment
        headers on the device (see r2s_unpack_segments_dev)"""
        L.check(L.lib().r2s_unpack_segments_dev(ctypes.c_void_p(buf.data_ptr()), int(world), int(seglen), int(cap_full), int(cap_mask),
                                                ctypes.byref(grid.c), float(magnitude), ctypes.c_void_p(volume.data_ptr()),
                                                DevicePlan._stream(stream)))

    @staticmethod
    def fill(t, value, stream=None):
        L.check(L.lib().r2s_fill_dev(ctypes.c_void_p(t.data_ptr()), int(t.numel()), float(value), DevicePlan._stream(stream)))


# ---------------------------------------------------------------------------------------
# pre-stage and post-processing (same thin marshalling; reference citations at each function)
# ---------------------------------------------------------------------------------------
def _f(a):
    return a.ctypes.data_as(L.c_float_p)


def calculate_mesh_volume(mesh, rho, *, device=-1):
    """calculate_mesh_volume(X, IEN, rho, T) -> [V_domain, V_frac]   src/MeshGrid/MeshVolume.jl:4-42"""
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    vd, vf = ctypes.c_double(), ctypes.c_double()
    L.check(L.lib().r2s_mesh_volume(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.element_type, _d(rho),
                                    int(device), ctypes.byref(vd), ctypes.byref(vf)))
    return vd.value, vf.value


def DenseInNodes(mesh, rho, *, device=-1):
    """DenseInNodes(mesh, rho) -> rho_n   src/MeshGrid/NodalDensities.jl:89-108"""
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    if rho.shape != (mesh.nel,):
        raise L.R2SError("length of element densities does not match number of elements")
    out = np.empty(mesh.nnp)
    L.check(L.lib().r2s_dense_in_nodes(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.element_type, _d(rho),
                                       int(device), _d(out)))
    return out


def find_threshold_for_volume(mesh, rho_n, target_volume, tolerance=1e-4, max_iterations=60, *, device=-1):
    """find_threshold_for_volume(mesh, nodal_values, tol, maxit); target_volume = V_domain*V_frac
    src/MeshGrid/Isocontour_volume.jl:77-154"""
    r = _rho(mesh, rho_n)
    rt = ctypes.c_double()
    it = ctypes.c_int32()
    L.check(L.lib().r2s_find_threshold_et(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.element_type, _d(r),
                                          float(target_volume), float(tolerance), int(max_iterations), int(device),
                                          ctypes.byref(rt), ctypes.byref(it)))
    return rt.value


def calculate_isocontour_volume(mesh, rho_n, iso_threshold, *, device=-1):
    """calculate_isocontour_volume(mesh, nodal_values, iso_threshold)   src/MeshGrid/Isocontour_volume.jl:1-75
    (TET4: the iso-volume assembled from the reference's TET4 quadrature, see include/rho2sdf_hip.h)"""
    r = _rho(mesh, rho_n)
    v = ctypes.c_double()
    L.check(L.lib().r2s_isocontour_volume(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.element_type, _d(r),
                                          float(iso_threshold), int(device), ctypes.byref(v)))
    return v.value


def remove_sdf_artifacts(sdf, grid, *, threshold=0.0, min_component_ratio=0.01, device=-1):
    """remove_sdf_artifacts!(sdf, grid; threshold, min_component_ratio) -> nodes flipped (sdf modified in place)
    src/SignedDistances/SdfArtifactRemoval.jl:134-245"""
    if sdf.dtype != np.float64 or not sdf.flags.c_contiguous:
        raise L.R2SError("sdf must be a contiguous float64 array (it is modified in place)")
    if sdf.size != grid.ngp:   # SdfArtifactRemoval.jl:141-143
        raise L.R2SError(f"SDF values length ({sdf.size}) doesn't match grid points ({grid.ngp})")
    n = ctypes.c_int64()
    L.check(L.lib().r2s_remove_artifacts(_d(sdf), ctypes.byref(grid.c), float(threshold), float(min_component_ratio),
                                         int(device), ctypes.byref(n)))
    return int(n.value)


def calculate_volume_from_sdf(fine_sdf, edge, *, iso_threshold=0.0, detailed_quad_order=9, device=-1):
    """calculate_volume_from_sdf(fine_sdf, fine_grid; iso_threshold, detailed_quad_order) -> Float32
    src/SdfSmoothing/CalcVolumeFromSDF.jl:26-125; fine_sdf is (nz, ny, nx) float32, `edge` the grid spacing."""
    a = np.ascontiguousarray(fine_sdf, dtype=np.float32)
    nz, ny, nx = a.shape
    v = ctypes.c_float()
    L.check(L.lib().r2s_volume_from_sdf(_f(a), nx, ny, nz, float(edge), float(iso_threshold), int(detailed_quad_order),
                                        int(device), ctypes.byref(v)))
    return float(v.value)


def RBFs_smoothing(sdf, grid, Is_interpolation, smooth, target_volume, threshold=1e-3, *, device=-1, info=None):
    """RBFs_smoothing(mesh, dist, grid, Is_interpolation, smooth, taskName, threshold) -> fine_sdf
    src/SdfSmoothing/RBFs4Smoothing.jl:321-377 (mesh only contributes V_frac*V_domain = target_volume).
    Returns fine_sdf as (nz', ny', nx') float32; the fine grid is AABB_min + spacing*(i,j,k)."""
    sdf = np.ascontiguousarray(sdf, dtype=np.float64)
    dims = tuple(int(n) * int(smooth) + 1 for n in grid.c.N)
    fine = np.empty(dims[2] * dims[1] * dims[0], dtype=np.float32)
    lsf = np.empty(grid.ngp, dtype=np.float32)
    th = ctypes.c_float()
    its = ctypes.c_int32()
    L.check(L.lib().r2s_rbf_smooth(_d(sdf), ctypes.byref(grid.c), int(bool(Is_interpolation)), int(smooth),
                                   float(threshold), float(target_volume), int(device), _f(fine), ctypes.byref(th),
                                   ctypes.byref(its), _f(lsf)))
    if info is not None:
        info.update(th=float(th.value), cg_iterations=int(its.value), lsf=lsf.reshape(grid.dims[2], grid.dims[1], grid.dims[0]))
    return fine.reshape(dims[2], dims[1], dims[0])


def exportSdfToVTI(filename, grid, values, value_label, smooth=None, compress=0):
    """exportSdfToVTI(filename, grid, values, value_label, smooth) - VTK ImageData (.vti)
    src/DataExport/ExportToVTI.jl:22-67: dimensions N(*smooth)+1, origin AABB_min, spacing cell_size(/smooth).
    `values` is float32 or float64, x fastest (any shape).  Returns the path written."""
    a = np.ascontiguousarray(values)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    L.check(L.lib().r2s_export_vti_z(str(filename).encode(), ctypes.byref(grid.c), a.ctypes.data_as(ctypes.c_void_p),
                                     int(a.dtype == np.float32), int(a.size), str(value_label).encode(),
                                     0 if smooth is None else int(smooth), int(compress)))
    filename = str(filename)
    return filename if filename.endswith(".vti") else filename + ".vti"


def exportToVTU(fileName, X, IEN, VTK_CODE=None, rho=None):
    """exportToVTU(fileName, X, IEN, VTK_CODE, rho) - ASCII UnstructuredGrid of the mesh with optional nodal
    densities (src/DataExport/ExportToVTU.jl:2-99).  X (nnp, 3), IEN (nel, nen) 1-based; VTK_CODE defaults to
    12 (hexahedron) / 10 (tetra) by the number of element nodes, like the reference's callers."""
    mesh = Mesh(X, IEN)
    code = (12 if mesh.nen == 8 else 10) if VTK_CODE is None else int(VTK_CODE)
    r = None
    if rho is not None:
        r = np.ascontiguousarray(rho, dtype=np.float64)
        if r.shape != (mesh.nnp,):
            raise L.R2SError("length of nodal densities does not match number of nodes")
    L.check(L.lib().r2s_export_vtu(str(fileName).encode(), _d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, mesh.nen, code,
                                   _d(r) if r is not None else None))
    return str(fileName)


def import_vtu_mesh(vtu_file, info=None):
    """import_vtu_mesh(vtu_file) -> (X, IEN, rho)   (src/DataImport/VTUImport.jl:22-112)
    X (nnp, 3) Float64, IEN (nel, nen) Int64 1-based, rho (nel,) element densities.  ASCII .vtu only.
    `info` (a dict) receives element_type, n_skipped and the cell-data field the densities came from."""
    m = L.R2SVtuMesh()
    L.check(L.lib().r2s_import_vtu(str(vtu_file).encode(), ctypes.byref(m)))
    try:
        X = np.ctypeslib.as_array(m.X, shape=(m.nnp, 3)).copy()
        IEN = np.ctypeslib.as_array(m.IEN, shape=(m.nel, m.nen)).copy()
        rho = np.ctypeslib.as_array(m.rho, shape=(m.nel,)).copy()
        if info is not None:
            info.update(element_type=int(m.elem_type), n_skipped=int(m.n_skipped),
                        density_field=m.density_field.decode())
    finally:
        L.lib().r2s_free_vtu_mesh(ctypes.byref(m))
    return X, IEN, rho


def MeshInformations(mat_file):
    """MeshInformations(matread(file)) -> (X, IEN, rho)   (src/MeshGrid/MeshInformations.jl:3-12) for MATLAB level-5
    .mat files: X (nnp, 3), IEN (nel, nen) = stored connectivity + 1, rho (nel,)."""
    m = L.R2SVtuMesh()
    L.check(L.lib().r2s_import_mat(str(mat_file).encode(), ctypes.byref(m)))
    try:
        X = np.ctypeslib.as_array(m.X, shape=(m.nnp, 3)).copy()
        IEN = np.ctypeslib.as_array(m.IEN, shape=(m.nel, m.nen)).copy()
        rho = np.ctypeslib.as_array(m.rho, shape=(m.nel,)).copy()
    finally:
        L.lib().r2s_free_vtu_mesh(ctypes.byref(m))
    return X, IEN, rho


def export_sdf_results(fine_sdf, sdf_grid, taskName, smooth, is_interpolation, element_type):
    """export_sdf_results_with_element_type (src/RhoToSDF.jl:249-283), the .vti part: same file name
    `<task>_<HEX8|TET4>_B-<round(cell,4)>_smooth-<s>_<Interpolation|Approximation>.vti`, point array "distance".
    (The two .jld2 dumps are Julia serialisation and stay in the Julia package.)"""
    name = "Interpolation" if is_interpolation else "Approximation"
    ename = "HEX8" if element_type == L.HEX8 else "TET4"
    B = round(float(sdf_grid.cell_size), 4)
    return exportSdfToVTI(f"{taskName}_{ename}_B-{B}_smooth-{smooth}_{name}.vti", sdf_grid, fine_sdf, "distance", smooth)


class Rho2sdfOptions:
    """Rho2sdfOptions (src/RhoToSDF.jl:9-77): same fields, defaults and validation rules; file-export
    switches are accepted and ignored here (file I/O stays in the Julia package)."""

    def __init__(self, threshold_density=None, sdf_grid_setup="manual", export_input_data=False,
                 export_nodal_densities=False, export_raw_sdf=False, rbf_interp=True, rbf_grid="same",
                 remove_artifacts=True, artifact_min_component_ratio=0.01, export_analysis=False,
                 element_type=None):
        import warnings
        if threshold_density is not None and not (0.0 <= threshold_density <= 1.0):
            warnings.warn(f"Threshold density {threshold_density} is outside the valid range [0.0, 1.0]. "
                          "Will use automatic calculation instead.")
            threshold_density = None
        if sdf_grid_setup not in ("manual", "automatic"):
            warnings.warn(f"Invalid sdf_grid_setup: {sdf_grid_setup}. Using default manual instead.")
            sdf_grid_setup = "manual"
        if rbf_grid not in ("same", "fine"):
            warnings.warn(f"Invalid rbf_grid: {rbf_grid}. Using default same instead.")
            rbf_grid = "same"
        self.threshold_density = threshold_density
        self.sdf_grid_setup = sdf_grid_setup
        self.rbf_interp = rbf_interp
        self.rbf_grid = rbf_grid
        self.remove_artifacts = remove_artifacts
        self.artifact_min_component_ratio = artifact_min_component_ratio
        self.element_type = element_type


def rho2sdf(taskName, X, IEN, rho, *, options=None, sdf_grid=None, device=-1, export_results=False, n_gpus=1,
            info=None, pinned_results=False):
    """rho2sdf(taskName, X, IEN, rho; options) -> (fine_sdf, fine_grid, sdf_grid, sdf_dists)
    src/RhoToSDF.jl:116-242.  ONE call into the library (r2s_rho2sdf): the mesh goes up once, mesh volume ->
    nodal densities -> threshold -> raw SDF -> artifact removal -> RBF smoothing run on HBM-resident data, the two
    result arrays come down once (pinned_results=True: into arrays from r2s_host_alloc - plain DMA, but pinning 1.6 GB
    costs ~0.2 s, so it only pays when the arrays are reused; the default staged path runs within 5 % of it).
    `sdf_grid` replaces the interactive prompt of sdf_grid_setup = :manual (Grid_setup.jl:111-154 is out of scope).
    fine_grid is returned as (origin, spacing, dims) instead of one heap vector per voxel.  export_results=True
    writes the final `.vti` like RhoToSDF.jl:230-238 (the .jld2 dumps stay in the Julia package).  `info` (a dict)
    receives V_domain, V_frac, rho_t, n_flipped, level_shift, cg_iters, per-stage milliseconds and rho_n."""
    options = options or Rho2sdfOptions()
    mesh = Mesh(X, IEN, options.element_type)
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    if rho.shape != (mesh.nel,):
        raise L.R2SError("length of element densities does not match number of elements")
    if sdf_grid is None:
        if options.sdf_grid_setup != "automatic":
            raise L.R2SError("sdf_grid_setup = :manual needs an explicit sdf_grid here")
        sdf_grid = noninteractive_sdf_grid_setup(mesh)                                       # :141-145
    smooth = 1 if options.rbf_grid == "same" else 2                                          # :222
    o = L.R2SOptions()
    L.lib().r2s_default_options(ctypes.byref(o))
    if options.threshold_density is not None:                                                # :151-156
        o.threshold_density = float(options.threshold_density)
    o.elem_type = mesh.element_type
    o.rbf_interp = int(bool(options.rbf_interp))
    o.rbf_smooth = smooth
    o.remove_artifacts = int(bool(options.remove_artifacts))
    o.artifact_min_component_ratio = float(options.artifact_min_component_ratio)
    o.device = int(device)
    o.n_gpus = int(n_gpus)
    dims = tuple(int(nn) * smooth + 1 for nn in sdf_grid.c.N)
    nfine = dims[0] * dims[1] * dims[2]
    alloc = host_array if pinned_results else (lambda n, dt=np.float64: np.empty(n, dtype=dt))
    sdf_dists = alloc(sdf_grid.ngp)
    fine = alloc(nfine, np.float32)
    rho_n = np.empty(mesh.nnp)
    ri = L.R2SRunInfo()
    L.check(L.lib().r2s_rho2sdf(_d(mesh.X), mesh.nnp, _i(mesh.IEN), mesh.nel, _d(rho), ctypes.byref(o),
                                ctypes.byref(sdf_grid.c), _d(rho_n), None, _d(sdf_dists), _f(fine), ctypes.byref(ri)))
    if info is not None:
        info.update(ri.as_dict())
        info["rho_n"] = rho_n
    fine_sdf = fine.reshape(dims[2], dims[1], dims[0])
    xmin, xmax = np.float32(sdf_grid.AABB_min[0]), np.float32(sdf_grid.AABB_max[0])
    spacing = (xmax - xmin) / np.float32(fine_sdf.shape[2] - 1)
    fine_grid = (sdf_grid.AABB_min.astype(np.float32), float(spacing), fine_sdf.shape[::-1])
    if export_results:
        export_sdf_results(fine_sdf, sdf_grid, taskName, smooth, options.rbf_interp, mesh.element_type)
    return fine_sdf, fine_grid, sdf_grid, sdf_dists
