"""Loader for the C-ABI shared library (include/rho2sdf_hip.h).

The library is the product; there is no CPU fallback.  If the in-tree
librho2sdf_hip.so is missing or does not load, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R2S_LIB_OVERRIDE") or os.path.join(_HERE, "librho2sdf_hip.so")  # override: A/B builds

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_float_p = ctypes.POINTER(ctypes.c_float)


class R2SGrid(ctypes.Structure):
    """mirrors `mutable struct Grid` (reference src/MeshGrid/Grid.jl:2-7)"""
    _fields_ = [("aabb_min", ctypes.c_double * 3), ("aabb_max", ctypes.c_double * 3),
                ("N", ctypes.c_int64 * 3), ("cell_size", ctypes.c_double), ("ngp", ctypes.c_int64)]


class R2SParams(ctypes.Structure):
    _fields_ = [("band_factor", ctypes.c_double), ("elem_type", ctypes.c_int32),
                ("device", ctypes.c_int32), ("zstride", ctypes.c_int32), ("zphase", ctypes.c_int32),
                ("n_gpus", ctypes.c_int32), ("true_min", ctypes.c_int32), ("sign_no_inner", ctypes.c_int32),
                ("reserved_", ctypes.c_int32)]


class R2SOptions(ctypes.Structure):
    """mirrors r2s_options (= Rho2sdfOptions, reference src/RhoToSDF.jl:9-77)"""
    _fields_ = [("threshold_density", ctypes.c_double), ("band_factor", ctypes.c_double),
                ("artifact_min_component_ratio", ctypes.c_double), ("rbf_kernel_threshold", ctypes.c_double),
                ("elem_type", ctypes.c_int32), ("rbf_interp", ctypes.c_int32), ("rbf_smooth", ctypes.c_int32),
                ("remove_artifacts", ctypes.c_int32), ("device", ctypes.c_int32), ("n_gpus", ctypes.c_int32),
                ("skip_rbf", ctypes.c_int32), ("true_min", ctypes.c_int32), ("sign_no_inner", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 3)]


class R2SRunInfo(ctypes.Structure):
    _fields_ = [("V_domain", ctypes.c_double), ("V_frac", ctypes.c_double), ("rho_t", ctypes.c_double),
                ("n_flipped", ctypes.c_int64), ("level_shift", ctypes.c_float), ("cg_iters", ctypes.c_int32),
                ("threshold_iters", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("ms_upload", ctypes.c_double), ("ms_pre", ctypes.c_double), ("ms_sdf", ctypes.c_double),
                ("ms_sdf_kernels", ctypes.c_double), ("ms_artifacts", ctypes.c_double), ("ms_rbf", ctypes.c_double),
                ("ms_download", ctypes.c_double), ("ms_total", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class R2SStats(ctypes.Structure):
    _fields_ = [("n_solid", ctypes.c_int64), ("n_iso", ctypes.c_int64), ("n_items", ctypes.c_int64),
                ("n_band_entries", ctypes.c_int64), ("n_sign_entries", ctypes.c_int64),
                ("n_tiles", ctypes.c_int64), ("n_active_tiles", ctypes.c_int64),
                ("n_active_sign_tiles", ctypes.c_int64), ("n_iso_chunks", ctypes.c_int64),
                ("n_any_tiles", ctypes.c_int64),
                ("ms_prep", ctypes.c_double), ("ms_bins", ctypes.c_double),
                ("ms_fill", ctypes.c_double), ("ms_main", ctypes.c_double),
                ("ms_gather", ctypes.c_double), ("ms_sign", ctypes.c_double),
                ("n_sign_only_tiles", ctypes.c_int64), ("ms_iso_fast", ctypes.c_double),
                ("n_iso_straggler", ctypes.c_int64), ("n_iso_fail", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class R2SVtuMesh(ctypes.Structure):
    _fields_ = [("nnp", ctypes.c_int64), ("nel", ctypes.c_int64), ("nen", ctypes.c_int32),
                ("elem_type", ctypes.c_int32), ("n_skipped", ctypes.c_int64), ("X", c_double_p),
                ("IEN", c_int64_p), ("rho", c_double_p), ("density_field", ctypes.c_char * 64)]


# every symbol include/rho2sdf_hip.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
_MESH = [c_double_p, ctypes.c_int64, c_int64_p, ctypes.c_int64]
_COMMON = _MESH + [c_double_p, ctypes.c_double, ctypes.POINTER(R2SGrid), ctypes.POINTER(R2SParams)]
SYMBOLS = [
    ("r2s_version", ctypes.c_int, []),
    ("r2s_last_error", ctypes.c_char_p, []),
    ("r2s_device_count", ctypes.c_int, []),
    ("r2s_default_params", None, [ctypes.POINTER(R2SParams)]),
    ("r2s_grid_make", ctypes.c_int, [c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.POINTER(R2SGrid)]),
    ("r2s_auto_grid", ctypes.c_int, _MESH + [ctypes.c_int32, ctypes.POINTER(R2SGrid), c_double_p]),
    ("r2s_eval_distances", ctypes.c_int, _COMMON + [c_double_p, c_double_p, ctypes.POINTER(R2SStats)]),
    ("r2s_sign_detection", ctypes.c_int, _COMMON + [c_double_p, ctypes.POINTER(R2SStats)]),
    ("r2s_sdf", ctypes.c_int, _COMMON + [c_double_p, ctypes.POINTER(R2SStats)]),
    ("r2s_plan_create", ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(_P)]),
    ("r2s_plan_destroy", None, [_P]),
    ("r2s_plan_run_dev", ctypes.c_int,
     [_P, _P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_double, ctypes.POINTER(R2SGrid),
      ctypes.POINTER(R2SParams), ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _P, _P, _P, _P, _P,
      ctypes.POINTER(R2SStats)]),
    ("r2s_plan_pack_tiles_dev", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int64, c_int64_p, _P]),
    ("r2s_unpack_tiles_dev", ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.POINTER(R2SGrid), _P, _P]),
    ("r2s_fill_dev", ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_double, _P]),
    ("r2s_plan_pack_tiles2_dev", ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int64, _P, _P, ctypes.c_int64, c_int64_p,
                                                c_int64_p, _P]),
    ("r2s_unpack_masks_dev", ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.POINTER(R2SGrid), ctypes.c_double, _P, _P]),
    ("r2s_unpack_segments_dev", ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.POINTER(R2SGrid), ctypes.c_double, _P, _P]),
    ("r2s_mesh_volume", ctypes.c_int, _MESH + [ctypes.c_int32, c_double_p, ctypes.c_int32, c_double_p, c_double_p]),
    ("r2s_dense_in_nodes", ctypes.c_int, _MESH + [ctypes.c_int32, c_double_p, ctypes.c_int32, c_double_p]),
    ("r2s_find_threshold", ctypes.c_int, _MESH + [c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                                  ctypes.c_int32, c_double_p, ctypes.POINTER(ctypes.c_int32)]),
    ("r2s_find_threshold_et", ctypes.c_int, _MESH + [ctypes.c_int32, c_double_p, ctypes.c_double, ctypes.c_double,
                                                     ctypes.c_int32, ctypes.c_int32, c_double_p,
                                                     ctypes.POINTER(ctypes.c_int32)]),
    ("r2s_isocontour_volume", ctypes.c_int, _MESH + [ctypes.c_int32, c_double_p, ctypes.c_double, ctypes.c_int32,
                                                     c_double_p]),
    ("r2s_default_options", None, [ctypes.POINTER(R2SOptions)]),
    ("r2s_rho2sdf", ctypes.c_int, _MESH + [c_double_p, ctypes.POINTER(R2SOptions), ctypes.POINTER(R2SGrid), c_double_p,
                                           c_double_p, c_double_p, c_float_p, ctypes.POINTER(R2SRunInfo)]),
    ("r2s_last_host_phases", None, [c_double_p]),
    ("r2s_host_alloc", ctypes.c_void_p, [ctypes.c_size_t]),
    ("r2s_host_free", None, [ctypes.c_void_p]),
    ("r2s_remove_artifacts", ctypes.c_int, [c_double_p, ctypes.POINTER(R2SGrid), ctypes.c_double, ctypes.c_double,
                                            ctypes.c_int32, c_int64_p]),
    ("r2s_remove_artifacts_dev", ctypes.c_int, [_P, ctypes.POINTER(R2SGrid), ctypes.c_double, ctypes.c_double, _P,
                                                c_int64_p]),
    ("r2s_volume_from_sdf", ctypes.c_int, [c_float_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int32, ctypes.c_int32, c_float_p]),
    ("r2s_rbf_smooth", ctypes.c_int, [c_double_p, ctypes.POINTER(R2SGrid), ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_int32, c_float_p, c_float_p,
                                      ctypes.POINTER(ctypes.c_int32), c_float_p]),
    ("r2s_release_cache", None, []),
    ("r2s_export_vtu", ctypes.c_int, [ctypes.c_char_p] + _MESH + [ctypes.c_int32, ctypes.c_int32, c_double_p]),
    ("r2s_import_vtu", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(R2SVtuMesh)]),
    ("r2s_free_vtu_mesh", None, [ctypes.POINTER(R2SVtuMesh)]),
    ("r2s_export_vti", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(R2SGrid), _P, ctypes.c_int32, ctypes.c_int64,
                                      ctypes.c_char_p, ctypes.c_int32]),
    ("r2s_export_vti_z", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(R2SGrid), _P, ctypes.c_int32, ctypes.c_int64,
                                        ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]),
    ("r2s_import_mat", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(R2SVtuMesh)]),
    ("r2s_rbf_smooth_dev", ctypes.c_int, [_P, ctypes.POINTER(R2SGrid), ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                          ctypes.c_double, _P, c_float_p, ctypes.POINTER(ctypes.c_int32), _P]),
]

OUT_DIST, OUT_SIGN, OUT_SDF, OUT_XP = 1, 2, 4, 8
HEX8, TET4 = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same SONAME as
        # /opt/rocm's).  If torch is going to be used for device memory / streams it has to be
        # loaded first so this library binds to the runtime torch already initialised;
        # otherwise two runtimes fight over the device ("No HIP GPUs are available").
        if os.environ.get("R2S_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)          # AttributeError if the symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class R2SError(RuntimeError):
    pass


def check(rc):
    """C-ABI status -> exception, like the reference's `error(...)` calls."""
    if rc != 0:
        raise R2SError(f"rho2sdf_hip error {rc}: {lib().r2s_last_error().decode()}")
