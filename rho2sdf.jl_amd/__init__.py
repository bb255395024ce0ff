"""rho2sdf.jl_amd - MI355X-native signed-distance extraction (hot path of Rho2sdf.jl).

The directory name contains a dot, so load it with tests/conftest.py's
`load_package()` (importlib) or `__graft_entry__.load_package()`; it registers
itself as module `rho2sdf_jl_amd`.
"""
from . import _lib
from .api import (DenseInNodes, DevicePlan, Grid, Mesh, RBFs_smoothing, Rho2sdfOptions, Sign_Detection,
                  calculate_mesh_volume, calculate_volume_from_sdf, evalDistances, find_threshold_for_volume,
                  exportSdfToVTI, exportToVTU, export_sdf_results, getMesh_AABB, import_vtu_mesh, noninteractive_sdf_grid_setup,
                  remove_sdf_artifacts, rho2sdf, sdf_fused, host_array, calculate_isocontour_volume, MeshInformations)

__all__ = ["DenseInNodes", "DevicePlan", "Grid", "Mesh", "RBFs_smoothing", "Rho2sdfOptions", "Sign_Detection",
           "calculate_mesh_volume", "calculate_volume_from_sdf", "evalDistances", "find_threshold_for_volume",
           "exportSdfToVTI", "exportToVTU", "export_sdf_results", "getMesh_AABB", "import_vtu_mesh", "noninteractive_sdf_grid_setup", "remove_sdf_artifacts",
           "rho2sdf", "sdf_fused", "host_array", "calculate_isocontour_volume", "MeshInformations", "_lib"]
