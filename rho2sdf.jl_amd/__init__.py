"""rho2sdf.jl_amd - MI355X-native signed-distance extraction (hot path of Rho2sdf.jl).

The directory name contains a dot, so load it with tests/conftest.py's
`load_package()` (importlib) or `__graft_entry__.load_package()`; it registers
itself as module `rho2sdf_jl_amd`.
"""
from . import _lib
from .api import (DevicePlan, Grid, Mesh, Sign_Detection, evalDistances, getMesh_AABB,
                  noninteractive_sdf_grid_setup, sdf_fused)

__all__ = ["DevicePlan", "Grid", "Mesh", "Sign_Detection", "evalDistances", "getMesh_AABB",
           "noninteractive_sdf_grid_setup", "sdf_fused", "_lib"]
