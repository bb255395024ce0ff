// Cross-file internals of the library (C++ linkage, not part of the C ABI): device-pointer forms of the
// pre-stage and post-processing stages, used by the host-pointer entry points of their own files and by the
// chained r2s_rho2sdf() in r2s_host.hip.  All of them run on the CURRENT device and are synchronous on return.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "r2s_common.hpp"

namespace r2s_int {

// calculate_mesh_volume (MeshVolume.jl:4-42): d_rho_e[nel] element densities
int mesh_volume_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_e,
                    double* V_domain, double* V_frac, DevBuf* ws = nullptr);

// DenseInNodes (NodalDensities.jl:89-218); the ascending node -> element lists are built on the device.
// ws: five buffers the caller keeps between calls (nullptr: temporaries of the call)
int dense_in_nodes_dev(const double* dX, int64_t nnp, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_e,
                       double* d_rho_n_out, DevBuf* ws = nullptr);

// find_threshold_for_volume (Isocontour_volume.jl:77-154); TET4: the same bisection over the TET4 iso-volume
// (the reference has none, SURVEY 8(f)2 - see r2s_pre.hip)
int find_threshold_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_n,
                       double target_volume, double tol, int maxit, double* rho_t_out, int* iters_out);

// calculate_isocontour_volume (Isocontour_volume.jl:1-75) at one threshold
int isocontour_volume_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_n,
                          double thr, double* volume_out);

// remove_sdf_artifacts! on a device-resident field (SdfArtifactRemoval.jl:134-245)
int remove_artifacts_dev(double* d_sdf, const r2s_grid* g, double threshold, double min_ratio, hipStream_t st,
                         int64_t* n_flipped);

// RBFs_smoothing with device-resident input / output (RBFs4Smoothing.jl:321-377)
// fine_chunk (optional): the output field is evaluated in a few Z chunks; after the launch of each one (default stream)
// fine_chunk(first, last) is called with the range [first, last) of d_fine_out that kernel fills, so that the caller
// can send finished chunks to the host while the next one is computed.  A non-zero return aborts.
// fine_early: the chunks are evaluated BEFORE the level bisection and forwarded WITHOUT the level shift (the evaluation needs
// the weights only; the caller adds *th_out to what it received - the same Float32 sum); d_fine_out then stays WITHOUT the
// shift (its chunks may still be travelling when the level is known).
int rbf_smooth_dev(const double* d_sdf, const r2s_grid* g, int is_interp, int smooth, double kthr, double target_volume,
                   float* d_fine_out, float* th_out, int* cg_iters,
                   const std::function<int(int64_t, int64_t)>* fine_chunk = nullptr, void* workspace = nullptr,
                   bool fine_early = false);
// workspace (optional): the call's device buffers, kept between calls on the CURRENT device (create / release there)
void* rbf_workspace_create();
void rbf_workspace_release(void* workspace);

// ---- Z-slab distributed post-processing (single process, one entry per device; SURVEY 8(e) second half) ----------
// A slab OWNS the grid planes [k0, k1) and HOLDS [h0, h1) (its planes plus the halo the stencils reach into);
// d_sdf is the Float64 field of the held planes on `device`, plane k at (k - h0) * plane.
struct Slab {
    int device;
    hipStream_t stream;
    int k0, k1, h0, h1;
    double* d_sdf;
};
// copies every held-but-not-owned plane from the slab that owns it (peer copies over xGMI); base[q] = the array of
// slab q (held planes), elem = bytes per value
int exchange_halo_slabs(const std::vector<Slab>& S, const std::vector<void*>& base, size_t elem, int64_t plane, int radius);
// remove_sdf_artifacts! with the components labelled per slab and merged across the slab interfaces on the host
// (boundary-plane labels only); owned planes of d_sdf are modified, halos are NOT refreshed
int remove_artifacts_slabs(const std::vector<Slab>& S, const r2s_grid* g, double threshold, double min_ratio, int64_t* n_flipped);
// RBFs_smoothing on slabs: halo exchanges of the CG direction / weights / LSF, plane-wise dot products summed in k
// order on the host, volume row sums reduced on slab 0 - bit-identical to rbf_smooth_dev on one device.
// fine_out_host: the caller's (host) array of the whole fine grid, filled slab by slab.
int rbf_smooth_slabs(const std::vector<Slab>& S, const r2s_grid* g, int is_interp, int smooth, double kthr, double target_volume,
                     float* fine_out_host, float* th_out, int* cg_iters);

// frees the cached per-device host sessions (r2s_host.hip); called by r2s_release_cache()
void release_host_sessions();

}  // namespace r2s_int
