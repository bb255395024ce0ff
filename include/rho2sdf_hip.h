/*
 * rho2sdf_hip.h - C ABI of the MI355X-native signed-distance extraction engine.
 *
 * The reference (kopacja/rho2sdf.jl) has no FFI of its own for this path; the
 * drop-in boundary is the set of Julia functions `rho2sdf()` calls
 * (src/RhoToSDF.jl:148-224).  Each entry point below replaces the body of one of
 * them; the Julia-side `ccall` stubs are in INTEGRATION.md and
 * rho2sdf.jl_amd/julia/Rho2sdfHIP.jl.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams are void*).
 *   - array layouts are the reference's Julia column-major layouts:
 *       X    3 x nnp  Float64  -> double[nnp][3]
 *       IEN  nen x nel Int64, 1-based -> int64_t[nel][nen]
 *       grid linear index (0-based) = k*(N1+1)*(N2+1) + j*(N1+1) + i   (Grid.jl:84-92)
 *   - return 0 on success, negative on error; r2s_last_error() gives the text
 *     (the Julia wrapper turns it into `error(...)`, as the reference does).
 *   - `*_dev` entry points take DEVICE pointers (resident in HBM) and a HIP
 *     stream handle; the host-pointer entry points allocate/copy internally and
 *     never keep caller pointers after returning.
 *   - the library fails with R2S_ERR_NO_DEVICE when no gfx950 device is usable;
 *     there is no CPU fallback.
 */
#ifndef RHO2SDF_HIP_H
#define RHO2SDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R2S_VERSION 100 /* 0.1.0 */

#define R2S_OK 0
#define R2S_ERR_ARG (-1)
#define R2S_ERR_NO_DEVICE (-2)
#define R2S_ERR_HIP (-3)
#define R2S_ERR_UNSUPPORTED (-4)
#define R2S_ERR_NOMEM (-5)

#define R2S_HEX8 0
#define R2S_TET4 1

/* mirrors `mutable struct Grid` (src/MeshGrid/Grid.jl:2-7) */
typedef struct {
    double aabb_min[3];
    double aabb_max[3];
    int64_t N[3];      /* cells per axis; grid points per axis = N+1 */
    double cell_size;
    int64_t ngp;
} r2s_grid;

/* constants that are hard-coded in the reference become fields with the same defaults */
typedef struct {
    double band_factor; /* 1.1  : delta = band_factor*cell_size (sdfOnDensityField.jl:158) */
    int32_t elem_type;  /* R2S_HEX8 / R2S_TET4 (Rho2sdfOptions.element_type, RhoToSDF.jl:20) */
    int32_t device;     /* HIP device ordinal, -1 = current device */
    /* multi-GPU Z partition of r2s_plan_run_dev: zstride <= 1 -> contiguous planes [k_begin,k_end);
     * zstride = G > 1 -> this call computes the 4-plane tile layers t with t % G == zphase of the whole
     * grid (k_begin = 0, k_end = N3+1) and stores them consecutively: local plane 4*i+l holds lattice
     * plane 4*(i*G + zphase) + l; the output then has 4*ceil((layers - zphase)/G) planes. */
    int32_t zstride;
    int32_t zphase;
    /* host-pointer entry points (r2s_sdf, r2s_eval_distances, r2s_sign_detection): number of devices the ONE call
     * fans out over (single process, one host thread per device 0..n_gpus-1, interleaved tile layers, every
     * device sends its layers straight to their place in the caller's array); <= 1 = the device named above */
    int32_t n_gpus;
    /* 0 = the reference's single-thread semantics (bit parity with a Julia run); 1 = order-independent "true
     * minimum" (SURVEY 8(f)4): every valid candidate of a boundary triangle takes part in the minimum (no
     * first-improving-edge break, sdfOnDensityField.jl:769-771; no vertex fall-back only after failures, :777),
     * exact ties go to the lexicographically smaller projection point, and the HEX8 sign is +1 when ANY candidate
     * element holding the point (max|xi| < 1.01) has rho >= rho_t (instead of SignDetection.jl:56-69's
     * improving-sequence rule).  dist_true <= dist_ordered everywhere; see DESIGN.md for the measured deviation. */
    int32_t true_min;
    /* HEX8 sign pass: 1 = no inner-region shortcut (every candidate pair runs its inverse map and the ordered walk of
     * SignDetection.jl:41-68) - for OVERLAPPING / non-conforming meshes, where the shortcut's precondition does not hold
     * (see r2s_sign_detection below).  0 = shortcut on.  The environment variable R2S_SIGN_NO_INNER=1 forces it for every
     * call of the process. */
    int32_t sign_no_inner;
    int32_t reserved_;
} r2s_params;

/* per-call counters (optional; pass NULL) */
typedef struct {
    int64_t n_solid;        /* elements with min(rho_e) >= rho_t           */
    int64_t n_iso;          /* elements crossed by the iso-surface         */
    int64_t n_items;        /* band work items (boundary triangles + iso)  */
    int64_t n_band_entries; /* tile->item list entries                     */
    int64_t n_sign_entries; /* tile->element list entries                  */
    int64_t n_tiles;        /* 4x4x4 voxel tiles in the slab               */
    int64_t n_active_tiles; /* tiles that ran the distance kernel          */
    int64_t n_active_sign_tiles; /* tiles that ran the sign kernel         */
    int64_t n_iso_chunks;   /* 64-voxel chunks swept by iso_project_kernel    */
    int64_t n_any_tiles;    /* tiles that can hold non-sentinel voxels (sparse gather) */
    /* HIP-event times of the last call, measured on the call's stream: mesh prep+items,
     * tile bins, sentinel sweep, iso_project_kernel (ms_main), ordered gather
     * (sdf_tiles_kernel<dist>), sign kernel */
    double ms_prep, ms_bins, ms_fill, ms_main, ms_gather, ms_sign;
    int64_t n_sign_only_tiles; /* tiles without band items whose voxels are all +-1e10 (compressed stitching) */
    /* HEX8: iso_project_hex_pl_kernel alone; ms_main also holds iso_straggler_kernel and iso_sweep_kernel */
    double ms_iso_fast;
    /* HEX8 iso-surface projections (one per iso element x band voxel): pairs the fast Newton-SQP lane machine handed to the
     * complete solver, and runs of the complete solver that ended WITHOUT a KKT point (iteration / non-convex-step caps,
     * cycle): their voxel takes the nearest on-surface iterate - the reference likewise uses whatever NLopt returns and
     * only warns on :FAILURE (ComputeCoordsOnIso.jl:79-86).  SURVEY A6: reported, not hidden. */
    int64_t n_iso_straggler, n_iso_fail;
} r2s_stats;

int r2s_version(void);
const char *r2s_last_error(void);
int r2s_device_count(void);
void r2s_default_params(r2s_params *p);

/* Grid(AABB_min, AABB_max, N_max, margineCells)          src/MeshGrid/Grid.jl:10-34 */
int r2s_grid_make(const double xmin[3], const double xmax[3], int64_t n_max, int64_t margin,
                  r2s_grid *out);

/* noninteractive_sdf_grid_setup(mesh)                    src/MeshGrid/Grid_setup.jl:94-108 */
int r2s_auto_grid(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, int32_t elem_type,
                  r2s_grid *out, double *median_edge);

/* evalDistances(mesh, grid, points, rho_n, rho_t) -> (dist, xp)
 *                                         src/SignedDistances/sdfOnDensityField.jl:139-486
 * dist_out[ngp] (1e10 = untouched), xp_out[ngp][3] or NULL. */
int r2s_eval_distances(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel,
                       const double *rho_n, double rho_t, const r2s_grid *grid,
                       const r2s_params *params, double *dist_out, double *xp_out, r2s_stats *stats);

/* Sign_Detection(mesh, grid, points, rho_n, rho_t) -> signs in {-1,+1}
 *                                         src/SignedDistances/SignDetection.jl:275-283
 * HEX8 precondition of one shortcut: a lattice point inside the convex inner region of an element whose nodal densities
 * all lie on one side of rho_t gets that element's answer without a Newton solve, which equals the reference's ordered
 * walk (SignDetection.jl:41-68) on a CONFORMING mesh (no other element holds the point with a smaller max|xi|).  On
 * overlapping / non-conforming meshes - where the reference's own result depends on which of the overlapping elements
 * comes first - set the environment variable R2S_SIGN_NO_INNER=1: every candidate pair then runs its inverse map
 * (tests/test_parity_gpu.py::test_overlapping_elements_without_the_inner_region_shortcut). */
int r2s_sign_detection(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel,
                       const double *rho_n, double rho_t, const r2s_grid *grid,
                       const r2s_params *params, double *signs_out, r2s_stats *stats);

/* fused `dists .* signs` (RhoToSDF.jl:169-171) - the path rho2sdf() uses */
int r2s_sdf(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, const double *rho_n,
            double rho_t, const r2s_grid *grid, const r2s_params *params, double *sdf_out,
            r2s_stats *stats);

/* ---- the whole of rho2sdf() in one call ---------------------------------------------
 * rho2sdf(taskName, X, IEN, rho; options)                          src/RhoToSDF.jl:116-242
 * minus the file exports (which stay with the caller): mesh volume (:128) -> DenseInNodes (:148) ->
 * find_threshold_for_volume (:151-156) -> evalDistances / Sign_Detection / product (:169-171) ->
 * remove_sdf_artifacts! (:174-208) -> RBFs_smoothing (:222-224).  The mesh goes up once, every stage
 * runs on HBM-resident data, the results come down once.  Fields mirror Rho2sdfOptions (:9-77). */
typedef struct {
    double threshold_density;            /* NaN = nothing: find_threshold_for_volume (TET4: the TET4 iso-volume) */
    double band_factor;                  /* 1.1 */
    double artifact_min_component_ratio; /* 0.01 */
    double rbf_kernel_threshold;         /* 1e-3 (RBFs4Smoothing.jl:328) */
    int32_t elem_type;                   /* R2S_HEX8 / R2S_TET4 */
    int32_t rbf_interp;                  /* 1 */
    int32_t rbf_smooth;                  /* 1 = rbf_grid :same, 2 = :fine */
    int32_t remove_artifacts;            /* 1 */
    int32_t device;                      /* -1 = current */
    int32_t n_gpus;                      /* > 1: devices 0..n_gpus-1: raw SDF on interleaved tile layers, then components and RBF smoothing
                                            slab-distributed (planes move between devices as peer copies over xGMI; nothing is gathered on one device) */
    int32_t skip_rbf;                    /* 1: stop after artifact removal (fine_sdf_out may be NULL) */
    int32_t true_min;                    /* r2s_params.true_min for the raw SDF */
    int32_t sign_no_inner;               /* r2s_params.sign_no_inner for the raw SDF */
    int32_t reserved[3];
} r2s_options;

typedef struct {
    double V_domain, V_frac;             /* calculate_mesh_volume */
    double rho_t;                        /* threshold used */
    int64_t n_flipped;                   /* remove_sdf_artifacts! */
    float level_shift;                   /* LS_Threshold `th` */
    int32_t cg_iters;
    int32_t threshold_iters;
    int32_t pad;
    /* wall-clock milliseconds of the stages of this call */
    double ms_upload, ms_pre, ms_sdf, ms_sdf_kernels, ms_artifacts, ms_rbf, ms_download, ms_total;
} r2s_run_info;

void r2s_default_options(r2s_options *o);

/* grid: from r2s_grid_make / r2s_auto_grid (the caller sizes its result arrays from it).
 * Optional outputs (NULL = not wanted): rho_n_out[nnp] nodal densities, sdf_raw_out[ngp] the field before
 * artifact removal, sdf_dists_out[ngp] the returned `sdf_dists`; fine_sdf_out[prod(N*rbf_smooth+1)] Float32
 * is required unless skip_rbf.  Result arrays from r2s_host_alloc come down by plain DMA. */
int r2s_rho2sdf(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, const double *rho_e,
                const r2s_options *options, const r2s_grid *grid, double *rho_n_out, double *sdf_raw_out,
                double *sdf_dists_out, float *fine_sdf_out, r2s_run_info *info);

/* Pinned host memory for result arrays (the Julia wrapper `unsafe_wrap`s it): device -> host copies into it run
 * at PCIe rate with no staging.  Any other host pointer works too (staged, multi-threaded).  The fused field of
 * r2s_sdf (one device, >= 4 M voxels) does not cross PCIe whole: host threads write the sentinel -1e10 into the caller's
 * array while the device works and only the tiles that can differ from it are transferred (R2S_HOST_SPARSE=0: dense). */
/* Host phases of the calling thread's last host-pointer SDF call on ONE device (r2s_eval_distances / r2s_sign_detection /
 * r2s_sdf), in ms: [0] upload of the mesh, [1] run (launches + the waits of the plan), [2] packing of the non-sentinel tiles
 * and issue of their copies, [3] wait for the host threads' sentinel fill, [4] wait for the copies, [5] scatter into the
 * caller's array (dense download: the whole transfer), [6] the whole call, [7] host threads used.  A diagnostic: where the
 * end-to-end time of a call went (the sparse download's floor is the host's DRAM write bandwidth, not the GPU). */
void r2s_last_host_phases(double out[8]);
void *r2s_host_alloc(size_t bytes);
void r2s_host_free(void *p);

/* ---- device-resident plan API (bench, multi-GPU Z-slabs) ------------------ */
typedef struct r2s_plan r2s_plan;

int r2s_plan_create(int32_t device, r2s_plan **out);
void r2s_plan_destroy(r2s_plan *plan);

/* mode bits for r2s_plan_run_dev */
#define R2S_OUT_DIST 1 /* d_dist[nvox]            */
#define R2S_OUT_SIGN 2 /* d_sign[nvox]            */
#define R2S_OUT_SDF 4  /* d_sdf[nvox] = dist*sign */
#define R2S_OUT_XP 8   /* d_xp[nvox][3]           */

/* One pass of the hot path over the Z-slab of grid planes [k_begin, k_end):
 * all inputs/outputs are device pointers; outputs hold (k_end-k_begin)*(N1+1)*(N2+1)
 * voxels in the reference's x-fastest order.  Work is enqueued on `stream`
 * (hipStream_t as void*, NULL = default stream).  The FIRST call for a set of shapes waits for the stream twice in the
 * middle (item and list sizes are read back); later calls with the same shapes enqueue everything in one go from the
 * sizes of the previous call (a device-side check falls back to the waiting way when they do not hold) and wait once,
 * at the end, when `stats` is requested or an earlier speculated size has to be confirmed. */
int r2s_plan_run_dev(r2s_plan *plan, const double *dX, int64_t nnp, const int64_t *dIEN, int64_t nel,
                     const double *d_rho_n, double rho_t, const r2s_grid *grid,
                     const r2s_params *params, int64_t k_begin, int64_t k_end, int32_t mode,
                     double *d_dist, double *d_sign, double *d_sdf, double *d_xp, void *stream,
                     r2s_stats *stats);

/* ---- sparse stitching of the volume across GPUs (only non-sentinel tiles travel) ---------------
 * After r2s_plan_run_dev with R2S_OUT_SDF on a tile-layer-aligned Z partition (zstride > 1, or k_begin a
 * multiple of 4), pack the 4x4x4 tiles that can differ from the sentinel -1e10: 64 doubles per tile in
 * lane order (x + 4y + 16z) + the tile id in the whole grid's tile numbering.  Receivers pre-fill their
 * volume with the sentinel (r2s_fill_dev) and scatter every rank's tiles (r2s_unpack_tiles_dev). */
int r2s_plan_pack_tiles_dev(r2s_plan *plan, const double *d_local_sdf, double *d_payload, uint32_t *d_ids,
                            int64_t capacity_tiles, int64_t *n_tiles_out, void *stream);
int r2s_unpack_tiles_dev(const double *d_payload, const uint32_t *d_ids, int64_t n_tiles, const r2s_grid *grid,
                         double *d_volume, void *stream);
int r2s_fill_dev(double *d, int64_t n, double value, void *stream);
/* Compressed variant of the same exchange: tiles with band items (r2s_stats.n_active_tiles) travel as 64
 * doubles; tiles that only carry the sign (r2s_stats.n_sign_only_tiles, every voxel +-1e10) travel as one
 * 64-bit mask (bit l set = voxel l of the tile is +1e10) - 12 B instead of 516 B per tile. */
int r2s_plan_pack_tiles2_dev(r2s_plan *plan, const double *d_local_sdf, double *d_payload, uint32_t *d_ids,
                             int64_t capacity_full, uint64_t *d_masks, uint32_t *d_mask_ids, int64_t capacity_mask,
                             int64_t *n_full_out, int64_t *n_mask_out, void *stream);
int r2s_unpack_masks_dev(const uint64_t *d_masks, const uint32_t *d_mask_ids, int64_t n_tiles, const r2s_grid *grid,
                         double magnitude, double *d_volume, void *stream);

/* The whole exchange buffer of the compressed stitching at once: `world` segments of `seglen` doubles, each
 * [n_full, n_mask as two int64 | cap_full x 64 doubles | cap_full ids (uint32, padded to 8 B) | cap_mask masks |
 * cap_mask mask ids (uint32, padded)].  The counts are read from the segment headers on the device (no host round
 * trip, two launches for all ranks); a segment whose counts exceed the capacities is skipped. */
int r2s_unpack_segments_dev(const double *d_buf, int32_t world, int64_t seglen, int64_t cap_full, int64_t cap_mask,
                            const r2s_grid *grid, double magnitude, double *d_volume, void *stream);

/* ---- pre-stage: mesh volume, nodal densities, volume-preserving threshold ------------- */

/* calculate_mesh_volume(X, IEN, rho, HEX8) -> (V_domain, V_frac)     src/MeshGrid/MeshVolume.jl:4-42 */
int r2s_mesh_volume(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, int32_t elem_type,
                    const double *rho_e, int32_t device, double *V_domain, double *V_frac);

/* DenseInNodes(mesh, rho) -> rho_n[nnp]                              src/MeshGrid/NodalDensities.jl:89-218 */
int r2s_dense_in_nodes(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, int32_t elem_type,
                       const double *rho_e, int32_t device, double *rho_n_out);

/* find_threshold_for_volume(mesh, rho_n, tol=1e-4, maxit=60) with target = V_domain*V_frac
 *                                                                    src/MeshGrid/Isocontour_volume.jl:77-154
 * returns R2S_ERR_ARG ("outside the possible range", :93-95) like the reference's error(). */
int r2s_find_threshold(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, const double *rho_n,
                       double target_volume, double tol, int32_t maxit, int32_t device, double *rho_t_out,
                       int32_t *iters_out);

/* the same bisection for either element type.  TET4 has no iso-volume in the reference
 * (calculate_isocontour_volume hard-codes 8 nodes, Isocontour_volume.jl:27-38): it is assembled
 * from the reference's own pieces - the skip / whole / cut classification of :40-52 with the collapsed-cube
 * rule of MeshVolume.jl:75-117 (3^3 points for whole elements, 15^3 with the point test for cut ones) - so that
 * volume(0) == V_domain of calculate_mesh_volume for TET4. */
int r2s_find_threshold_et(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, int32_t elem_type,
                          const double *rho_n, double target_volume, double tol, int32_t maxit, int32_t device,
                          double *rho_t_out, int32_t *iters_out);

/* calculate_isocontour_volume(mesh, nodal_values, iso_threshold)     src/MeshGrid/Isocontour_volume.jl:1-75 */
int r2s_isocontour_volume(const double *X, int64_t nnp, const int64_t *IEN, int64_t nel, int32_t elem_type,
                          const double *rho_n, double threshold, int32_t device, double *volume_out);

/* ---- post-processing ---------------------------------------------------------------- */

/* remove_sdf_artifacts!(sdf, grid; threshold, min_component_ratio) -> nodes flipped
 *                                                src/SignedDistances/SdfArtifactRemoval.jl:134-245 */
int r2s_remove_artifacts(double *sdf_inout, const r2s_grid *grid, double threshold, double min_ratio,
                         int32_t device, int64_t *n_flipped);
int r2s_remove_artifacts_dev(double *d_sdf, const r2s_grid *grid, double threshold, double min_ratio,
                             void *stream, int64_t *n_flipped);

/* calculate_volume_from_sdf(sdf::Array{Float32,3}, grid; iso_threshold, detailed_quad_order)
 *                                                src/SdfSmoothing/CalcVolumeFromSDF.jl:26-125
 * sdf is (nx,ny,nz) x-fastest; edge = spacing of the (cubic-cell) grid. */
int r2s_volume_from_sdf(const float *sdf, int64_t nx, int64_t ny, int64_t nz, float edge, float iso,
                        int32_t quad_order, int32_t device, float *vol_out);

/* RBFs_smoothing(mesh, dist, grid, is_interp, smooth, taskName, threshold=1e-3) -> fine_sdf
 *                                                src/SdfSmoothing/RBFs4Smoothing.jl:321-377
 * target_volume = mesh.V_frac*mesh.V_domain; fine_sdf_out has prod(N*smooth+1) Float32 values
 * (x fastest); fine_grid is origin AABB_min + spacing (AABB_max[1]-AABB_min[1])/(N[1]*smooth) and is
 * materialised by the caller.  Optional outputs: level shift `th`, CG iterations, coarse LSF. */
int r2s_rbf_smooth(const double *sdf, const r2s_grid *grid, int32_t is_interp, int32_t smooth,
                   double kernel_threshold, double target_volume, int32_t device, float *fine_sdf_out,
                   float *level_shift_out, int32_t *cg_iters_out, float *lsf_out);

/* device-resident variant (chaining the stages without PCIe): d_sdf (Float64) and d_fine_out (Float32) are
 * device pointers on the current device; waits for `stream` first, synchronous on return. */
int r2s_rbf_smooth_dev(const double *d_sdf, const r2s_grid *grid, int32_t is_interp, int32_t smooth,
                       double kernel_threshold, double target_volume, float *d_fine_out, float *level_shift_out,
                       int32_t *cg_iters_out, void *stream);

/* ---- on-disk output ------------------------------------------------------------------ */

/* exportSdfToVTI(filename, grid, values, value_label, smooth)       src/DataExport/ExportToVTI.jl:22-67
 * VTK ImageData: dimensions N*smooth+1 (smooth = 0: N+1, i.e. `nothing`), Origin = AABB_min, Spacing =
 * cell_size(/smooth), one point-data array `value_label` (Float32 or Float64, x fastest).  ".vti" is appended
 * when missing.  Host pointers; no device needed. */
int r2s_export_vti(const char *filename, const r2s_grid *grid, const void *values, int32_t is_float32,
                   int64_t n_values, const char *value_label, int32_t smooth);

/* the same data set with the payload deflated block-wise (compressor="vtkZLibDataCompressor", WriteVTK's default
 * on-disk form): level 0 = raw appended (r2s_export_vti), 1..9 = zlib level */
int r2s_export_vti_z(const char *filename, const r2s_grid *grid, const void *values, int32_t is_float32,
                     int64_t n_values, const char *value_label, int32_t smooth, int32_t level);

/* exportToVTU(fileName, X, IEN, VTK_CODE, rho)                       src/DataExport/ExportToVTU.jl:2-99
 * ASCII UnstructuredGrid of the mesh (VTK_CODE 12 = hexahedron, 10 = tetra), optional nodal "density". */
int r2s_export_vtu(const char *filename, const double *X, int64_t nnp, const int64_t *IEN, int64_t nel,
                   int32_t nen, int32_t vtk_code, const double *rho_n);

/* import_vtu_mesh(vtu_file) -> (X, IEN, rho)                         src/DataImport/VTUImport.jl:22-112
 * Mesh of an ASCII UnstructuredGrid file: hexahedra (VTK type 12, 8 nodes) or tetrahedra (10, 4 nodes), other
 * cells are skipped and counted; IEN comes back 1-based.  Element densities (:117-226): the first cell-data
 * field named density / rho / volfrac / ... (the reference's list), else the first cell-data field, else 1.0;
 * taken by position, padded with 1.0.  The arrays are malloc'ed by the library: release them with
 * r2s_free_vtu_mesh.  Every DataArray encoding ReadVTK reads is understood: ascii, binary (inline base64) and
 * appended (raw / base64), plain or zlib-compressed, header_type UInt32 / UInt64, little endian. */
typedef struct r2s_vtu_mesh {
    int64_t nnp, nel;
    int32_t nen;         /* 8 or 4 */
    int32_t elem_type;   /* R2S_HEX8 / R2S_TET4 */
    int64_t n_skipped;   /* cells of other types */
    double *X;           /* [nnp][3] */
    int64_t *IEN;        /* [nel][nen], 1-based */
    double *rho;         /* [nel] */
    char density_field[64]; /* cell-data field the densities came from ("" = default 1.0) */
} r2s_vtu_mesh;
int r2s_import_vtu(const char *filename, r2s_vtu_mesh *out);
void r2s_free_vtu_mesh(r2s_vtu_mesh *mesh);

/* MeshInformations(matread(file)) -> (X, IEN, rho)                    src/MeshGrid/MeshInformations.jl:3-12
 * MATLAB level-5 .mat file (compressed or not) with `rho` [nel] and the struct `msh` holding X (3 x nnp) and IEN
 * (nen x nel; IEN + 1 is returned, as the reference adds 1, :8).  Same result struct and ownership as
 * r2s_import_vtu (release with r2s_free_vtu_mesh).  v7.3 (HDF5) files are refused. */
int r2s_import_mat(const char *filename, r2s_vtu_mesh *out);

/* frees what the library keeps between calls: the per-device host sessions of the host-pointer entry points (plan,
 * device copies of the mesh, output volumes, pinned staging buffers) and the shared work buffers of the smoothing stage.
 * Must not run concurrently with any other call of the library (it destroys the sessions other calls lock). */
void r2s_release_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* RHO2SDF_HIP_H */
