# Rho2sdfHIP.jl - thin `ccall` layer that re-points Rho2sdf.jl's hot path at
# librho2sdf_hip.so (C ABI: include/rho2sdf_hip.h).
#
# Two levels:
#   * `rho2sdf(taskName, X, IEN, rho; options)` itself (src/RhoToSDF.jl:116-242) is overridden: ONE ccall
#     (r2s_rho2sdf) uploads the mesh once, runs mesh volume -> DenseInNodes -> threshold -> raw SDF -> artifact
#     removal -> RBF smoothing on HBM-resident data and brings `sdf_dists` and `fine_sdf` down once.  `generateGridPoints` (3.2 GB at 512^3, :166) and the projection points `xp` (:169, discarded by the
#     reference) are never materialised.  All file exports stay Julia code and run on the returned arrays.
#   * every function `rho2sdf` and the reference's tests call directly keeps a ccall-backed method with the
#     reference's signature (leaf overrides), for callers that use the stages one by one.
#
# NOTE: there is no Julia toolchain in the build image, so this file is delivered as reviewed
# text; the same ABI is exercised by rho2sdf.jl_amd/api.py (ctypes) in the test-suite.
#
# Usage (in the reference checkout):
#     include("Rho2sdfHIP.jl"); using .Rho2sdfHIP
#     Rho2sdfHIP.enable!("/path/to/librho2sdf_hip.so"; n_gpus = 1)     # overrides the methods below
module Rho2sdfHIP

using Rho2sdf
using Rho2sdf.MeshGrid
using Rho2sdf.SignedDistances
using Rho2sdf.SdfSmoothing
using Rho2sdf.ElementTypes
using Rho2sdf.ShapeFunctions
using Rho2sdf.DataExport

const LIB = Ref{String}("librho2sdf_hip.so")
const SIGN_NO_INNER = Ref(false)      # true: HEX8 sign pass without the inner-region shortcut (overlapping / non-conforming meshes)
const N_GPUS = Ref{Int32}(1)          # devices one call fans out over (single process; r2s_params.n_gpus)

# mirrors r2s_grid / Grid (src/MeshGrid/Grid.jl:2-7)
struct R2SGrid
    aabb_min::NTuple{3,Float64}
    aabb_max::NTuple{3,Float64}
    N::NTuple{3,Int64}
    cell_size::Float64
    ngp::Int64
end
R2SGrid(g::MeshGrid.Grid) = R2SGrid(Tuple(g.AABB_min), Tuple(g.AABB_max), Tuple(g.N), g.cell_size, g.ngp)

struct R2SParams                      # mirrors r2s_params
    band_factor::Float64
    elem_type::Int32
    device::Int32
    zstride::Int32                    # device-pointer plan API only: interleaved tile layers
    zphase::Int32
    n_gpus::Int32                     # host-pointer entry points: devices 0..n_gpus-1 share the call
    true_min::Int32                   # 1 = order-independent semantics (SURVEY 8(f)4); 0 = the reference's
    sign_no_inner::Int32              # 1 = HEX8 sign pass without the inner-region shortcut (overlapping / non-conforming meshes)
    reserved_::Int32
end
etype(::Type{HEX8}) = Int32(0)
etype(::Type{TET4}) = Int32(1)
params(::Type{T}; band_factor = 1.1) where {T} = R2SParams(band_factor, etype(T), -1, 0, 0, N_GPUS[], 0, SIGN_NO_INNER[] ? 1 : 0, 0)

struct R2SOptions                     # mirrors r2s_options (= Rho2sdfOptions, RhoToSDF.jl:9-77)
    threshold_density::Float64        # NaN = nothing
    band_factor::Float64
    artifact_min_component_ratio::Float64
    rbf_kernel_threshold::Float64
    elem_type::Int32
    rbf_interp::Int32
    rbf_smooth::Int32
    remove_artifacts::Int32
    device::Int32
    n_gpus::Int32
    skip_rbf::Int32
    true_min::Int32
    sign_no_inner::Int32
    reserved::NTuple{3,Int32}
end

struct R2SRunInfo                     # mirrors r2s_run_info
    V_domain::Float64; V_frac::Float64; rho_t::Float64
    n_flipped::Int64
    level_shift::Float32; cg_iters::Int32; threshold_iters::Int32; pad::Int32
    ms_upload::Float64; ms_pre::Float64; ms_sdf::Float64; ms_sdf_kernels::Float64
    ms_artifacts::Float64; ms_rbf::Float64; ms_download::Float64; ms_total::Float64
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:r2s_last_error, LIB[]), Cstring, ()))
    error("rho2sdf_hip: $msg")          # same behaviour as the reference's error(...)
end

# Result arrays.  Ordinary Julia arrays by default: the library brings results down through pinned staging buffers
# with a multi-threaded copy, within 5 % of a plain DMA (24.9 vs 24.1 ms for 1 GB).  PINNED[] = true allocates them
# with r2s_host_alloc instead (the copy is then one DMA) - pinning 1.6 GB costs ~0.2 s per call, so that only pays
# for callers that keep and reuse the arrays.  The array owns nothing; a finalizer returns the block to the library.
const PINNED = Ref(false)
function pinned(::Type{T}, dims::Int...) where {T}
    PINNED[] || return Array{T}(undef, dims...)
    n = prod(dims)
    p = ccall((:r2s_host_alloc, LIB[]), Ptr{Cvoid}, (Csize_t,), max(n, 1) * sizeof(T))
    p == C_NULL && error("rho2sdf_hip: " * unsafe_string(ccall((:r2s_last_error, LIB[]), Cstring, ())))
    a = unsafe_wrap(Array, Ptr{T}(p), dims; own = false)
    finalizer(_ -> ccall((:r2s_host_free, LIB[]), Cvoid, (Ptr{Cvoid},), p), a)
    return a
end

# fine_grid.  The reference materialises one heap Vector{Float32} per grid point (create_smooth_grid, RBFs4Smoothing.jl:60-74:
# 134 M allocations = several seconds and 10 GB at 512^3) although only fine_grid[2,1,1] - fine_grid[1,1,1] is ever read
# (CalcVolumeFromSDF.jl:37).  LAZY_GRID[] = true returns the same points as an AbstractArray{Vector{Float32},3} over the three
# coordinate vectors instead - element for element equal to the reference's array (same explicit Float32 arithmetic), built
# on demand.  Default false: the reference's own calculate_volume_from_sdf is declared for Array{Vector{Float32},3} and would
# not accept it (the _hip method below does).
const LAZY_GRID = Ref(false)
struct LazyFineGrid <: AbstractArray{Vector{Float32},3}
    x::Vector{Float32}
    y::Vector{Float32}
    z::Vector{Float32}
end
Base.size(g::LazyFineGrid) = (length(g.x), length(g.y), length(g.z))
Base.getindex(g::LazyFineGrid, i::Int, j::Int, k::Int) = Float32[g.x[i], g.y[j], g.z[k]]
Base.IndexStyle(::Type{LazyFineGrid}) = IndexCartesian()
function fine_grid_of(grid::MeshGrid.Grid, smooth::Int)
    LAZY_GRID[] || return SdfSmoothing.create_smooth_grid(grid, smooth)[2]
    nx, ny, nz = (grid.N * smooth) .+ 1                                   # RBFs4Smoothing.jl:61-73, operation for operation
    xmin, ymin, zmin = Float32.(grid.AABB_min)
    xmax = Float32(grid.AABB_max[1])
    dx = (xmax - xmin) / (nx - 1)
    return LazyFineGrid([xmin + (i - 1) * dx for i in 1:nx], [ymin + (j - 1) * dx for j in 1:ny], [zmin + (k - 1) * dx for k in 1:nz])
end

# ---------------------------------------------------------------------------------------------------
# rho2sdf (src/RhoToSDF.jl:116-242): same signature, same return value, same files written
# ---------------------------------------------------------------------------------------------------
function rho2sdf_hip(taskName::String, X::Vector{Vector{Float64}}, IEN::Vector{Vector{Int64}},
                     rho::Vector{Float64}; options::Rho2sdfOptions = Rho2sdfOptions())
    T = options.element_type
    shape_func = coords -> shape_functions(T, coords)
    mesh = Mesh(X, IEN, rho, shape_func; element_type = T)                         # :128 (flat X / IEN for the ccall)
    options.export_input_data && InputDataToVTU(mesh, taskName * "-input_data")    # :137
    sdf_grid = options.sdf_grid_setup == :manual ? interactive_sdf_grid_setup(mesh) :
               noninteractive_sdf_grid_setup(mesh)                                 # :141-145
    smooth = options.rbf_grid == :same ? 1 : 2                                     # :222
    o = R2SOptions(options.threshold_density === nothing ? NaN : Float64(options.threshold_density), 1.1,
                   options.artifact_min_component_ratio, 1e-3, etype(T), Int32(options.rbf_interp), Int32(smooth),
                   Int32(options.remove_artifacts), Int32(-1), N_GPUS[], Int32(0), Int32(0), Int32(SIGN_NO_INNER[] ? 1 : 0),
                   (0, 0, 0))
    ρₙ = Vector{Float64}(undef, mesh.nnp)
    sdf_dists = pinned(Float64, sdf_grid.ngp)
    fine_sdf = pinned(Float32, ((sdf_grid.N .* smooth) .+ 1)...)
    want_raw = options.remove_artifacts && options.export_analysis                 # :181-189 exports the field before cleanup
    sdf_raw = want_raw ? pinned(Float64, sdf_grid.ngp) : nothing
    info = Ref{R2SRunInfo}()
    check(ccall((:r2s_rho2sdf, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Ref{R2SOptions}, Ref{R2SGrid},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}, Ref{R2SRunInfo}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, rho, Ref(o), Ref(R2SGrid(sdf_grid)),
                ρₙ, want_raw ? sdf_raw : C_NULL, sdf_dists, fine_sdf, info))
    element_name = string(T.name.name)
    B = round(sdf_grid.cell_size, digits = 4)
    if options.export_nodal_densities                                              # :159-162
        exportToVTU(taskName * "_nodal_densities.vtu", X, IEN, T == HEX8 ? 12 : 10, ρₙ)
    end
    if want_raw                                                                    # :181-206
        exportSdfToVTI(taskName * "_SDF_raw_$(element_name)_B-$(B).vti", sdf_grid, sdf_raw, "distance")
        info[].n_flipped > 0 &&
            exportSdfToVTI(taskName * "_SDF_cleaned_$(element_name)_B-$(B).vti", sdf_grid, sdf_dists, "distance")
    end
    if options.export_raw_sdf                                                      # :211-219
        exportSdfToVTI(taskName * "_SDF_$(element_name)_CellSize-" * string(B) * ".vti", sdf_grid, sdf_dists, "distance")
    end
    fine_grid = fine_grid_of(sdf_grid, smooth)                                     # the point list of RBFs4Smoothing.jl:341
    Rho2sdf.export_sdf_results_with_element_type(fine_sdf, fine_grid, sdf_grid, taskName, smooth,
                                                 options.rbf_interp, T)            # :230-238
    return (fine_sdf, fine_grid, sdf_grid, sdf_dists)
end

# ---------------------------------------------------------------------------------------------------
# leaf functions
# ---------------------------------------------------------------------------------------------------
# evalDistances (src/SignedDistances/sdfOnDensityField.jl:139-486).  The reference returns (dist, xp); its only
# caller throws xp away (RhoToSDF.jl:169), and xp is 24 B/voxel of PCIe traffic: it is produced only when
# `want_xp = true` is passed (plot_projection_points_and_lines implies it), otherwise an empty 3 x 0 matrix.
function evalDistances_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, points, ρₙ::Vector{Float64}, ρₜ::Float64;
                           band_factor = 1.1, want_xp::Bool = false, plot_projection_points_and_lines::Bool = false,
                           kwargs...) where {T}
    want_xp |= plot_projection_points_and_lines
    dist = pinned(Float64, grid.ngp)
    xp = want_xp ? pinned(Float64, 3, grid.ngp) : Matrix{Float64}(undef, 3, 0)
    g = Ref(R2SGrid(grid)); p = Ref(params(T; band_factor))
    check(ccall((:r2s_eval_distances, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, dist, want_xp ? xp : C_NULL, C_NULL))
    return dist, xp
end

# Sign_Detection (src/SignedDistances/SignDetection.jl:275-283)
function Sign_Detection_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, points, ρₙ::Vector{Float64}, ρₜ::Float64) where {T}
    signs = pinned(Float64, grid.ngp)
    g = Ref(R2SGrid(grid)); p = Ref(params(T))
    check(ccall((:r2s_sign_detection, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, signs, C_NULL))
    return signs
end

# fused dists .* signs (src/RhoToSDF.jl:169-171)
function sdf_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, ρₙ::Vector{Float64}, ρₜ::Float64; band_factor = 1.1) where {T}
    sdf = pinned(Float64, grid.ngp)
    g = Ref(R2SGrid(grid)); p = Ref(params(T; band_factor))
    check(ccall((:r2s_sdf, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, sdf, C_NULL))
    return sdf
end

# calculate_mesh_volume (src/MeshGrid/MeshVolume.jl:4-42) -> [V_domain, V_frac]
function calculate_mesh_volume_hip(X::Matrix{Float64}, IEN::Matrix{Int64}, rho::Vector{Float64}, ::Type{T}) where {T}
    vd = Ref{Float64}(0.0); vf = Ref{Float64}(0.0)
    check(ccall((:r2s_mesh_volume, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Int32, Ptr{Float64}, Int32, Ref{Float64}, Ref{Float64}),
                X, size(X, 2), IEN, size(IEN, 2), etype(T), rho, Int32(-1), vd, vf))
    return [vd[], vf[]]
end

# DenseInNodes (src/MeshGrid/NodalDensities.jl:89-108)
function DenseInNodes_hip(mesh::Mesh{T}, rho::Vector{Float64}) where {T}
    length(rho) == mesh.nel || error("length of element densities does not match number of elements")
    ρₙ = Vector{Float64}(undef, mesh.nnp)
    check(ccall((:r2s_dense_in_nodes, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float64}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, etype(T), rho, Int32(-1), ρₙ))
    return ρₙ
end

# find_threshold_for_volume(mesh, nodal_values, tolerance = 1e-4, max_iterations = 60)
# (src/MeshGrid/Isocontour_volume.jl:77-80: positional, like the reference).  TET4 meshes, for which the reference
# has no iso-volume, use the library's TET4 rule (include/rho2sdf_hip.h, r2s_find_threshold_et).
function find_threshold_for_volume_hip(mesh::Mesh{T}, ρₙ::Vector{Float64}, tolerance::Float64 = 1e-4,
                                       max_iterations::Int = 60) where {T}
    ρₜ = Ref{Float64}(0.0); its = Ref{Int32}(0)
    check(ccall((:r2s_find_threshold_et, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Int32, Ptr{Float64}, Float64, Float64, Int32, Int32,
                 Ref{Float64}, Ref{Int32}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, etype(T), ρₙ, mesh.V_domain * mesh.V_frac, tolerance,
                Int32(max_iterations), Int32(-1), ρₜ, its))
    return ρₜ[]
end

# calculate_isocontour_volume(mesh, nodal_values, iso_threshold) (src/MeshGrid/Isocontour_volume.jl:1-75)
function calculate_isocontour_volume_hip(mesh::Mesh{T}, ρₙ::Vector{Float64}, iso_threshold::Float64) where {T}
    v = Ref{Float64}(0.0)
    check(ccall((:r2s_isocontour_volume, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Int32, Ptr{Float64}, Float64, Int32, Ref{Float64}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, etype(T), ρₙ, iso_threshold, Int32(-1), v))
    return v[]
end

# remove_sdf_artifacts! (src/SignedDistances/SdfArtifactRemoval.jl:134-245) -> nodes flipped
function remove_sdf_artifacts_hip!(sdf::Vector{Float64}, grid::MeshGrid.Grid; threshold = 0.0,
                                   min_component_ratio = 0.01)
    length(sdf) == grid.ngp || error("SDF values length ($(length(sdf))) doesn't match grid points ($(grid.ngp))")   # :141-143
    n = Ref{Int64}(0)
    check(ccall((:r2s_remove_artifacts, LIB[]), Cint,
                (Ptr{Float64}, Ref{R2SGrid}, Float64, Float64, Int32, Ref{Int64}),
                sdf, Ref(R2SGrid(grid)), threshold, min_component_ratio, Int32(-1), n))
    return n[]
end

# calculate_volume_from_sdf (src/SdfSmoothing/CalcVolumeFromSDF.jl:26-125); `grid` is the reference's array of
# per-voxel coordinate vectors - only the spacing is used (:36-39)
function calculate_volume_from_sdf_hip(sdf::Array{Float32,3}, grid::AbstractArray{Vector{Float32},3}; iso_threshold = 0.0f0,
                                       detailed_quad_order = 9)
    @assert size(grid) == size(sdf) "Dimensions of fine_sdf and fine_grid must match"
    d = grid[2, 1, 1] .- grid[1, 1, 1]
    edge = sqrt(sum(d .* d))                                             # norm(edge_vector), :37-38
    v = Ref{Float32}(0.0f0)
    check(ccall((:r2s_volume_from_sdf, LIB[]), Cint,
                (Ptr{Float32}, Int64, Int64, Int64, Float32, Float32, Int32, Int32, Ref{Float32}),
                sdf, size(sdf, 1), size(sdf, 2), size(sdf, 3), edge, Float32(iso_threshold),
                Int32(detailed_quad_order), Int32(-1), v))
    return v[]
end

# RBFs_smoothing (src/SdfSmoothing/RBFs4Smoothing.jl:321-377) -> (fine_sdf::Array{Float32,3}, fine_grid)
function RBFs_smoothing_hip(mesh::Mesh, dist::Vector{Float64}, grid::MeshGrid.Grid, is_interp::Bool,
                            smooth::Int, taskName::String, threshold::Float64 = 1e-3)
    dim = (grid.N .* smooth) .+ 1
    fine_grid = fine_grid_of(grid, smooth)                              # point list only (:341), stays in Julia
    fine = pinned(Float32, dim...)
    check(ccall((:r2s_rbf_smooth, LIB[]), Cint,
                (Ptr{Float64}, Ref{R2SGrid}, Int32, Int32, Float64, Float64, Int32, Ptr{Float32}, Ptr{Float32},
                 Ptr{Int32}, Ptr{Float32}),
                dist, Ref(R2SGrid(grid)), Int32(is_interp), Int32(smooth), threshold,
                mesh.V_frac * mesh.V_domain, Int32(-1), fine, C_NULL, C_NULL, C_NULL))
    return fine, fine_grid
end

"Replace the reference methods by the HIP-backed ones (method overwrite).  `n_gpus` > 1: every call fans out over
devices 0..n_gpus-1 inside the library (single Julia process, no MPI)."
function enable!(libpath::AbstractString = LIB[]; n_gpus::Integer = 1)
    LIB[] = libpath
    N_GPUS[] = Int32(n_gpus)
    @eval Rho2sdf begin
        rho2sdf(taskName::String, X::Vector{Vector{Float64}}, IEN::Vector{Vector{Int64}}, rho::Vector{Float64};
                options::Rho2sdfOptions = Rho2sdfOptions()) = $(rho2sdf_hip)(taskName, X, IEN, rho; options = options)
    end
    @eval SignedDistances begin
        evalDistances(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64; kw...) =
            $(evalDistances_hip)(mesh, grid, points, ρₙ, ρₜ; want_xp = true, kw...)   # direct callers get the reference's (dist, xp)
        Sign_Detection(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64) =
            $(Sign_Detection_hip)(mesh, grid, points, ρₙ, ρₜ)
        remove_sdf_artifacts!(sdf::Vector{Float64}, grid::Grid; kw...) = $(remove_sdf_artifacts_hip!)(sdf, grid; kw...)
    end
    @eval MeshGrid begin
        DenseInNodes(mesh::Mesh, rho::Vector{Float64}) = $(DenseInNodes_hip)(mesh, rho)
        find_threshold_for_volume(mesh::Mesh, ρₙ::Vector{Float64}, tolerance::Float64 = 1e-4, max_iterations::Int = 60) =
            $(find_threshold_for_volume_hip)(mesh, ρₙ, tolerance, max_iterations)
        calculate_isocontour_volume(mesh::Mesh, ρₙ::Vector{Float64}, iso_threshold::Float64) =
            $(calculate_isocontour_volume_hip)(mesh, ρₙ, iso_threshold)
    end
    @eval SdfSmoothing begin
        RBFs_smoothing(mesh::Mesh, dist::Vector, grid::Grid, is_interp::Bool, smooth::Int, taskName::String,
                       threshold::Float64 = 1e-3) = $(RBFs_smoothing_hip)(mesh, Vector{Float64}(dist), grid, is_interp, smooth, taskName, threshold)
        calculate_volume_from_sdf(sdf::Array{Float32,3}, grid::Array{Vector{Float32},3}; kw...) =
            $(calculate_volume_from_sdf_hip)(sdf, grid; kw...)
    end
    return nothing
end

"Free the device buffers the library keeps between calls (plan, volumes, staging, RBF matrix)."
release!() = ccall((:r2s_release_cache, LIB[]), Cvoid, ())

end # module
