# Rho2sdfHIP.jl - thin `ccall` layer that re-points Rho2sdf.jl's hot path at
# librho2sdf_hip.so (C ABI: include/rho2sdf_hip.h).  `rho2sdf(taskName, X, IEN, rho; options)`
# and all file outputs are untouched: only the bodies of the functions it calls are replaced.
#
# NOTE: there is no Julia toolchain in the build image, so this file is delivered as reviewed
# text; the same ABI is exercised by rho2sdf.jl_amd/api.py (ctypes) in the test-suite.
#
# Usage (in the reference checkout):
#     include("Rho2sdfHIP.jl"); using .Rho2sdfHIP
#     Rho2sdfHIP.enable!("/path/to/librho2sdf_hip.so")     # overrides the methods below
module Rho2sdfHIP

using Rho2sdf
using Rho2sdf.MeshGrid
using Rho2sdf.SignedDistances
using Rho2sdf.SdfSmoothing
using Rho2sdf.ElementTypes

const LIB = Ref{String}("librho2sdf_hip.so")

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
    zstride::Int32                    # multi-GPU only: interleaved tile layers (0/1 = contiguous planes)
    zphase::Int32
    reserved::NTuple{2,Int32}
end
params(::Type{HEX8}; band_factor = 1.1) = R2SParams(band_factor, 0, -1, 1, 0, (0, 0))
params(::Type{TET4}; band_factor = 1.1) = R2SParams(band_factor, 1, -1, 1, 0, (0, 0))
etype(::Type{HEX8}) = Int32(0)
etype(::Type{TET4}) = Int32(1)

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:r2s_last_error, LIB[]), Cstring, ()))
    error("rho2sdf_hip: $msg")          # same behaviour as the reference's error(...)
end

# evalDistances (src/SignedDistances/sdfOnDensityField.jl:139-486)
function evalDistances_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, points::Matrix, ρₙ::Vector{Float64},
                           ρₜ::Float64; band_factor = 1.1, kwargs...) where {T}
    dist = Vector{Float64}(undef, grid.ngp)
    xp = Matrix{Float64}(undef, 3, grid.ngp)
    g = Ref(R2SGrid(grid)); p = Ref(params(T; band_factor))
    check(ccall((:r2s_eval_distances, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, dist, xp, C_NULL))
    return dist, xp
end

# Sign_Detection (src/SignedDistances/SignDetection.jl:275-283)
function Sign_Detection_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, points::Matrix, ρₙ::Vector{Float64},
                            ρₜ::Float64) where {T}
    signs = Vector{Float64}(undef, grid.ngp)
    g = Ref(R2SGrid(grid)); p = Ref(params(T))
    check(ccall((:r2s_sign_detection, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, signs, C_NULL))
    return signs
end

# fused dists .* signs (src/RhoToSDF.jl:169-171)
function sdf_hip(mesh::Mesh{T}, grid::MeshGrid.Grid, ρₙ::Vector{Float64}, ρₜ::Float64) where {T}
    sdf = Vector{Float64}(undef, grid.ngp)
    g = Ref(R2SGrid(grid)); p = Ref(params(T))
    check(ccall((:r2s_sdf, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Ref{R2SGrid},
                 Ref{R2SParams}, Ptr{Float64}, Ptr{Cvoid}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, ρₜ, g, p, sdf, C_NULL))
    return sdf
end

# ---- stages around the raw SDF (same names, arguments and error behaviour as the reference) ----

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

# find_threshold_for_volume (src/MeshGrid/Isocontour_volume.jl:77-154)
function find_threshold_for_volume_hip(mesh::Mesh{HEX8}, ρₙ::Vector{Float64}; tol = 1e-4, maxit = 60)
    ρₜ = Ref{Float64}(0.0); its = Ref{Int32}(0)
    check(ccall((:r2s_find_threshold, LIB[]), Cint,
                (Ptr{Float64}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Float64, Int32, Int32,
                 Ref{Float64}, Ref{Int32}),
                mesh.X, mesh.nnp, mesh.IEN, mesh.nel, ρₙ, mesh.V_domain * mesh.V_frac, tol, Int32(maxit),
                Int32(-1), ρₜ, its))
    return ρₜ[]
end

# remove_sdf_artifacts! (src/SignedDistances/SdfArtifactRemoval.jl:134-245) -> nodes flipped
function remove_sdf_artifacts_hip!(sdf::Vector{Float64}, grid::MeshGrid.Grid; threshold = 0.0,
                                   min_component_ratio = 0.01)
    n = Ref{Int64}(0)
    check(ccall((:r2s_remove_artifacts, LIB[]), Cint,
                (Ptr{Float64}, Ref{R2SGrid}, Float64, Float64, Int32, Ref{Int64}),
                sdf, Ref(R2SGrid(grid)), threshold, min_component_ratio, Int32(-1), n))
    return n[]
end

# calculate_volume_from_sdf (src/SdfSmoothing/CalcVolumeFromSDF.jl:26-125)
function calculate_volume_from_sdf_hip(sdf::Array{Float32,3}, edge::Float32; iso_threshold = 0.0f0,
                                       detailed_quad_order = 9)
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
    (_, fine_grid) = SdfSmoothing.create_smooth_grid(grid, smooth)      # point list only (:341), stays in Julia
    fine = Array{Float32,3}(undef, dim...)
    check(ccall((:r2s_rbf_smooth, LIB[]), Cint,
                (Ptr{Float64}, Ref{R2SGrid}, Int32, Int32, Float64, Float64, Int32, Ptr{Float32}, Ptr{Float32},
                 Ptr{Int32}, Ptr{Float32}),
                dist, Ref(R2SGrid(grid)), Int32(is_interp), Int32(smooth), threshold,
                mesh.V_frac * mesh.V_domain, Int32(-1), fine, C_NULL, C_NULL, C_NULL))
    return fine, fine_grid
end

"Replace the reference methods by the HIP-backed ones (method overwrite)."
function enable!(libpath::AbstractString = LIB[])
    LIB[] = libpath
    @eval SignedDistances begin
        evalDistances(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64; kw...) =
            $(evalDistances_hip)(mesh, grid, points, ρₙ, ρₜ; kw...)
        Sign_Detection(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64) =
            $(Sign_Detection_hip)(mesh, grid, points, ρₙ, ρₜ)
        remove_sdf_artifacts!(sdf::Vector{Float64}, grid::Grid; kw...) = $(remove_sdf_artifacts_hip!)(sdf, grid; kw...)
    end
    @eval MeshGrid begin
        DenseInNodes(mesh::Mesh, rho::Vector{Float64}) = $(DenseInNodes_hip)(mesh, rho)
        find_threshold_for_volume(mesh::Mesh, ρₙ::Vector{Float64}; kw...) = $(find_threshold_for_volume_hip)(mesh, ρₙ; kw...)
    end
    @eval SdfSmoothing begin
        RBFs_smoothing(mesh::Mesh, dist::Vector, grid::Grid, is_interp::Bool, smooth::Int, taskName::String,
                       threshold::Float64 = 1e-3) = $(RBFs_smoothing_hip)(mesh, Vector{Float64}(dist), grid, is_interp, smooth, taskName, threshold)
    end
    return nothing
end

end # module
