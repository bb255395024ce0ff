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

struct R2SParams
    band_factor::Float64
    elem_type::Int32
    device::Int32
    reserved::NTuple{4,Int32}
end
params(::Type{HEX8}; band_factor = 1.1) = R2SParams(band_factor, 0, -1, (0, 0, 0, 0))
params(::Type{TET4}; band_factor = 1.1) = R2SParams(band_factor, 1, -1, (0, 0, 0, 0))

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

"Replace the reference methods by the HIP-backed ones (method overwrite)."
function enable!(libpath::AbstractString = LIB[])
    LIB[] = libpath
    @eval SignedDistances begin
        evalDistances(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64; kw...) =
            $(evalDistances_hip)(mesh, grid, points, ρₙ, ρₜ; kw...)
        Sign_Detection(mesh::Mesh, grid::Grid, points::Matrix, ρₙ::Vector{Float64}, ρₜ::Float64) =
            $(Sign_Detection_hip)(mesh, grid, points, ρₙ, ρₜ)
    end
    return nothing
end

end # module
