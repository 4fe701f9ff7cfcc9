# VoronoiRT_hip.jl -- reference-side binding of libvrt_hip.so (include/voronoirt.h).
#
# `include` this file AFTER VoronoiRT.jl: it redefines the formal-solver methods of the reference
# (src/irregular_ray_tracing.jl:15-82, :96-163 and the body of J_λ_voronoi,
# src/lambda_iteration.jl:60-113 / src/lambda_continuum.jl:27-56) so that they ccall the MI355X
# library.  Julia is not available in the build image: this file is shipped UNTESTED by
# execution; tests/ exercise the same C entry points through the Python mirror.
#
# Unitful quantities are bit-identical to Float64 in memory, so `ustrip.(x)` / reinterpret gives
# the plain double* the C ABI takes.  Julia arrays are column-major and 1-based, exactly the
# conventions of the C ABI, so no transposition or index shift happens anywhere.

module VoronoiRTHip

using Unitful
import ..VoronoiRT: VoronoiSites, read_quadrature

const libvrt = get(ENV, "VRT_LIB", joinpath(@__DIR__, "..", "voronoirt_amd", "libvrt_hip.so"))

vrt_error() = unsafe_string(ccall((:vrt_last_error, libvrt), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("libvrt_hip error $rc: $(vrt_error())")

# one device-resident grid handle per VoronoiSites object
const GRIDS = IdDict{Any,Ptr{Cvoid}}()

function grid_handle(sites::VoronoiSites; device::Integer=0)
    get!(GRIDS, sites) do
        pos = ustrip.(u"m", sites.positions)                     # (3, n) z,x,y
        bounds = Float64[ustrip(u"m", b) for b in (sites.z_min, sites.z_max, sites.x_min,
                                                   sites.x_max, sites.y_min, sites.y_max)]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:vrt_grid_create, libvrt), Cint,
                    (Int64, Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
                    sites.n, pos, sites.neighbours, size(sites.neighbours, 2), bounds, device, out))
        out[]
    end
end

function solve(sym::Symbol, k, S, I_0, α, sites::VoronoiSites, n_sweeps::Int)
    g = grid_handle(sites)
    Sv = Vector{Float64}(ustrip.(S)); I0 = Vector{Float64}(ustrip.(I_0)); αv = Vector{Float64}(ustrip.(α))
    I = similar(Sv)
    GC.@preserve Sv I0 αv I begin
        check(ccall((sym, libvrt), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Cint, Ptr{Float64}),
                    g, Vector{Float64}(k), Sv, I0, length(I0), αv, n_sweeps, I))
    end
    return I * unit(eltype(S))
end

# batched replacement of the angle x wavelength loop (lambda_iteration.jl:84-111)
const PLANS = Dict{Tuple{Ptr{Cvoid},String,Int},Ptr{Cvoid}}()

function J_voronoi(S_λ::AbstractMatrix, α_tot::AbstractArray, I0_up::AbstractMatrix,
                   sites::VoronoiSites, quadrature::String; n_sweeps::Int=3)
    weights, θ, ϕ, n_angles = read_quadrature(quadrature)
    g = grid_handle(sites)
    plan = get!(PLANS, (g, quadrature, n_sweeps)) do
        k = Matrix{Float64}(undef, 3, n_angles)
        for i in 1:n_angles
            k[:, i] = [cos(θ[i]*π/180), cos(ϕ[i]*π/180)*sin(θ[i]*π/180), sin(ϕ[i]*π/180)*sin(θ[i]*π/180)]
        end
        dirs = Cint[θ[i] > 90 ? 1 : (θ[i] < 90 ? -1 : 0) for i in 1:n_angles]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:vrt_plan_create_ex, libvrt), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
                    g, n_angles, k, dirs, n_sweeps, out))
        out[]
    end
    nλ, n = size(S_λ)
    S = Matrix{Float64}(ustrip.(S_λ)); α = Array{Float64}(ustrip.(α_tot)); I0 = Matrix{Float64}(ustrip.(I0_up))
    mode = ndims(α) == 1 ? 0 : (ndims(α) == 2 ? 1 : 2)      # VRT_ALPHA_*; 3-D α is (nλ, n, n_angles)
    J = similar(S)
    GC.@preserve S α I0 J begin
        check(ccall((:vrt_plan_execute, libvrt), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    plan, nλ, nλ, S, α, mode, I0, C_NULL, weights, J, C_NULL))
    end
    return J * unit(eltype(S_λ))
end

# regular-grid short characteristics (src/characteristics.jl:19-95, :110-180): S_0, α are
# (nz, nx, ny) Julia arrays, I_0 is (nx, ny); `atmos` contributes its three axes only
function regular_solve(up::Bool, k, S_0::AbstractArray{<:Any,3}, I_0::AbstractMatrix, α::AbstractArray{<:Any,3},
                       atmos; n_sweeps::Int=3, device::Integer=0)
    z = Vector{Float64}(ustrip.(u"m", atmos.z)); x = Vector{Float64}(ustrip.(u"m", atmos.x))
    y = Vector{Float64}(ustrip.(u"m", atmos.y))
    S = Array{Float64,3}(ustrip.(S_0)); A = Array{Float64,3}(ustrip.(α)); I0 = Matrix{Float64}(ustrip.(I_0))
    I = similar(S)
    kk = Vector{Float64}(k); upv = Cint[up ? 1 : 0]
    GC.@preserve z x y S A I0 I kk upv begin
        check(ccall((:vrt_short_characteristics, libvrt), Cint,
                    (Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cint},
                     Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Cint, Cint, Ptr{Float64}),
                    length(z), length(x), length(y), z, x, y, 1, kk, upv, S, 0, A, 0, I0, n_sweeps, device, I))
    end
    return I * unit(eltype(S_0))
end

end # module

# ---- drop-in redefinitions ----------------------------------------------------------------------
VoronoiRT.Delaunay_upII(k::Vector{Float64}, S, I_0, α, sites::VoronoiRT.VoronoiSites, n_sweeps::Int) =
    VoronoiRTHip.solve(:vrt_delaunay_up, k, S, I_0, α, sites, n_sweeps)
VoronoiRT.Delaunay_downII(k::Vector{Float64}, S, I_0, α, sites::VoronoiRT.VoronoiSites, n_sweeps::Int) =
    VoronoiRTHip.solve(:vrt_delaunay_down, k, S, I_0, α, sites, n_sweeps)
VoronoiRT.short_characteristics_up(k::Vector{Float64}, S_0, I_0, α, atmos::VoronoiRT.Atmosphere;
                                   pt::Bool=false, n_sweeps::Int=3) =
    VoronoiRTHip.regular_solve(true, k, S_0, I_0, α, atmos; n_sweeps=n_sweeps)
VoronoiRT.short_characteristics_down(k::Vector{Float64}, S_0, I_0, α, atmos::VoronoiRT.Atmosphere;
                                     pt::Bool=false, n_sweeps::Int=3) =
    VoronoiRTHip.regular_solve(false, k, S_0, I_0, α, atmos; n_sweeps=n_sweeps)
