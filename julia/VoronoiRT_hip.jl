# VoronoiRT_hip.jl -- reference-side binding of libvrt_hip.so (include/voronoirt.h).
#
# `include` this file AFTER VoronoiRT.jl.  It redefines, inside the reference's module,
#   * both methods of `J_λ_voronoi` (src/lambda_iteration.jl:60-113, the line case, and
#     src/lambda_continuum.jl:27-56, the continuum case) so that the angle x wavelength loop the
#     reference threads over λ becomes ONE batched device solve (vrt_plan_execute), and
#   * `Delaunay_upII` / `Delaunay_downII` (src/irregular_ray_tracing.jl:15-82, :96-163) for the
#     direct call sites in compare_searchlight.jl (:113,129,434),
#   * `short_characteristics_up/down` (src/characteristics.jl:19-95, :110-180).
# The physics that produces S, α and I_0 (γ, damping, Voigt profile, αline_λ, B_λ) stays in Julia
# exactly as the reference writes it; only the formal solves leave the process.
#
# Julia is not available in the build image: this file is shipped UNTESTED by execution.
# examples/c_caller.c (scenario 2) plays exactly the caller written below -- same arrays, same
# call sequence -- and is checked against the oracle on the GPU (tests/test_gpu_parity.py).
#
# Unitful quantities are bit-identical to Float64 in memory, so `ustrip.(x)` gives the plain
# double* the C ABI takes.  Julia arrays are column-major and 1-based, exactly the conventions of
# the C ABI, so no transposition or index shift happens anywhere.
#
# `ccall` needs its (symbol, library) pair to be a constant expression, so every entry point gets
# its own literal-symbol wrapper below (no symbol is passed through a variable).

module VoronoiRTHip

using Unitful
import ..VoronoiRT
import ..VoronoiRT: VoronoiSites, HydrogenicLine, read_quadrature

const libvrt = get(ENV, "VRT_LIB", joinpath(@__DIR__, "..", "voronoirt_amd", "libvrt_hip.so"))

vrt_error() = unsafe_string(ccall((:vrt_last_error, libvrt), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("libvrt_hip error $rc: $(vrt_error())")

const VRT_ALPHA_SITE = Cint(0)            # α[n]
const VRT_ALPHA_SITE_LAM = Cint(1)        # α[nλ, n]
const VRT_ALPHA_ANGLE_SITE_LAM = Cint(2)  # α[nλ, n, n_angles]

# ---- one device-resident grid handle per VoronoiSites object -----------------------------------
const GRIDS = IdDict{Any,Ptr{Cvoid}}()

function grid_handle(sites::VoronoiSites; device::Integer=0)
    get!(GRIDS, sites) do
        pos = Matrix{Float64}(ustrip.(u"m", sites.positions))    # (3, n) z,x,y
        bounds = Float64[ustrip(u"m", b) for b in (sites.z_min, sites.z_max, sites.x_min,
                                                   sites.x_max, sites.y_min, sites.y_max)]
        nbr = Matrix{Int64}(sites.neighbours)                    # (n, D+1), column 1 = count
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve pos bounds nbr begin
            check(ccall((:vrt_grid_create, libvrt), Cint,
                        (Int64, Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
                        sites.n, pos, nbr, size(nbr, 2), bounds, device, out))
        end
        out[]
    end
end

# ---- one plan (upwind tables + sweep schedule) per (grid, quadrature, n_sweeps) -----------------
const PLANS = Dict{Tuple{Ptr{Cvoid},String,Int},Ptr{Cvoid}}()

direction(θ, ϕ) = [cos(θ*π/180), cos(ϕ*π/180)*sin(θ*π/180), sin(ϕ*π/180)*sin(θ*π/180)]   # lambda_iteration.jl:87

function plan_handle(sites::VoronoiSites, quadrature::String, n_sweeps::Int)
    g = grid_handle(sites)
    get!(PLANS, (g, quadrature, n_sweeps)) do
        weights, θ, ϕ, n_angles = read_quadrature(quadrature)
        k = Matrix{Float64}(undef, 3, n_angles)
        for i in 1:n_angles
            k[:, i] = direction(θ[i], ϕ[i])
        end
        # the reference branches on θ in degrees (lambda_iteration.jl:98,104), not on sign(k_z)
        dirs = Cint[θ[i] > 90 ? 1 : (θ[i] < 90 ? -1 : 0) for i in 1:n_angles]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve k dirs begin
            check(ccall((:vrt_plan_create_ex, libvrt), Cint,
                        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
                        g, n_angles, k, dirs, n_sweeps, out))
        end
        out[]
    end
end

"""
    execute(plan, S, α, mode, I0_up, weights) -> J

One call = every angle x every wavelength of the quadrature (vrt_plan_execute).  `S` is (nλ, n),
`α` is `[n]`, `(nλ, n)` or `(nλ, n, n_angles)` according to `mode`, `I0_up` is
(nλ, layers_up[2]-1) ordered like perm_up (lambda_iteration.jl:99-101); down rays start from zeros
(:105-106), which is what a NULL I0_down means.
"""
function execute(plan::Ptr{Cvoid}, S::Matrix{Float64}, α::Array{Float64}, mode::Cint,
                 I0_up::Matrix{Float64}, weights::Vector{Float64})
    nλ, n = size(S)
    J = similar(S)
    GC.@preserve S α I0_up weights J begin
        check(ccall((:vrt_plan_execute, libvrt), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    plan, nλ, nλ, S, α, mode, I0_up, C_NULL, weights, J, C_NULL))
    end
    return J
end

# ---- J_λ_voronoi, line case: src/lambda_iteration.jl:60-113 -------------------------------------
# γ, damping_λ, the Voigt profile per angle and αline_λ are computed by the reference's own
# functions (:72-80, :89, :93-96); what changes is that α_tot is collected for every angle into
# one (nλ, n, n_angles) array and the n_angles x nλ formal solves run as one batched call.
function J_line(S_λ, α_cont, populations, sites::VoronoiSites, line::HydrogenicLine, quadrature::String)
    weights, θ_array, ϕ_array, n_angles = read_quadrature(quadrature)
    nλ, n = size(S_λ)

    γ = VoronoiRT.γ_constant(line, sites.temperature,
                             (populations[:, 1] .+ populations[:, 2]), sites.electron_density)
    damping_λ = Matrix{Float64}(undef, size(S_λ))
    Threads.@threads for l in eachindex(line.λ)
        damping_λ[l, :] = VoronoiRT.damping.(γ, line.λ[l], line.ΔD)
    end

    α_tot = Array{Float64,3}(undef, nλ, n, n_angles)
    αc = ustrip.(u"m^-1", α_cont)
    for i in 1:n_angles
        k = direction(θ_array[i], ϕ_array[i])
        profile = VoronoiRT.compute_voigt_profile(line, sites, damping_λ, k)
        Threads.@threads for l in eachindex(line.λ)
            αl = VoronoiRT.αline_λ(line, profile[l, :], populations[:, 2], populations[:, 1])
            α_tot[l, :, i] = ustrip.(u"m^-1", αl) .+ αc
        end
    end

    # I_0 for up rays: B_λ(λ_l, T) of the bottom layer in perm_up order (:99-101); rows = λ
    bottom_layer = sites.layers_up[2] - 1
    bottom_layer_idx = sites.perm_up[1:bottom_layer]
    I_unit = unit(eltype(S_λ))
    I0_up = Matrix{Float64}(undef, nλ, bottom_layer)
    for l in eachindex(line.λ)
        I0_up[l, :] = ustrip.(I_unit, VoronoiRT.B_λ.(line.λ[l], sites.temperature[bottom_layer_idx]))
    end

    plan = plan_handle(sites, quadrature, 3)            # n_sweeps = 3, :82
    J = execute(plan, Matrix{Float64}(ustrip.(I_unit, S_λ)), α_tot, VRT_ALPHA_ANGLE_SITE_LAM,
                I0_up, Vector{Float64}(weights))
    return J * I_unit, damping_λ
end

# ---- J_λ_voronoi, continuum case: src/lambda_continuum.jl:27-56 ---------------------------------
# one wavelength (500 nm), α independent of the angle; S_λ and α_cont are n-vectors there
function J_continuum(S_λ::AbstractVector, α_cont::AbstractVector, sites::VoronoiSites, quadrature::String)
    weights, θ_array, ϕ_array, n_points = read_quadrature(quadrature)
    bottom_layer = sites.layers_up[2] - 1
    bottom_layer_idx = sites.perm_up[1:bottom_layer]
    I_unit = unit(eltype(S_λ))
    I0_up = reshape(Vector{Float64}(ustrip.(I_unit,
                VoronoiRT.blackbody_λ.(500u"nm", sites.temperature[bottom_layer_idx]))), 1, bottom_layer)
    plan = plan_handle(sites, quadrature, 3)
    S = reshape(Vector{Float64}(ustrip.(I_unit, S_λ)), 1, length(S_λ))
    α = Vector{Float64}(ustrip.(u"m^-1", α_cont))
    J = execute(plan, S, α, VRT_ALPHA_SITE, I0_up, Vector{Float64}(weights))
    return vec(J) * I_unit
end

# ---- single solves: src/irregular_ray_tracing.jl:15-20, :96-101 ----------------------------------
# literal-symbol wrappers (the name/library tuple of a ccall must not reference a local variable)
c_delaunay_up(g, k, S, I0, α, n_sweeps, I) =
    ccall((:vrt_delaunay_up, libvrt), Cint,
          (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Cint, Ptr{Float64}),
          g, k, S, I0, length(I0), α, n_sweeps, I)
c_delaunay_down(g, k, S, I0, α, n_sweeps, I) =
    ccall((:vrt_delaunay_down, libvrt), Cint,
          (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Cint, Ptr{Float64}),
          g, k, S, I0, length(I0), α, n_sweeps, I)

function solve(up::Bool, k, S, I_0, α, sites::VoronoiSites, n_sweeps::Int)
    g = grid_handle(sites)
    I_unit = unit(eltype(S))
    kv = Vector{Float64}(k)
    Sv = Vector{Float64}(ustrip.(I_unit, S)); I0 = Vector{Float64}(ustrip.(I_unit, I_0))
    αv = Vector{Float64}(ustrip.(u"m^-1", α))
    I = similar(Sv)
    GC.@preserve kv Sv I0 αv I begin
        check(up ? c_delaunay_up(g, kv, Sv, I0, αv, n_sweeps, I) : c_delaunay_down(g, kv, Sv, I0, αv, n_sweeps, I))
    end
    return I * I_unit
end

# ---- voro (src/functions.jl:13-23): the voro++ fork/exec, in-process --------------------------------
# Reads the sites file the driver has just written (write_arrays, src/io.jl:16-20: "id\tx\ty\tz"),
# tessellates with vrt_tessellate and writes the "%i %n" neighbours file read_cell parses, so the
# driver's own `voro(...); read_cell(...)` sequence (compare_line.jl:100-103) runs unchanged.
function voro_inprocess(sites_file::String, neighbours_file::String,
                        x_min::Float64, x_max::Float64, y_min::Float64, y_max::Float64,
                        z_min::Float64, z_max::Float64)
    rows = [split(l) for l in eachline(sites_file) if !isempty(strip(l))]
    n = length(rows)
    pos = Matrix{Float64}(undef, 3, n)                      # (3, n) rows z, x, y
    for r in rows
        i = parse(Int, r[1])
        pos[2, i] = parse(Float64, r[2]); pos[3, i] = parse(Float64, r[3]); pos[1, i] = parse(Float64, r[4])
    end
    bounds = Float64[z_min, z_max, x_min, x_max, y_min, y_max]
    D1 = 71                                                  # max_guess + 1, voronoi_utils.jl:42
    nbr = zeros(Int64, n, D1)
    mx = Ref{Int64}(0)
    GC.@preserve pos bounds nbr begin
        check(ccall((:vrt_tessellate, libvrt), Cint,
                    (Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Ref{Int64}),
                    n, pos, bounds, D1, nbr, mx))
        check(ccall((:vrt_write_neighbours_file, libvrt), Cint, (Cstring, Int64, Ptr{Int64}, Int64),
                    neighbours_file, n, nbr, D1))
    end
    return nothing
end

# ---- regular-grid short characteristics (src/characteristics.jl:19-95, :110-180) ----------------
# S_0, α are (nz, nx, ny) Julia arrays, I_0 is (nx, ny); `atmos` contributes its three axes only
function regular_solve(up::Bool, k, S_0::AbstractArray{<:Any,3}, I_0::AbstractMatrix, α::AbstractArray{<:Any,3},
                       atmos; n_sweeps::Int=3, device::Integer=0)
    z = Vector{Float64}(ustrip.(u"m", atmos.z)); x = Vector{Float64}(ustrip.(u"m", atmos.x))
    y = Vector{Float64}(ustrip.(u"m", atmos.y))
    I_unit = unit(eltype(S_0))
    S = Array{Float64,3}(ustrip.(I_unit, S_0)); A = Array{Float64,3}(ustrip.(u"m^-1", α))
    I0 = Matrix{Float64}(ustrip.(I_unit, I_0))
    I = similar(S)
    kk = Vector{Float64}(k); upv = Cint[up ? 1 : 0]
    GC.@preserve z x y S A I0 I kk upv begin
        check(ccall((:vrt_short_characteristics, libvrt), Cint,
                    (Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cint},
                     Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Cint, Cint, Ptr{Float64}),
                    length(z), length(x), length(y), z, x, y, 1, kk, upv, S, 0, A, 0, I0, n_sweeps, device, I))
    end
    return I * I_unit
end

end # module

# ---- drop-in redefinitions (same signatures as the reference's methods) --------------------------
# the batched path: what compare_line.jl / Λ_voronoi (lambda_iteration.jl:259) and the continuum
# driver (lambda_continuum.jl) call
VoronoiRT.J_λ_voronoi(S_λ::Matrix{<:VoronoiRT.UnitsIntensity_λ}, α_cont::Vector{<:VoronoiRT.PerLength},
                      populations::Matrix{<:VoronoiRT.NumberDensity}, sites::VoronoiRT.VoronoiSites,
                      line::VoronoiRT.HydrogenicLine, quadrature::String) =
    VoronoiRTHip.J_line(S_λ, α_cont, populations, sites, line, quadrature)
VoronoiRT.J_λ_voronoi(S_λ::AbstractArray, α_cont::AbstractArray, sites::VoronoiRT.VoronoiSites,
                      quadrature::String) =
    VoronoiRTHip.J_continuum(S_λ, α_cont, sites, quadrature)
# preprocessing (compare_line.jl:100, compare_continuum.jl, compare_searchlight.jl): the executable's
# path is ignored, the tessellation runs inside the library
VoronoiRT.voro(voro_executable::String, sites_file::String, neighbours_file::String,
               x_min::Float64, x_max::Float64, y_min::Float64, y_max::Float64,
               z_min::Float64, z_max::Float64) =
    VoronoiRTHip.voro_inprocess(sites_file, neighbours_file, x_min, x_max, y_min, y_max, z_min, z_max)
# single solves (compare_searchlight.jl:113,129,434)
VoronoiRT.Delaunay_upII(k::Vector{Float64}, S, I_0, α, sites::VoronoiRT.VoronoiSites, n_sweeps::Int) =
    VoronoiRTHip.solve(true, k, S, I_0, α, sites, n_sweeps)
VoronoiRT.Delaunay_downII(k::Vector{Float64}, S, I_0, α, sites::VoronoiRT.VoronoiSites, n_sweeps::Int) =
    VoronoiRTHip.solve(false, k, S, I_0, α, sites, n_sweeps)
VoronoiRT.short_characteristics_up(k::Vector{Float64}, S_0, I_0, α, atmos::VoronoiRT.Atmosphere;
                                   pt::Bool=false, n_sweeps::Int=3) =
    VoronoiRTHip.regular_solve(true, k, S_0, I_0, α, atmos; n_sweeps=n_sweeps)
VoronoiRT.short_characteristics_down(k::Vector{Float64}, S_0, I_0, α, atmos::VoronoiRT.Atmosphere;
                                     pt::Bool=false, n_sweeps::Int=3) =
    VoronoiRTHip.regular_solve(false, k, S_0, I_0, α, atmos; n_sweeps=n_sweeps)
