# VoronoiRT_hip.jl -- reference-side binding of libvrt_hip.so (include/voronoirt.h).
#
# `include` this file AFTER VoronoiRT.jl.  It redefines, inside the reference's module,
#   * both methods of `J_λ_voronoi` (src/lambda_iteration.jl:60-113, the line case, and
#     src/lambda_continuum.jl:27-56, the continuum case) so that the angle x wavelength loop the
#     reference threads over λ becomes ONE batched device solve -- the line case through
#     vrt_plan_execute_line: seven per-site vectors + S go in, J comes out, the Voigt profiles and
#     α_tot (nλ, n, n_angles) are made on the device and never exist on the host,
#   * `Λ_voronoi` (src/lambda_iteration.jl:205-300) over vrt_lambda_create / _iterate / _get: the loop's
#     state lives on the device, per iteration only the criterion's scalar comes back (plus the
#     populations and S_new the reference checkpoints); with ENV["VRT_DEVICES"] = "0,1,...,7" the same loop
#     runs across those GPUs (vrt_multi_create + vrt_multi_lambda_*: wavelength blocks per device, one RCCL
#     all-reduce of the rate-integral shares per iteration), and
#   * `Delaunay_upII` / `Delaunay_downII` (src/irregular_ray_tracing.jl:15-82, :96-163) for the
#     direct call sites in compare_searchlight.jl (:113,129,434),
#   * `short_characteristics_up/down` (src/characteristics.jl:19-95, :110-180).
# The physics that produces S, α and I_0 (γ, damping, Voigt profile, αline_λ, B_λ) stays in Julia
# exactly as the reference writes it; only the formal solves leave the process.
#
# Julia is not available in the build image: this file is shipped UNTESTED by execution.
# examples/c_caller.c plays exactly the callers written below -- same arrays, same call sequences:
# scenario 2 the batched J_λ_voronoi, scenario 3 the Λ_voronoi loop, scenario 4 that loop across several
# device handles -- and is checked against the
# oracle on the GPU (tests/test_gpu_parity.py).
#
# Unitful quantities are bit-identical to Float64 in memory, so `ustrip.(x)` gives the plain
# double* the C ABI takes.  Julia arrays are column-major and 1-based, exactly the conventions of
# the C ABI, so no transposition or index shift happens anywhere.
#
# `ccall` needs its (symbol, library) pair to be a constant expression, so every entry point gets
# its own literal-symbol wrapper below (no symbol is passed through a variable).

module VoronoiRTHip

using Unitful
using Libdl
import ..VoronoiRT
import ..VoronoiRT: VoronoiSites, HydrogenicLine, read_quadrature

const libvrt = get(ENV, "VRT_LIB", joinpath(@__DIR__, "..", "voronoirt_amd", "libvrt_hip.so"))

# entry points chosen at run time (one device or several: Λ below) are called through dlsym'd function pointers
const LIBVRT_HANDLE = Ref{Ptr{Cvoid}}(C_NULL)
function libvrt_handle()
    LIBVRT_HANDLE[] == C_NULL && (LIBVRT_HANDLE[] = dlopen(libvrt))
    return LIBVRT_HANDLE[]
end

vrt_error() = unsafe_string(ccall((:vrt_last_error, libvrt), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("libvrt_hip error $rc: $(vrt_error())")

const VRT_ALPHA_SITE = Cint(0)            # α[n]
const VRT_ALPHA_SITE_LAM = Cint(1)        # α[nλ, n]
const VRT_ALPHA_ANGLE_SITE_LAM = Cint(2)  # α[nλ, n, n_angles]
const VRT_ALPHA_ANGLE_NATIVE = Cint(3)    # per angle in the library's own layout (vrt_line_opacity_dev writes it)
const VRT_ALPHA_SITE_LAM_NATIVE = Cint(4) # α[nλ, n] in sweep order, both directions (vrt_plan_to_native_dev), with vrt_plan_execute_native_dev

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

# ---- several GPUs of the node from this one Julia process: ENV["VRT_DEVICES"] = "0,1,2,3,4,5,6,7" -------------
# (vrt_multi_*: a grid + plan per device and an in-process RCCL communicator; the Λ-iteration then runs as
# vrt_multi_lambda_*: wavelength blocks per device, ONE all-reduce of the rate-integral shares per iteration)
vrt_devices() = haskey(ENV, "VRT_DEVICES") && !isempty(ENV["VRT_DEVICES"]) ?
                Cint[parse(Cint, d) for d in split(ENV["VRT_DEVICES"], ",")] : Cint[]

const MULTIS = Dict{Tuple{UInt,String,Int},Ptr{Cvoid}}()

function multi_handle(sites::VoronoiSites, quadrature::String, n_sweeps::Int)
    get!(MULTIS, (objectid(sites), quadrature, n_sweeps)) do
        devices = vrt_devices()
        pos = Matrix{Float64}(ustrip.(u"m", sites.positions))
        bounds = Float64[ustrip(u"m", b) for b in (sites.z_min, sites.z_max, sites.x_min,
                                                   sites.x_max, sites.y_min, sites.y_max)]
        nbr = Matrix{Int64}(sites.neighbours)
        weights, θ, ϕ, n_angles = read_quadrature(quadrature)
        k = Matrix{Float64}(undef, 3, n_angles)
        for i in 1:n_angles
            k[:, i] = direction(θ[i], ϕ[i])
        end
        dirs = Cint[θ[i] > 90 ? 1 : (θ[i] < 90 ? -1 : 0) for i in 1:n_angles]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve devices pos bounds nbr k dirs begin
            check(ccall((:vrt_multi_create, libvrt), Cint,
                        (Cint, Ptr{Cint}, Int64, Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                         Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
                        length(devices), devices, sites.n, pos, nbr, size(nbr, 2), bounds, n_angles, k, dirs, n_sweeps, out))
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
# What stays in Julia: γ_constant of the current populations (:72-75) and damping_λ (:77-80; the
# caller, Λ_voronoi, hands it on to calculate_R).  What leaves: compute_voigt_profile per angle (:89),
# αline_λ + α_cont (:93-96) and the n_angles x nλ formal solves (:98-108) -- one call,
# vrt_plan_execute_line, with the per-site ingredients of α_tot instead of α_tot itself.
const I_unit = u"kW*m^-2*nm^-1"

# αline_λ for a unit profile (1 m^-1): the λ-independent factor h c_0/(4π λ0) (n_i B_ij - n_j B_ji) of
# src/line.jl:219-225, evaluated by the reference's own function so that no unit is guessed here
line_strength(line::HydrogenicLine, n_j, n_i) =
    Vector{Float64}(ustrip.(u"m^-1", VoronoiRT.αline_λ(line, fill(1.0u"m^-1", length(n_i)), n_j, n_i)))

function site_velocity(sites::VoronoiSites)               # (3, n) rows z, x, y: line_of_sight_velocity, line.jl:198-208
    v = Matrix{Float64}(undef, 3, sites.n)
    v[1, :] = ustrip.(u"m/s", sites.velocity_z); v[2, :] = ustrip.(u"m/s", sites.velocity_x)
    v[3, :] = ustrip.(u"m/s", sites.velocity_y)
    return v
end

function bottom_planck(sites::VoronoiSites, line::HydrogenicLine)
    # I_0 for up rays: B_λ(λ_l, T) of the bottom layer in perm_up order (:99-101); rows = λ
    bottom_layer = sites.layers_up[2] - 1
    bottom_layer_idx = sites.perm_up[1:bottom_layer]
    I0_up = Matrix{Float64}(undef, length(line.λ), bottom_layer)
    for l in eachindex(line.λ)
        I0_up[l, :] = ustrip.(I_unit, VoronoiRT.B_λ.(line.λ[l], sites.temperature[bottom_layer_idx]))
    end
    return I0_up
end

function J_line(S_λ, α_cont, populations, sites::VoronoiSites, line::HydrogenicLine, quadrature::String)
    weights, θ_array, ϕ_array, n_angles = read_quadrature(quadrature)
    nλ, n = size(S_λ)

    γ = VoronoiRT.γ_constant(line, sites.temperature,
                             (populations[:, 1] .+ populations[:, 2]), sites.electron_density)
    damping_λ = Matrix{Float64}(undef, size(S_λ))
    Threads.@threads for l in eachindex(line.λ)
        damping_λ[l, :] = VoronoiRT.damping.(γ, line.λ[l], line.ΔD)
    end

    λ = Vector{Float64}(ustrip.(u"m", line.λ))
    vel = site_velocity(sites)
    ΔD = Vector{Float64}(ustrip.(u"m", line.ΔD))
    γv = Vector{Float64}(ustrip.(u"s^-1", γ))
    strength = line_strength(line, populations[:, 2], populations[:, 1])
    αc = Vector{Float64}(ustrip.(u"m^-1", α_cont))
    S = Matrix{Float64}(ustrip.(I_unit, S_λ))
    I0_up = bottom_planck(sites, line)
    w = Vector{Float64}(weights)
    J = similar(S)
    plan = plan_handle(sites, quadrature, 3)            # n_sweeps = 3, :82
    GC.@preserve λ vel ΔD γv strength αc S I0_up w J begin
        check(ccall((:vrt_plan_execute_line, libvrt), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    plan, nλ, nλ, λ, ustrip(u"m", line.λ0), ustrip(u"m/s", VoronoiRT.c_0), vel, ΔD, γv, strength, αc,
                    S, I0_up, C_NULL, w, J))
    end
    return J * I_unit, damping_λ
end

# ---- Λ_voronoi: src/lambda_iteration.jl:205-300 ----------------------------------------------------
# mirrors `struct vrt_line_case` of include/voronoirt.h field for field (all 8-byte members)
struct LineCase
    nlam::Int64
    lambda::Ptr{Float64}
    blocks::NTuple{6,Int64}
    lambda0::Float64
    c0::Float64
    velocity::Ptr{Float64}
    doppler_width::Ptr{Float64}
    gamma_static::Ptr{Float64}
    gamma_unsold::Ptr{Float64}
    alpha_cont::Ptr{Float64}
    eps::Ptr{Float64}
    temperature::Ptr{Float64}
    atom_density::Ptr{Float64}
    B0::Ptr{Float64}
    lte_populations::Ptr{Float64}
    C::Ptr{Float64}
    planck2::Ptr{Float64}
    sigma_bf1::Ptr{Float64}
    sigma_bf2::Ptr{Float64}
    strength_const::Float64
    Bij::Float64
    Bji::Float64
    sigma_bb_const::Float64
    hc_over_kB::Float64
    pref_ij::Float64
    pref_ji::Float64
end

"""
    Λ(ϵ, maxiter, sites, line, quadrature, DATA) -> (J_new, S_new, α_cont, populations)

The reference's Λ_voronoi with its loop body on the device.  Everything Λ_voronoi derives BEFORE the loop
(LTE populations, α_cont, B_0, ε, C: :217-246) is computed by the reference's own functions; the loop
(:253-283) is vrt_lambda_iterate; after every iteration populations and S_new are fetched and written to the
HDF5 file exactly as the reference checkpoints them (:280-281); `criterion` keeps writing the history (:346).
"""
function Λ(ϵ::AbstractFloat, maxiter::Integer, sites::VoronoiSites, line::HydrogenicLine, quadrature::String, DATA::String)
    println("---Iterating---")
    LTE_pops = VoronoiRT.LTE_populations(line, sites)
    populations = copy(LTE_pops)
    α_cont = VoronoiRT.α_absorption.(line.λ0, sites.temperature, sites.electron_density * 1.0,
                                     LTE_pops[:, 1] .+ LTE_pops[:, 2], LTE_pops[:, 3]) .+
             VoronoiRT.α_scattering.(line.λ0, sites.electron_density, LTE_pops[:, 1])
    nλ, n = length(line.λ), sites.n
    B_0 = Matrix{Float64}(undef, nλ, n)
    for l in eachindex(line.λ)
        B_0[l, :] = ustrip.(I_unit, VoronoiRT.B_λ.(line.λ[l], sites.temperature))
    end
    ελ = VoronoiRT.destruction(LTE_pops, sites.electron_density, sites.temperature, line)
    println("Minimum $(minimum(ελ)) destruction probability")
    C = VoronoiRT.calculate_C(sites, LTE_pops)

    # γ_constant (src/broadening.jl:63-82) split into the part that follows the populations -- γ_unsold is linear in
    # the neutral-hydrogen density -- and the rest (natural width + the two Stark terms: n_e and T only)
    one_density = fill(1.0u"m^-3", n)
    γ_unsold_unit = VoronoiRT.γ_unsold.(VoronoiRT.const_unsold(line), sites.temperature, one_density)
    γ_static = VoronoiRT.γ_constant(line, sites.temperature, 0.0 .* one_density, sites.electron_density)

    λ = Vector{Float64}(ustrip.(u"m", line.λ))
    blocks = (Int64(line.λidx[1]), Int64(line.λidx[2]), Int64(line.λidx[2]), Int64(line.λidx[3]),
              Int64(line.λidx[3]), Int64(line.λidx[4]))                    # [lo, hi) 0-based: rates.jl:166-167,181-182
    hc = VoronoiRT.h * VoronoiRT.c_0
    # unit factors the reference gets from Unitful in Rij / Rji (rates.jl:226-364): J is a plain number in I_unit
    pref_ij = ustrip(u"s^-1", 2π / hc * 1u"m" * 1u"m^2" * 1I_unit * 1u"m") / 1000      # the explicit /1000 of :237,263
    pref_ji = ustrip(u"s^-1", 2π / hc * 1u"m" * 1u"m^2" * 1I_unit * 1u"m")
    planck2 = Vector{Float64}(ustrip.(I_unit, 2 * VoronoiRT.h * VoronoiRT.c_0^2 ./ line.λ .^ 5))
    σ1 = Vector{Float64}(ustrip.(u"m^2", VoronoiRT.σic(1, line, line.λ[line.λidx[2]+1:line.λidx[3]])))
    σ2 = Vector{Float64}(ustrip.(u"m^2", VoronoiRT.σic(2, line, line.λ[line.λidx[3]+1:line.λidx[4]])))
    vel = site_velocity(sites)
    ΔD = Vector{Float64}(ustrip.(u"m", line.ΔD))
    γs = Vector{Float64}(ustrip.(u"s^-1", γ_static)); γu = Vector{Float64}(ustrip.(u"s^-1", γ_unsold_unit))
    αc = Vector{Float64}(ustrip.(u"m^-1", α_cont)); εv = Vector{Float64}(ελ)
    T = Vector{Float64}(ustrip.(u"K", sites.temperature))
    atom = Vector{Float64}(ustrip.(u"m^-3", sites.hydrogen_populations))
    lte = Matrix{Float64}(ustrip.(u"m^-3", LTE_pops)); Cm = Array{Float64,3}(ustrip.(u"s^-1", C))
    weights, _, _, _ = read_quadrature(quadrature)
    w = Vector{Float64}(weights)
    # αline_λ = strength_const (n_1 B_ij - n_2 B_ji) φ: the two coefficients per unit density, from αline_λ itself
    a_i = line_strength(line, [0.0u"m^-3"], [1.0u"m^-3"])[1]
    a_j = -line_strength(line, [1.0u"m^-3"], [0.0u"m^-3"])[1]
    # one device: vrt_lambda_* on the plan; ENV["VRT_DEVICES"] lists several: vrt_multi_lambda_* (same arguments)
    multi = length(vrt_devices()) > 1
    plan = multi ? multi_handle(sites, quadrature, 3) : plan_handle(sites, quadrature, 3)
    f_create = multi ? :vrt_multi_lambda_create : :vrt_lambda_create
    f_iterate = multi ? :vrt_multi_lambda_iterate : :vrt_lambda_iterate
    f_get = multi ? :vrt_multi_lambda_get : :vrt_lambda_get
    f_destroy = multi ? :vrt_multi_lambda_destroy : :vrt_lambda_destroy
    ses = Ref{Ptr{Cvoid}}(C_NULL)
    J = Matrix{Float64}(undef, nλ, n); S = Matrix{Float64}(undef, nλ, n); pops = Matrix{Float64}(undef, n, 3)
    GC.@preserve λ vel ΔD γs γu αc εv T atom B_0 lte Cm planck2 σ1 σ2 w begin
        lc = Ref(LineCase(nλ, pointer(λ), blocks, ustrip(u"m", line.λ0), ustrip(u"m/s", VoronoiRT.c_0), pointer(vel),
                          pointer(ΔD), pointer(γs), pointer(γu), pointer(αc), pointer(εv), pointer(T), pointer(atom),
                          pointer(B_0), pointer(lte), pointer(Cm), pointer(planck2), pointer(σ1), pointer(σ2),
                          1.0, a_i, a_j,
                          ustrip(u"m^3", hc / (4 * π * line.λ0) * line.Bij),     # σ_constant of σij, rates.jl:398: x profile [1/m] = m^2
                          ustrip(u"m*K", hc / VoronoiRT.k_B), pref_ij, pref_ji))
        check(ccall(dlsym(libvrt_handle(), f_create), Cint, (Ptr{Cvoid}, Ref{LineCase}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    plan, lc, w, ses))
    end
    i = 0
    diff = Ref{Float64}(1.0)                  # criterion(S_new = B_0, S_old = 0) = |1 - 0/B| = 1, :325-349
    VoronoiRT.write_to_file(diff[], i + 1, DATA)
    while diff[] > ϵ && i < maxiter
        @time check(ccall(dlsym(libvrt_handle(), f_iterate), Cint, (Ptr{Cvoid}, Ref{Float64}), ses[], diff))
        isnan(diff[]) && println("NaN DIFF!")
        println("   Rel. diff.: $(diff[])")
        GC.@preserve S pops check(ccall(dlsym(libvrt_handle(), f_get), Cint,
                                        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                        ses[], C_NULL, S, pops, C_NULL, C_NULL))
        VoronoiRT.write_to_file(pops * 1u"m^-3", DATA)                    # the checkpoint, :280-281
        VoronoiRT.write_to_file(S * I_unit, DATA)
        i += 1
        VoronoiRT.write_to_file(diff[], i + 1, DATA)                      # convergence history, :346
    end
    GC.@preserve J S pops check(ccall(dlsym(libvrt_handle(), f_get), Cint,
                                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                      ses[], J, S, pops, C_NULL, C_NULL))
    ccall(dlsym(libvrt_handle(), f_destroy), Cvoid, (Ptr{Cvoid},), ses[])
    println(i == maxiter ? "Did not converge inside scope" : "Converged in $i iterations")
    return J * I_unit, S * I_unit, α_cont, pops * 1u"m^-3"
end

# ---- J_λ_voronoi, continuum case: src/lambda_continuum.jl:27-56 ---------------------------------
# one wavelength (500 nm), α independent of the angle; S_λ and α_cont are n-vectors there
function J_continuum(S_λ::AbstractVector, α_cont::AbstractVector, sites::VoronoiSites, quadrature::String)
    weights, θ_array, ϕ_array, n_points = read_quadrature(quadrature)
    bottom_layer = sites.layers_up[2] - 1
    bottom_layer_idx = sites.perm_up[1:bottom_layer]
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
# the Λ-iteration driver of compare_line.jl:125 (src/lambda_iteration.jl:205-300): loop body on the device
VoronoiRT.Λ_voronoi(ϵ::AbstractFloat, maxiter::Integer, sites::VoronoiRT.VoronoiSites, line::VoronoiRT.HydrogenicLine,
                    quadrature::String, DATA::String) =
    VoronoiRTHip.Λ(ϵ, maxiter, sites, line, quadrature, DATA)
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
