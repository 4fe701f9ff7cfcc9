// Host-pointer entry points of the line case: what a single-process host WITHOUT device arrays of its own
// (the reference's Julia driver) calls instead of shipping α_tot (nλ, n, n_angles) through PCIe.
//   vrt_line_terms_dev     γ and the line strength from the current populations (device pointers)
//   vrt_plan_execute_line  the body of J_λ_voronoi, line case (src/lambda_iteration.jl:72-111): per-site line
//                          vectors + S in, J out; α_tot is made on the device and never exists on the host
//   vrt_lambda_*           Λ_voronoi's loop (src/lambda_iteration.jl:205-300) with library-owned device state:
//                          per iteration only the criterion's scalar comes back
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "vrt_internal.h"

using namespace vrt;

namespace {

template <typename T>
int dalloc(T **p, size_t count)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    return VRT_OK;
}

int upload(double **d, const double *h, size_t count, hipStream_t st)
{
    int rc = dalloc(d, count);
    if (rc) return rc;
    VRT_HIP_TRY(hipMemcpyAsync(*d, h, sizeof(double) * count, hipMemcpyHostToDevice, st));
    return VRT_OK;
}

}  // namespace

struct vrt_lambda {
    vrt_plan *p = nullptr;
    int device = 0;                     // of the plan's grid (kept here: destroying the session must not look into a plan that may be gone)
    int64_t n = 0, nlam = 0;
    int64_t blocks[6] = {0, 0, 0, 0, 0, 0};
    double lambda0 = 0, c0 = 0, strength_const = 0, Bij = 0, Bji = 0, sigma_bb_const = 0, hc_over_kB = 0, pref_ij = 0,
           pref_ji = 0;
    std::vector<double> weights;
    // device state
    double *d_small = nullptr;          // lambda | planck2 | sigma_bf1 | sigma_bf2
    double *d_velocity = nullptr, *d_doppler = nullptr, *d_gamma_static = nullptr, *d_gamma_unsold = nullptr,
           *d_alpha_cont = nullptr, *d_eps = nullptr, *d_temperature = nullptr, *d_atom = nullptr, *d_B0 = nullptr,
           *d_lte = nullptr, *d_C = nullptr;
    double *d_gamma = nullptr, *d_strength = nullptr, *d_pops = nullptr, *d_pops_new = nullptr, *d_R = nullptr;
    double *d_S_old = nullptr, *d_S_new = nullptr, *d_J = nullptr, *d_I0 = nullptr, *d_native = nullptr;
    unsigned long long *d_scalars = nullptr;
    int iterations = 0;
};

static void lambda_free(vrt_lambda *s)
{
    if (!s) return;
    for (double *q : {s->d_small, s->d_velocity, s->d_doppler, s->d_gamma_static, s->d_gamma_unsold, s->d_alpha_cont,
                      s->d_eps, s->d_temperature, s->d_atom, s->d_B0, s->d_lte, s->d_C, s->d_gamma, s->d_strength,
                      s->d_pops, s->d_pops_new, s->d_R, s->d_S_old, s->d_S_new, s->d_J, s->d_I0, s->d_native})
        if (q) (void)hipFree(q);
    if (s->d_scalars) (void)hipFree(s->d_scalars);
    delete s;
}

extern "C" {

int vrt_line_terms_dev(vrt_grid *g, const double *d_gamma_static, const double *d_gamma_unsold,
                       const double *d_populations, double strength_const, double Bij, double Bji, double *d_gamma,
                       double *d_line_strength, void *stream)
{
    DeviceScope scope;
    if (!g || !d_populations) return fail(VRT_EINVAL, "NULL argument");
    if (d_gamma && (!d_gamma_static || !d_gamma_unsold)) return fail(VRT_EINVAL, "gamma needs gamma_static and gamma_unsold");
    int rc = use_device(g->device);
    if (rc) return rc;
    return launch_line_terms(g->n, d_gamma_static, d_gamma_unsold, d_populations, strength_const, Bij, Bji, d_gamma,
                             d_line_strength, (hipStream_t)stream);
}

int vrt_plan_execute_line(vrt_plan *p, int64_t nlam, int64_t ld, const double *lambda, double lambda0, double c0,
                          const double *velocity, const double *doppler_width, const double *gamma,
                          const double *line_strength, const double *alpha_cont, const double *S, const double *I0_up,
                          const double *I0_down, const double *weights, double *J)
{
    DeviceScope scope;
    if (!p || !lambda || !velocity || !doppler_width || !gamma || !line_strength || !alpha_cont || !S || !weights || !J)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (!(lambda0 > 0) || !(c0 > 0)) return fail(VRT_EINVAL, "lambda0 and c0 must be positive");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        vrt_grid *g = p->g;
        int rc = use_device(g->device);
        if (rc) return rc;
        if (!p->patch_ok && (!p->tile_ok || p->tile_max_layer_size > steps_max_layer(false)))
            return fail(VRT_EINVAL, "the line entry needs a layer path (at most 4 visits per site and 255 levels per layer)");
        if (p->A != (int)p->n_angles_user)
            return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
        const size_t n = (size_t)g->n, nS = n * (size_t)ld;
        const size_t nU = (size_t)g->up.n1 * (size_t)nlam, nD = (size_t)g->down.n1 * (size_t)nlam;
        hipStream_t st = g->stream;
        // staging: S | J in the plan's stage buffers; the nine per-site vectors + λ in stage 1; α_tot native in ws_AA
        auto ensure = [&](double *&buf, size_t &cap, size_t count) -> int {
            if (buf && count <= cap) return VRT_OK;
            if (buf) (void)hipFree(buf);
            buf = nullptr;
            cap = 0;
            int r = dalloc(&buf, count);
            if (!r) cap = count;
            return r;
        };
        const size_t vecs = 7 * n + (size_t)nlam;        // velocity (3n), ΔλD, γ, strength, α_cont, λ
        if ((rc = ensure(p->d_stage[0], p->stage_cap[0], nS))) return rc;
        if ((rc = ensure(p->d_stage[1], p->stage_cap[1], vecs))) return rc;
        if ((rc = ensure(p->d_stage[4], p->stage_cap[4], nS))) return rc;
        const size_t nnat = (size_t)vrt_plan_native_alpha_count(p, nlam);
        if ((rc = ensure(p->ws_AA, p->ws_AA_cap, nnat))) return rc;
        double *dv = p->d_stage[1];
        double *d_vel = dv, *d_dop = dv + 3 * n, *d_gam = dv + 4 * n, *d_str = dv + 5 * n, *d_ac = dv + 6 * n, *d_lam = dv + 7 * n;
        VRT_HIP_TRY(hipMemcpyAsync(p->d_stage[0], S, sizeof(double) * nS, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_vel, velocity, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_dop, doppler_width, sizeof(double) * n, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_gam, gamma, sizeof(double) * n, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_str, line_strength, sizeof(double) * n, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_ac, alpha_cont, sizeof(double) * n, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(d_lam, lambda, sizeof(double) * (size_t)nlam, hipMemcpyHostToDevice, st));
        double *dU = nullptr, *dD = nullptr;
        if (I0_up && nU) {
            if ((rc = ensure(p->d_stage[2], p->stage_cap[2], nU))) return rc;
            dU = p->d_stage[2];
            VRT_HIP_TRY(hipMemcpyAsync(dU, I0_up, sizeof(double) * nU, hipMemcpyHostToDevice, st));
        }
        if (I0_down && nD) {
            if ((rc = ensure(p->d_stage[3], p->stage_cap[3], nD))) return rc;
            dD = p->d_stage[3];
            VRT_HIP_TRY(hipMemcpyAsync(dD, I0_down, sizeof(double) * nD, hipMemcpyHostToDevice, st));
        }
        // α_tot of every angle straight into the native layout (lambda_iteration.jl:72-80, :89, :93-96), then the sweep
        if ((rc = launch_line_opacity(p, nlam, d_lam, lambda0, c0, d_vel, d_dop, d_gam, d_str, d_ac, p->ws_AA, st))) return rc;
        // (execute_dev_locked reuses ws_AA only for the CALLER-layout per-angle alpha, not for the native one)
        rc = execute_dev_locked(p, nlam, ld, p->d_stage[0], p->ws_AA, VRT_ALPHA_ANGLE_NATIVE, dU, dD, weights,
                                p->d_stage[4], nullptr, st);
        if (rc) return rc;
        VRT_HIP_TRY(hipMemcpyAsync(J, p->d_stage[4], sizeof(double) * nS, hipMemcpyDeviceToHost, st));
        VRT_HIP_TRY(hipStreamSynchronize(st));
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_lambda_create(vrt_plan *p, const vrt_line_case *lc, const double *weights, vrt_lambda **out)
{
    DeviceScope scope;
    if (!out) return fail(VRT_EINVAL, "out is NULL");
    *out = nullptr;
    if (!p || !lc || !weights) return fail(VRT_EINVAL, "NULL argument");
    const int64_t nlam = lc->nlam;
    if (nlam < 2) return fail(VRT_EINVAL, "nlam must be >= 2");
    if (!lc->lambda || !lc->velocity || !lc->doppler_width || !lc->gamma_static || !lc->gamma_unsold || !lc->alpha_cont ||
        !lc->eps || !lc->temperature || !lc->atom_density || !lc->B0 || !lc->lte_populations || !lc->C || !lc->planck2 ||
        !lc->sigma_bf1 || !lc->sigma_bf2)
        return fail(VRT_EINVAL, "NULL array in the line case");
    for (int b = 0; b < 3; b++)
        if (lc->blocks[2 * b] < 0 || lc->blocks[2 * b + 1] > nlam || lc->blocks[2 * b + 1] - lc->blocks[2 * b] < 2)
            return fail(VRT_EINVAL, "each wavelength block needs at least two wavelengths inside [0, nlam)");
    if (!(lc->lambda0 > 0) || !(lc->c0 > 0)) return fail(VRT_EINVAL, "lambda0 and c0 must be positive");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        vrt_grid *g = p->g;
        int rc = use_device(g->device);
        if (rc) return rc;
        if (!p->patch_ok && (!p->tile_ok || p->tile_max_layer_size > steps_max_layer(false)))
            return fail(VRT_EINVAL, "the line session needs a layer path (at most 4 visits per site and 255 levels per layer)");
        if (p->A != (int)p->n_angles_user)
            return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
        vrt_lambda *s = new (std::nothrow) vrt_lambda();
        if (!s) return fail(VRT_ENOMEM, "out of host memory");
        s->p = p;
        s->device = g->device;
        s->n = g->n;
        s->nlam = nlam;
        for (int q = 0; q < 6; q++) s->blocks[q] = lc->blocks[q];
        s->lambda0 = lc->lambda0; s->c0 = lc->c0; s->strength_const = lc->strength_const; s->Bij = lc->Bij; s->Bji = lc->Bji;
        s->sigma_bb_const = lc->sigma_bb_const; s->hc_over_kB = lc->hc_over_kB; s->pref_ij = lc->pref_ij; s->pref_ji = lc->pref_ji;
        s->weights.assign(weights, weights + p->n_angles_user);
        const size_t n = (size_t)g->n, nl = (size_t)nlam;
        hipStream_t st = g->stream;
        const size_t nb1 = (size_t)(lc->blocks[3] - lc->blocks[2]), nb2 = (size_t)(lc->blocks[5] - lc->blocks[4]);
        std::vector<double> small;
        small.insert(small.end(), lc->lambda, lc->lambda + nl);
        small.insert(small.end(), lc->planck2, lc->planck2 + nl);
        small.insert(small.end(), lc->sigma_bf1, lc->sigma_bf1 + nb1);
        small.insert(small.end(), lc->sigma_bf2, lc->sigma_bf2 + nb2);
#define VRT_S(expr) do { rc = (expr); if (rc) { lambda_free(s); return rc; } } while (0)
        VRT_S(upload(&s->d_small, small.data(), small.size(), st));
        VRT_S(upload(&s->d_velocity, lc->velocity, 3 * n, st));
        VRT_S(upload(&s->d_doppler, lc->doppler_width, n, st));
        VRT_S(upload(&s->d_gamma_static, lc->gamma_static, n, st));
        VRT_S(upload(&s->d_gamma_unsold, lc->gamma_unsold, n, st));
        VRT_S(upload(&s->d_alpha_cont, lc->alpha_cont, n, st));
        VRT_S(upload(&s->d_eps, lc->eps, n, st));
        VRT_S(upload(&s->d_temperature, lc->temperature, n, st));
        VRT_S(upload(&s->d_atom, lc->atom_density, n, st));
        VRT_S(upload(&s->d_B0, lc->B0, n * nl, st));
        VRT_S(upload(&s->d_lte, lc->lte_populations, 3 * n, st));
        VRT_S(upload(&s->d_C, lc->C, 9 * n, st));
        VRT_S(upload(&s->d_pops, lc->lte_populations, 3 * n, st));        // populations = copy(LTE_pops), :232
        VRT_S(upload(&s->d_S_new, lc->B0, n * nl, st));                   // S_new = B_0, :236-239
        VRT_S(dalloc(&s->d_S_old, n * nl));
        VRT_S(dalloc(&s->d_J, n * nl));
        VRT_S(dalloc(&s->d_gamma, n));
        VRT_S(dalloc(&s->d_strength, n));
        VRT_S(dalloc(&s->d_pops_new, 3 * n));
        VRT_S(dalloc(&s->d_R, 9 * n));
        VRT_S(dalloc(&s->d_I0, (size_t)g->up.n1 * nl));
        VRT_S(dalloc(&s->d_native, (size_t)vrt_plan_native_alpha_count(p, nlam)));
        VRT_S(dalloc(&s->d_scalars, 2));
        if (hipMemsetAsync(s->d_S_old, 0, sizeof(double) * n * nl, st) != hipSuccess ||      // S_old = zero(S_new), :240
            hipMemsetAsync(s->d_J, 0, sizeof(double) * n * nl, st) != hipSuccess) {
            lambda_free(s);
            return fail(VRT_ENODEVICE, "hipMemsetAsync failed");
        }
        VRT_S(launch_gather_rows(g->up.n1, nlam, nlam, g->up.d_order, s->d_B0, s->d_I0, st));   // I_0 = B_λ(λ_l, T) of the bottom layer, :99-101
#undef VRT_S
        if (hipStreamSynchronize(st) != hipSuccess) {                       // the host arrays may go after return
            lambda_free(s);
            return fail(VRT_ENODEVICE, "uploading the line case failed");
        }
        *out = s;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_lambda_iterate(vrt_lambda *s, double *max_rel_change)
{
    DeviceScope scope;
    if (!s || !max_rel_change) return fail(VRT_EINVAL, "NULL argument");
    try {
        vrt_plan *p = s->p;
        std::lock_guard<std::mutex> lock(p->mu);
        vrt_grid *g = p->g;
        int rc = use_device(g->device);
        if (rc) return rc;
        hipStream_t st = g->stream;
        const int64_t n = s->n, nlam = s->nlam;
        const size_t bytes = sizeof(double) * (size_t)n * (size_t)nlam;
        VRT_HIP_TRY(hipMemcpyAsync(s->d_S_old, s->d_S_new, bytes, hipMemcpyDeviceToDevice, st));     // S_old = copy(S_new), :258
        // γ and the line strength of the current populations (:72-75, line.jl:219-225), α_tot of every angle (:89-96)
        if ((rc = launch_line_terms(n, s->d_gamma_static, s->d_gamma_unsold, s->d_pops, s->strength_const, s->Bij,
                                    s->Bji, s->d_gamma, s->d_strength, st)))
            return rc;
        if ((rc = launch_line_opacity(p, nlam, s->d_small, s->lambda0, s->c0, s->d_velocity, s->d_doppler, s->d_gamma,
                                      s->d_strength, s->d_alpha_cont, s->d_native, st)))
            return rc;
        // J_λ (:84-111)
        if ((rc = execute_dev_locked(p, nlam, nlam, s->d_S_old, s->d_native, VRT_ALPHA_ANGLE_NATIVE, s->d_I0, nullptr,
                                     s->weights.data(), s->d_J, nullptr, st)))
            return rc;
        // S_new = (1 - ε) J + ε B_0 and the criterion's scalar (:261-263, :325-349)
        if ((rc = launch_lambda_update(n, nlam, nlam, s->d_J, s->d_B0, s->d_eps, s->d_S_old, s->d_S_new, s->d_scalars, st)))
            return rc;
        // R, populations (:269, :274)
        if ((rc = launch_rates_populations(g, nlam, nlam, s->blocks, s->d_small, s->d_J, s->lambda0, s->c0, s->d_doppler,
                                           s->d_gamma, s->sigma_bb_const, s->d_temperature, s->d_lte, s->hc_over_kB,
                                           s->pref_ij, s->pref_ji, s->d_C, s->d_atom, s->d_R, s->d_pops_new, st)))
            return rc;
        std::swap(s->d_pops, s->d_pops_new);
        unsigned long long h[2] = {0, 0};
        VRT_HIP_TRY(hipMemcpyAsync(h, s->d_scalars, sizeof(h), hipMemcpyDeviceToHost, st));
        VRT_HIP_TRY(hipStreamSynchronize(st));
        double d;
        std::memcpy(&d, &h[0], sizeof(double));
        *max_rel_change = h[1] ? std::nan("") : d;
        s->iterations++;
        return VRT_OK;
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_lambda_get(vrt_lambda *s, double *J, double *S, double *populations, double *R, double *gamma)
{
    DeviceScope scope;
    if (!s) return fail(VRT_EINVAL, "NULL session");
    vrt_plan *p = s->p;
    std::lock_guard<std::mutex> lock(p->mu);
    int rc = use_device(p->g->device);
    if (rc) return rc;
    const size_t n = (size_t)s->n, nl = (size_t)s->nlam;
    if (J) VRT_HIP_TRY(hipMemcpy(J, s->d_J, sizeof(double) * n * nl, hipMemcpyDeviceToHost));
    if (S) VRT_HIP_TRY(hipMemcpy(S, s->d_S_new, sizeof(double) * n * nl, hipMemcpyDeviceToHost));
    if (populations) VRT_HIP_TRY(hipMemcpy(populations, s->d_pops, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    if (R) VRT_HIP_TRY(hipMemcpy(R, s->d_R, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
    if (gamma) VRT_HIP_TRY(hipMemcpy(gamma, s->d_gamma, sizeof(double) * n, hipMemcpyDeviceToHost));
    return VRT_OK;
}

void vrt_lambda_destroy(vrt_lambda *s)
{
    DeviceScope scope;
    if (s) (void)hipSetDevice(s->device);
    lambda_free(s);
}

}  // extern "C"
