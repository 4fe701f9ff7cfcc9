// Host-pointer entry points of the line case: what a single-process host WITHOUT device arrays of its own
// (the reference's Julia driver) calls instead of shipping α_tot (nλ, n, n_angles) through PCIe.
//   vrt_line_terms_dev     γ and the line strength from the current populations (device pointers)
//   vrt_plan_execute_line  the body of J_λ_voronoi, line case (src/lambda_iteration.jl:72-111): per-site line
//                          vectors + S in, J out; α_tot is made on the device and never exists on the host
//   vrt_lambda_*           Λ_voronoi's loop (src/lambda_iteration.jl:205-300) with library-owned device state:
//                          per iteration only the criterion's scalar comes back
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

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

// ---- pageable host arrays at PCIe speed: lanes of (host thread, copy stream, two pinned staging buffers) ----------------
struct CopyJob {
    const char *host_src = nullptr;     // upload: rows of `width` bytes, `hstride` apart
    char *host_dst = nullptr;           // download
    char *dev = nullptr;                // dense on the device: rows x width bytes
    size_t rows = 0, width = 0, hstride = 0;
};

int ensure_copy_lanes(vrt_plan *p, int lanes)
{
    if ((int)p->copy_lanes.size() >= lanes && p->copy_done) return VRT_OK;
    if (!p->copy_done) VRT_HIP_TRY(hipEventCreateWithFlags(&p->copy_done, hipEventDisableTiming));
    while ((int)p->copy_lanes.size() < lanes) {
        CopyLane l;
        VRT_HIP_TRY(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            VRT_HIP_TRY(hipHostMalloc(&l.pin[b], kCopyChunk, hipHostMallocDefault));
            VRT_HIP_TRY(hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming));
        }
        p->copy_lanes.push_back(l);
    }
    return VRT_OK;
}

// rows x width bytes per job <= kCopyChunk
void split_jobs(std::vector<CopyJob> &jobs, const void *host, void *dev, size_t rows, size_t width, size_t hstride, bool download)
{
    const size_t per = std::max<size_t>(1, kCopyChunk / std::max<size_t>(width, 1));
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        CopyJob j;
        j.rows = std::min(per, rows - r0);
        j.width = width;
        j.hstride = hstride;
        j.dev = (char *)dev + r0 * width;
        if (download) j.host_dst = (char *)const_cast<void *>(host) + r0 * hstride;
        else j.host_src = (const char *)host + r0 * hstride;
        jobs.push_back(j);
    }
}

void rows_copy(char *dst, size_t dstride, const char *src, size_t sstride, size_t rows, size_t width)
{
    if (dstride == width && sstride == width) { std::memcpy(dst, src, rows * width); return; }
    for (size_t r = 0; r < rows; r++) std::memcpy(dst + r * dstride, src + r * sstride, width);
}

// every lane takes the jobs lane, lane + L, ...; uploads leave on the lanes' streams (`st` then waits for them), downloads
// start behind `after` (an event of the producing stream) and are complete in host memory on return
int run_copy_jobs(vrt_plan *p, const std::vector<CopyJob> &jobs, bool download, hipEvent_t after, hipStream_t st, int device)
{
    const int L = (int)p->copy_lanes.size();
    std::vector<int> rcs((size_t)L, VRT_OK);
    auto lane_work = [&](int li) {
        CopyLane &l = p->copy_lanes[(size_t)li];
        if (hipSetDevice(device) != hipSuccess) { rcs[(size_t)li] = VRT_ENODEVICE; return; }
        auto ok = [&](hipError_t e) { if (e != hipSuccess) rcs[(size_t)li] = VRT_ENODEVICE; return e == hipSuccess; };
        if (download && after && !ok(hipStreamWaitEvent(l.st, after, 0))) return;
        // (the staging buffers may still feed the last transfers of an EARLIER call: those first)
        for (int b = 0; b < 2; b++)
            if (!ok(hipEventSynchronize(l.ev[b]))) return;
        int pending[2] = {-1, -1};
        int slot = 0;
        for (size_t k = (size_t)li; k < jobs.size(); k += (size_t)L, slot ^= 1) {
            const CopyJob &j = jobs[k];
            const size_t bytes = j.rows * j.width;
            if (pending[slot] >= 0) {                        // the buffer's previous transfer: finished (and, downloading, taken home)
                if (!ok(hipEventSynchronize(l.ev[slot]))) return;
                if (download) {
                    const CopyJob &o = jobs[(size_t)pending[slot]];
                    rows_copy(o.host_dst, o.hstride, (const char *)l.pin[slot], o.width, o.rows, o.width);
                }
            }
            if (download) {
                if (!ok(hipMemcpyAsync(l.pin[slot], j.dev, bytes, hipMemcpyDeviceToHost, l.st))) return;
            } else {
                rows_copy((char *)l.pin[slot], j.width, j.host_src, j.hstride, j.rows, j.width);
                if (!ok(hipMemcpyAsync(j.dev, l.pin[slot], bytes, hipMemcpyHostToDevice, l.st))) return;
            }
            if (!ok(hipEventRecord(l.ev[slot], l.st))) return;
            pending[slot] = (int)k;
        }
        for (int b = 0; b < 2; b++) {                        // drain (oldest first: the slot that is next in turn)
            const int sl = slot ^ b;
            if (pending[sl] < 0) continue;
            if (download) {
                if (!ok(hipEventSynchronize(l.ev[sl]))) return;
                const CopyJob &o = jobs[(size_t)pending[sl]];
                rows_copy(o.host_dst, o.hstride, (const char *)l.pin[sl], o.width, o.rows, o.width);
            }
        }
    };
    if (!run_workers(L, lane_work)) return fail(VRT_ENOMEM, "out of host memory in a copy lane");
    for (int li = 0; li < L; li++)
        if (rcs[(size_t)li]) return fail(rcs[(size_t)li], "a host <-> device copy lane failed");
    if (!download)
        for (int li = 0; li < L; li++) {                     // `st` follows the uploads
            CopyLane &l = p->copy_lanes[(size_t)li];
            for (int b = 0; b < 2; b++) VRT_HIP_TRY(hipStreamWaitEvent(st, l.ev[b], 0));
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
    // S and J in sweep order between the steps of an iteration (VRT_LAMBDA_NATIVE, default): the update kernel writes S where
    // the sweep reads it, the sweep reduces J where the update and the rate integrals read it -- no layout change inside
    // the loop; d_S_new / d_S_old / d_J (the caller's layout) then exist only while vrt_lambda_get fills them
    bool native = false;
    double *d_S_nat[2] = {nullptr, nullptr}, *d_J_nat[2] = {nullptr, nullptr}, *d_B_up = nullptr;
};

static void lambda_free(vrt_lambda *s)
{
    if (!s) return;
    for (double *q : {s->d_small, s->d_velocity, s->d_doppler, s->d_gamma_static, s->d_gamma_unsold, s->d_alpha_cont,
                      s->d_eps, s->d_temperature, s->d_atom, s->d_B0, s->d_lte, s->d_C, s->d_gamma, s->d_strength,
                      s->d_pops, s->d_pops_new, s->d_R, s->d_S_old, s->d_S_new, s->d_J, s->d_I0, s->d_native, s->d_S_nat[0],
                      s->d_S_nat[1], s->d_J_nat[0], s->d_J_nat[1], s->d_B_up})
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
        const size_t n = (size_t)g->n, nl = (size_t)nlam, nS = n * nl;           // dense (n, nlam) on the device
        const size_t nU = (size_t)g->up.n1 * nl, nD = (size_t)g->down.n1 * nl;
        hipStream_t st = g->stream;
        // staging: S | J in the plan's stage buffers; the seven per-site vectors + λ in stage 1; α_tot native in ws_AA
        auto ensure = [&](double *&buf, size_t &cap, size_t count) -> int {
            if (buf && count <= cap) return VRT_OK;
            if (buf) (void)hipFree(buf);
            buf = nullptr;
            cap = 0;
            int r = dalloc(&buf, count);
            if (!r) cap = count;
            return r;
        };
        const size_t vecs = 7 * n + nl;                  // velocity (3n), ΔλD, γ, strength, α_cont, λ
        if ((rc = ensure(p->d_stage[0], p->stage_cap[0], nS))) return rc;
        if ((rc = ensure(p->d_stage[1], p->stage_cap[1], vecs))) return rc;
        if ((rc = ensure(p->d_stage[4], p->stage_cap[4], nS))) return rc;
        const size_t nnat = (size_t)vrt_plan_native_alpha_count(p, nlam);
        if ((rc = ensure(p->ws_AA, p->ws_AA_cap, nnat))) return rc;
        double *dv = p->d_stage[1];
        double *d_vel = dv, *d_dop = dv + 3 * n, *d_gam = dv + 4 * n, *d_str = dv + 5 * n, *d_ac = dv + 6 * n, *d_lam = dv + 7 * n;
        double *dU = nullptr, *dD = nullptr;
        if (I0_up && nU) {
            if ((rc = ensure(p->d_stage[2], p->stage_cap[2], nU))) return rc;
            dU = p->d_stage[2];
        }
        if (I0_down && nD) {
            if ((rc = ensure(p->d_stage[3], p->stage_cap[3], nD))) return rc;
            dD = p->d_stage[3];
        }
        // The caller's arrays are pageable: they cross PCIe through copy lanes (a host thread, a copy stream and two pinned
        // buffers each).  Order: the per-site vectors -> the opacity kernel starts while S is still on its way -> sweep ->
        // J comes home in chunks.  Only the nlam columns of S and J are touched (ld >= nlam: the padding stays the caller's).
        unsigned hw = std::thread::hardware_concurrency();
        const int lanes = (int)std::max(2u, std::min(8u, (hw ? hw : 8u) / 2));
        if ((rc = ensure_copy_lanes(p, lanes))) return rc;
        const size_t w8 = sizeof(double);
#ifdef VRT_DIAG
        const auto t0 = std::chrono::steady_clock::now();
#endif
        std::vector<CopyJob> jobs;
        split_jobs(jobs, velocity, d_vel, 1, w8 * 3 * n, w8 * 3 * n, false);
        split_jobs(jobs, doppler_width, d_dop, 1, w8 * n, w8 * n, false);
        split_jobs(jobs, gamma, d_gam, 1, w8 * n, w8 * n, false);
        split_jobs(jobs, line_strength, d_str, 1, w8 * n, w8 * n, false);
        split_jobs(jobs, alpha_cont, d_ac, 1, w8 * n, w8 * n, false);
        // (a 1-row job wider than a staging buffer is cut by bytes: rows of 1 MiB)
        {
            std::vector<CopyJob> cut;
            for (const CopyJob &j : jobs)
                if (j.rows == 1 && j.width > kCopyChunk) {
                    const size_t piece = (size_t)1 << 20;
                    split_jobs(cut, j.host_src, j.dev, j.width / piece, piece, piece, false);
                    const size_t done = j.width / piece * piece;
                    if (done < j.width) split_jobs(cut, j.host_src + done, j.dev + done, 1, j.width - done, j.width - done, false);
                } else
                    cut.push_back(j);
            jobs.swap(cut);
        }
        split_jobs(jobs, lambda, d_lam, 1, w8 * nl, w8 * nl, false);
        if ((rc = run_copy_jobs(p, jobs, false, nullptr, st, g->device))) return rc;
#ifdef VRT_DIAG
        const auto t1 = std::chrono::steady_clock::now();
#endif
        // α_tot of every angle straight into the native layout (lambda_iteration.jl:72-80, :89, :93-96)
        if ((rc = launch_line_opacity(p, nlam, d_lam, lambda0, c0, d_vel, d_dop, d_gam, d_str, d_ac, p->ws_AA, st))) return rc;
        jobs.clear();
        split_jobs(jobs, S, p->d_stage[0], n, w8 * nl, w8 * (size_t)ld, false);
        if (dU) split_jobs(jobs, I0_up, dU, (size_t)g->up.n1, w8 * nl, w8 * nl, false);
        if (dD) split_jobs(jobs, I0_down, dD, (size_t)g->down.n1, w8 * nl, w8 * nl, false);
        if ((rc = run_copy_jobs(p, jobs, false, nullptr, st, g->device))) return rc;
#ifdef VRT_DIAG
        const auto t2 = std::chrono::steady_clock::now();
#endif
        // (execute_dev_locked reuses ws_AA only for the CALLER-layout per-angle alpha, not for the native one)
        rc = execute_dev_locked(p, nlam, nlam, p->d_stage[0], p->ws_AA, VRT_ALPHA_ANGLE_NATIVE, dU, dD, weights,
                                p->d_stage[4], nullptr, st);
        if (rc) return rc;
        VRT_HIP_TRY(hipEventRecord(p->copy_done, st));
#ifdef VRT_DIAG
        const auto t3 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(st);
        const auto t4 = std::chrono::steady_clock::now();
#endif
        jobs.clear();
        split_jobs(jobs, J, p->d_stage[4], n, w8 * nl, w8 * (size_t)ld, true);
        if ((rc = run_copy_jobs(p, jobs, true, p->copy_done, st, g->device))) return rc;
#ifdef VRT_DIAG
        {
            const auto t5 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[vrt execute_line] lanes %d: vectors up %.1f ms, S staged %.1f ms, launches %.1f ms, wait for the sweep %.1f ms, J down %.1f ms\n",
                         lanes, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
        }
#endif
        VRT_HIP_TRY(hipStreamSynchronize(st));
        return patch_chain_check(p);         // a chained sweep that gave up waiting: THIS call's J is invalid
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
        s->native = p->tune.lambda_native != 0 && native_planes_ok(p) == VRT_OK && p->tune.path != 1 && p->tune.path != 2;
        if (s->native) {
            const size_t np = (size_t)vrt_plan_native_plane_count(p, nlam);
            for (int d = 0; d < 2; d++) {
                VRT_S(dalloc(&s->d_S_nat[d], np));
                VRT_S(dalloc(&s->d_J_nat[d], np));
                if (hipMemsetAsync(s->d_J_nat[d], 0, sizeof(double) * np, st) != hipSuccess) { lambda_free(s); return fail(VRT_ENODEVICE, "hipMemsetAsync failed"); }
            }
            VRT_S(dalloc(&s->d_B_up, np));
            VRT_S(planes_to_native(p, nlam, nlam, s->d_B0, s->d_S_nat[0], s->d_S_nat[1], st));      // S_new = B_0, :236-239
            VRT_S(planes_to_native(p, nlam, nlam, s->d_B0, s->d_B_up, nullptr, st));
        } else {
        VRT_S(upload(&s->d_S_new, lc->B0, n * nl, st));                   // S_new = B_0, :236-239
        VRT_S(dalloc(&s->d_S_old, n * nl));
        VRT_S(dalloc(&s->d_J, n * nl));
        }
        VRT_S(dalloc(&s->d_gamma, n));
        VRT_S(dalloc(&s->d_strength, n));
        VRT_S(dalloc(&s->d_pops_new, 3 * n));
        VRT_S(dalloc(&s->d_R, 9 * n));
        VRT_S(dalloc(&s->d_I0, (size_t)g->up.n1 * nl));
        VRT_S(dalloc(&s->d_native, (size_t)vrt_plan_native_alpha_count(p, nlam)));
        VRT_S(dalloc(&s->d_scalars, 2));
        if (!s->native &&
            (hipMemsetAsync(s->d_S_old, 0, sizeof(double) * n * nl, st) != hipSuccess ||      // S_old = zero(S_new), :240
             hipMemsetAsync(s->d_J, 0, sizeof(double) * n * nl, st) != hipSuccess)) {
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
        if (!s->native)
            VRT_HIP_TRY(hipMemcpyAsync(s->d_S_old, s->d_S_new, bytes, hipMemcpyDeviceToDevice, st));     // S_old = copy(S_new), :258
        // γ and the line strength of the current populations (:72-75, line.jl:219-225), α_tot of every angle (:89-96)
        if ((rc = launch_line_terms(n, s->d_gamma_static, s->d_gamma_unsold, s->d_pops, s->strength_const, s->Bij,
                                    s->Bji, s->d_gamma, s->d_strength, st)))
            return rc;
        if ((rc = launch_line_opacity(p, nlam, s->d_small, s->lambda0, s->c0, s->d_velocity, s->d_doppler, s->d_gamma,
                                      s->d_strength, s->d_alpha_cont, s->d_native, st)))
            return rc;
        if (s->native) {
            // J_λ (:84-111) from the sweep-order S into the sweep-order J; S_new and the criterion (:261-263, :325-349: the old
            // S is read from the plane the new one is written to); R and the populations (:269, :274) from the same J planes
            if ((rc = execute_native_locked(p, nlam, s->d_S_nat[0], s->d_S_nat[1], s->d_native, VRT_ALPHA_ANGLE_NATIVE, s->d_I0, nullptr,
                                            s->weights.data(), s->d_J_nat[0], s->d_J_nat[1], st)))
                return rc;
            if ((rc = launch_lambda_update_native(g, nlam, s->d_J_nat[0], s->d_J_nat[1], s->d_B_up, s->d_eps, s->d_S_nat[0],
                                                  s->d_S_nat[1], s->d_scalars, st)))
                return rc;
            if ((rc = launch_rates_populations(g, nlam, nlam, s->blocks, s->d_small, nullptr, s->lambda0, s->c0, s->d_doppler,
                                               s->d_gamma, s->sigma_bb_const, s->d_temperature, s->d_lte, s->hc_over_kB,
                                               s->pref_ij, s->pref_ji, s->d_C, s->d_atom, s->d_R, s->d_pops_new, st,
                                               s->d_J_nat[0], s->d_J_nat[1])))
                return rc;
        } else {
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
        }
        std::swap(s->d_pops, s->d_pops_new);
        unsigned long long h[2] = {0, 0};
        VRT_HIP_TRY(hipMemcpyAsync(h, s->d_scalars, sizeof(h), hipMemcpyDeviceToHost, st));
        VRT_HIP_TRY(hipStreamSynchronize(st));
        if ((rc = patch_chain_check(p))) return rc;          // a chained sweep that gave up: THIS iteration's results are invalid
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
    if (s->native && (J || S)) {
        // the caller's layout is formed here, on request: one scratch array, freed again
        double *tmp = nullptr;
        if ((rc = dalloc(&tmp, n * nl))) return rc;
        hipStream_t st = p->g->stream;
        if (J) {
            rc = J_from_native(p, s->nlam, s->nlam, s->d_J_nat[0], s->d_J_nat[1], tmp, st);
            if (!rc && hipMemcpyAsync(J, tmp, sizeof(double) * n * nl, hipMemcpyDeviceToHost, st) != hipSuccess) rc = VRT_ENODEVICE;
            if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = VRT_ENODEVICE;
        }
        if (!rc && S) {
            rc = plane_from_native(p, 0, s->nlam, s->nlam, s->d_S_nat[0], tmp, st);
            if (!rc && hipMemcpyAsync(S, tmp, sizeof(double) * n * nl, hipMemcpyDeviceToHost, st) != hipSuccess) rc = VRT_ENODEVICE;
            if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = VRT_ENODEVICE;
        }
        (void)hipFree(tmp);
        if (rc) return rc == VRT_ENODEVICE ? fail(rc, "HIP error in vrt_lambda_get") : rc;
    } else {
    if (J) VRT_HIP_TRY(hipMemcpy(J, s->d_J, sizeof(double) * n * nl, hipMemcpyDeviceToHost));
    if (S) VRT_HIP_TRY(hipMemcpy(S, s->d_S_new, sizeof(double) * n * nl, hipMemcpyDeviceToHost));
    }
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
