// The two steps either side of the formal solve in a Λ-iteration, on the device (SURVEY.md 8f rows 2
// and 4), so that a whole iteration J -> S_new, R, populations -> α -> J ... stays in HBM:
//
//   k_line_opacity        per-angle α_tot = αline_λ + α_cont from per-site line parameters
//                         (src/lambda_iteration.jl:72-80, :89, :93-96; src/line.jl:121-137, :198-208,
//                         :219-225), written DIRECTLY in the sweep's native storage-pair layout
//                         (VRT_ALPHA_ANGLE_NATIVE): the (nλ, n, n_angles) array and its layout change
//                         never exist
//   k_rates_populations   radiative rates R (calculate_R, src/rates.jl:154-201 with Rij / Rji
//                         :226-364, σij :374-416, Gij :459-476) and the statistical-equilibrium
//                         populations (get_revised_populations, src/populations.jl:191-221) from J in
//                         place: per site a map over the wavelengths, nothing but R (9 n) and the
//                         populations (3 n) leaves the kernel
//
// The Voigt profile is Transparency.jl's `voigt_profile` (absent from the reference checkout,
// version unpinned): its documented algorithm, Humlíček's w4 (JQSRT 27, 437, 1982), is restated
// here as in oracle/vrt_oracle_physics.c (per-site divisors inverted once in the opacity kernel); parity for these two kernels is
// tolerance-based (1e-12 against that restatement; w4 itself is a 1e-4 approximation of the
// Faddeeva function).  All quantities are plain numbers in one unit system chosen by the caller;
// physical constants and unit factors come in as arguments.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "vrt_device.h"
#include "vrt_internal.h"

namespace vrt {

// d = a b + c as ONE three-address instruction.  hipcc turns a Horner step whose addend is a constant kept in a vector
// register into a copy of the constant plus a two-address v_fmac (the opacity kernel: 67 of its 583 vector instructions were
// such copies, and its time is its vector-instruction count); the three-address form needs no copy.
__device__ __forceinline__ double fma3(double a, double b, double c)
{
#ifdef VRT_NO_FMA3
    return fma(a, b, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}

struct cplx { double re, im; };
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx c_real(double x) { return {x, 0.0}; }
__device__ __forceinline__ cplx c_scale(cplx a, double s) { return {a.re * s, a.im * s}; }

// Re w4(x + i y), y >= 0: Humlíček's four regions.  The opacity kernel is bound by this arithmetic (51 evaluations
// per site and angle, ~290 fp64 instruction slots each as the oracle writes it), so the Horner steps use fused
// multiply-adds (c + t p in 4 instructions instead of 7) and the one real part that is needed is formed with a
// Newton-refined reciprocal instead of two divisions: last-bit differences from the oracle (contract 1e-12).
__device__ __forceinline__ cplx c_fma(cplx t, cplx p, double c)           // c + t p
{
    return {fma(t.re, p.re, fma(-t.im, p.im, c)), fma(t.re, p.im, t.im * p.re)};
}
__device__ __forceinline__ cplx c_fms(cplx t, cplx p, double c)           // c - t p
{
    return {fma(-t.re, p.re, fma(t.im, p.im, c)), -fma(t.re, p.im, t.im * p.re)};
}
__device__ __forceinline__ double c_div_re(cplx a, cplx b)                // Re(a / b)
{
    const double d = fma(b.re, b.re, b.im * b.im);
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);                        // (v_rcp_f64 is good to ~2^-23: the second step is needed below 1e-13)
    return fma(a.re, b.re, a.im * b.im) * r;
}
// cos z for the |z| <= ~12 that region 4 of w4 can produce (z = -2 x y, |x| + y < 5.5): when every lane of the wave has
// |z| <= pi/4 -- the narrow damping wings of a stellar atmosphere: always -- its Taylor polynomial to z^14 (remainder
// 1e-15); otherwise Cody-Waite reduction by pi/2 (fdlibm's two-part split: exact for |k| <= 2^20) and the sine /
// cosine polynomials of the reduced argument.  (libm's cos drags its large-argument reduction into the kernel:
// 102 -> VGPRs and a third of this region's instructions.)
__device__ __forceinline__ double cos_poly(double z2)
{
    double p = -1.0 / 87178291200.0;
    p = fma3(p, z2, 1.0 / 479001600.0);
    p = fma3(p, z2, -1.0 / 3628800.0);
    p = fma3(p, z2, 1.0 / 40320.0);
    p = fma3(p, z2, -1.0 / 720.0);
    p = fma3(p, z2, 1.0 / 24.0);
    p = fma(p, z2, -0.5);
    return fma(p, z2, 1.0);
}
__device__ __forceinline__ double cos_small(double z)
{
    if (__ballot(fabs(z) > 0.78539816339744831) == 0ull) return cos_poly(z * z);
    const double kf = rint(z * 0.63661977236758134308);
    double r = fma(-kf, 1.57079632673412561417e+00, z);
    r = fma(-kf, 6.07710050650619224932e-11, r);
    const double r2 = r * r;
    double sp = -1.0 / 1307674368000.0;
    sp = fma(sp, r2, 1.0 / 6227020800.0);
    sp = fma(sp, r2, -1.0 / 39916800.0);
    sp = fma(sp, r2, 1.0 / 362880.0);
    sp = fma(sp, r2, -1.0 / 5040.0);
    sp = fma(sp, r2, 1.0 / 120.0);
    sp = fma(sp, r2, -1.0 / 6.0);
    const double sn = fma(sp * r2, r, r), cs = cos_poly(r2);
    const int q = (int)kf & 3;
    const double v = (q & 1) ? sn : cs;
    return (q == 1 || q == 2) ? -v : v;
}

// A polynomial with REAL coefficients at a complex point z costs two fused multiply-adds per coefficient, not the four of
// a complex Horner step (Knuth, TAOCP 4.6.4 (3): with r = 2 Re z, s = |z|^2 the pair a_j = b_(j-1) + r a_(j-1),
// b_j = c_j - s a_(j-1) ends in P(z) = z a + b).  Regions 3 and 4 of w4 are quotients of such polynomials -- in t and in
// m = -t^2 -- and carry ~70 % of the kernel's instructions; against the oracle's complex Horner form the result differs by
// <= 7e-14 relative over the regions' whole domain (contract 1e-12; w4's own accuracy is 1e-4).
__device__ __forceinline__ void zp_step(double &a, double &b, double r, double ms, double c)
{
    const double a0 = a;
    a = fma(r, a0, b);
    b = fma3(ms, a0, c);
}
__device__ __forceinline__ cplx zp_value(cplx z, double a, double b) { return {fma(z.re, a, b), z.im * a}; }

__device__ double humlicek_w4_re(double x, double y)
{
    const cplx t = {y, -x};
    const double s = fabs(x) + y;
    if (s >= 15.0) return c_div_re(c_scale(t, 0.5641896), c_fma(t, t, 0.5));
    if (s >= 5.5) {
        const cplx u = c_mul(t, t);
        return c_div_re(c_mul(t, c_add(c_real(1.410474), c_scale(u, 0.5641896))), c_fma(u, c_add(c_real(3.0), u), 0.75));
    }
    const double t2 = fma(x, x, y * y);                    // |t|^2
    if (y >= 0.195 * fabs(x) - 0.176) {
        const double r = y + y, ms = -t2;
        double na = 0.5642236, nb = 3.778987;
        zp_step(na, nb, r, ms, 11.96482);
        zp_step(na, nb, r, ms, 20.20933);
        zp_step(na, nb, r, ms, 16.4955);
        double da = 1.0, db = 6.699398;
        zp_step(da, db, r, ms, 21.69274);
        zp_step(da, db, r, ms, 39.27121);
        zp_step(da, db, r, ms, 38.82363);
        zp_step(da, db, r, ms, 16.4955);
        return c_div_re(zp_value(t, na, nb), zp_value(t, da, db));
    }
    // region 4 in m = -t^2 = (x^2 - y^2, 2 x y): every coefficient positive
    const cplx m = {fma(x, x, -(y * y)), 2.0 * (x * y)};
    const double r = m.re + m.re, ms = -(t2 * t2);         // |m|^2 = |t|^4
    double na = 0.56419, nb = 1.320522;
    zp_step(na, nb, r, ms, 35.76683);
    zp_step(na, nb, r, ms, 219.0313);
    zp_step(na, nb, r, ms, 1540.787);
    zp_step(na, nb, r, ms, 3321.9905);
    zp_step(na, nb, r, ms, 36183.31);
    double da = 1.0, db = 1.841439;
    zp_step(da, db, r, ms, 61.57037);
    zp_step(da, db, r, ms, 364.2191);
    zp_step(da, db, r, ms, 2186.181);
    zp_step(da, db, r, ms, 9022.228);
    zp_step(da, db, r, ms, 24322.84);
    zp_step(da, db, r, ms, 32066.6);
    // exp(u.re) cos(u.im), u = t^2 = -m: here y < 0.195 |x| - 0.176, so u.re < 0 (table-driven exp_neg_tab,
    // vrt_device.h) and |u.im| is small for the narrow damping wings of a stellar atmosphere: when every lane of the
    // wave has |u.im| <= pi/4 the cosine is its Taylor polynomial to z^14 (remainder 1e-15), no range reduction.
    const double ex = exp_neg_tab(m.re);
    return ex * cos_small(m.im) - c_div_re(c_mul(t, zp_value(m, na, nb)), zp_value(m, da, db));
}

constexpr double kPi = 3.14159265358979323846;

// ---- opacity prologue -------------------------------------------------------------------------------
// one thread per storage position of an angle's direction (blockIdx.y = active angle: ONE launch for all of them); it
// walks the wavelength pairs, so the site's line parameters are read once and every pair plane is written coalesced
// (16 B/lane).  T2 = double2, or float2 for the fp32 VALUE path (the arithmetic stays fp64, the stored pair is rounded).
// The kernel is bound by the COUNT of its fp64 instructions (measured: time is linear in it, ~linear in nothing else --
// eight instead of four waves per SIMD change nothing), so what does not depend on the wavelength is
// folded per site: gamma / (4 pi c dlambda_D), strength / (sqrt(pi) dlambda_D), and the profile enters alpha_tot by one
// fused multiply-add (last-bit differences from the oracle's expression order; contract of this row: 1e-12).
struct OpacityAngles {
    double k[kMaxAngles][3];
    unsigned char down[kMaxAngles];
};
template <typename T2>
__global__ void __launch_bounds__(256)
k_line_opacity(int64_t n, int nlam, int npair, int lgB, const int32_t *__restrict__ store_up, const int32_t *__restrict__ store_down,
               OpacityAngles ang, const double *__restrict__ lambda, double lambda0, double c0, const double *__restrict__ velocity,
               const double *__restrict__ doppler, const double *__restrict__ gamma,
               const double *__restrict__ strength, const double *__restrict__ alpha_cont,
               T2 *__restrict__ out0 /* pair planes of the plan's native layout (vrt_device.h: pair_index), angle after angle */,
               size_t plane_pairs)
{
    exp2_table_fill();
    __syncthreads();
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int ia = blockIdx.y;
    const int32_t site = (ang.down[ia] ? store_down : store_up)[pos];
    T2 *__restrict__ out = out0 + (size_t)ia * plane_pairs;
    // v_los = dot(velocity, -k)   (line.jl:126, :205)
    const double v_los = velocity[3 * (size_t)site] * (-ang.k[ia][0]) + velocity[3 * (size_t)site + 1] * (-ang.k[ia][1]) +
                         velocity[3 * (size_t)site + 2] * (-ang.k[ia][2]);
    const double dD = doppler[site], ac = alpha_cont[site];
    const double r_dD = 1.0 / dD;
    const double ga = gamma[site] / (4.0 * kPi * c0 * dD);                 // a = gamma lambda^2 / (4 pi c dlambda_D), broadening.jl:87-89
    const double sp = strength[site] / (sqrt(kPi) * dD);                   // alpha_line = strength H / (sqrt(pi) dlambda_D), line.jl:133, :219-225
    const double shift = lambda0 * v_los / c0;
    for (int q = 0; q < npair; q++) {
        double v2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int l = 2 * q + h;
            if (l < nlam) {
                const double lam = lambda[l];
                const double a = ga * (lam * lam);
                const double v = (lam - lambda0 + shift) * r_dD;                       // line.jl:132
                v2[h] = fma(sp, humlicek_w4_re(v, a), ac);                             // lambda_iteration.jl:93-96
            } else
                v2[h] = ac;                                                            // padding wavelength: finite
        }
        T2 o;
        o.x = (decltype(o.x))v2[0];
        o.y = (decltype(o.y))v2[1];
        out[pair_index(q, pos, n, lgB, npair)] = o;
    }
}

int launch_line_opacity(vrt_plan *p, int64_t nlam, const double *d_lambda, double lambda0, double c0,
                        const double *d_velocity, const double *d_doppler, const double *d_gamma,
                        const double *d_strength, const double *d_alpha_cont, void *d_out, hipStream_t st, bool f32_out)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int npair = (int)((nlam + 1) / 2);
    const size_t plane_pairs = (size_t)npair * (size_t)n;
    if (p->A <= 0) return VRT_OK;
    OpacityAngles ang;
    std::memset(&ang, 0, sizeof(ang));
    for (int a = 0; a < p->A; a++) {
        for (int j = 0; j < 3; j++) ang.k[a][j] = p->k[3 * (size_t)a + (size_t)j];
        ang.down[a] = p->dir_of_active[(size_t)a] > 0 ? 0 : 1;
    }
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)p->A);
    if (f32_out)
        hipLaunchKernelGGL(k_line_opacity<float2>, grid, dim3(256), 0, st, n, (int)nlam, npair, native_lg(p, f32_out), g->up.d_store,
                           g->down.d_store, ang, d_lambda, lambda0, c0, d_velocity, d_doppler, d_gamma, d_strength, d_alpha_cont,
                           reinterpret_cast<float2 *>(d_out), plane_pairs);
    else
        hipLaunchKernelGGL(k_line_opacity<double2>, grid, dim3(256), 0, st, n, (int)nlam, npair, native_lg(p, f32_out), g->up.d_store,
                           g->down.d_store, ang, d_lambda, lambda0, c0, d_velocity, d_doppler, d_gamma, d_strength, d_alpha_cont,
                           reinterpret_cast<double2 *>(d_out), plane_pairs);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// ---- population-dependent line terms -----------------------------------------------------------------
// γ of the CURRENT populations and the λ-independent factor of αline_λ, per site:
//   γ = γ_static + γ_unsold (n_1 + n_2)     γ_constant, src/broadening.jl:63-82, called with
//                                            populations[:,1] .+ populations[:,2] each iteration (lambda_iteration.jl:72-75);
//                                            γ_static = natural width + the two Stark terms (electron density and
//                                            temperature only), γ_unsold = the van der Waals width per unit neutral density
//   strength = strength_const (n_1 B_ij - n_2 B_ji)      αline_λ, src/line.jl:219-225
// populations (n, 3) column-major as Julia's
__global__ void __launch_bounds__(256)
k_line_terms(int64_t n, const double *__restrict__ gamma_static, const double *__restrict__ gamma_unsold,
             const double *__restrict__ pops, double strength_const, double Bij, double Bji,
             double *__restrict__ gamma, double *__restrict__ strength)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double n1 = pops[i], n2 = pops[i + n];
    if (gamma) gamma[i] = gamma_static[i] + gamma_unsold[i] * (n1 + n2);
    if (strength) strength[i] = strength_const * (n1 * Bij - n2 * Bji);
}

int launch_line_terms(int64_t n, const double *d_gamma_static, const double *d_gamma_unsold, const double *d_pops,
                      double strength_const, double Bij, double Bji, double *d_gamma, double *d_strength, hipStream_t st)
{
    hipLaunchKernelGGL(k_line_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, d_gamma_static,
                       d_gamma_unsold, d_pops, strength_const, Bij, Bji, d_gamma, d_strength);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// ---- rates + populations epilogue ------------------------------------------------------------------
struct RatesArgs {
    int64_t n, nlam, ld;
    int64_t blocks[6];                  // [lo, hi) of the bb, bf-level-1, bf-level-2 wavelength blocks
    const double *lambda, *planck2;     // [nlam] device
    const double *sigma_bf1, *sigma_bf2;   // device, one entry per wavelength of the block
    const double *J;                    // (nlam, n), leading dimension ld
    // ... or per sweep direction in sweep order (vrt_plan_execute_native_dev): J = J_up + J_down; threads walk the up order
    const double *J_up = nullptr, *J_down = nullptr;
    const int32_t *store_up = nullptr, *rank_down = nullptr;
    double lambda0, c0, sigma_bb_const, hc_over_kB, pref_ij, pref_ji;
    const double *doppler, *gamma, *temperature, *lte, *C, *atom_density;
    double *R, *populations;
};

__device__ __forceinline__ void solve2(const double A[4], const double b[2], double x[2])
{
    double a00 = A[0], a10 = A[1], a01 = A[2], a11 = A[3], b0 = b[0], b1 = b[1];
    if (fabs(a10) > fabs(a00)) {                    // partial pivoting
        double t;
        t = a00; a00 = a10; a10 = t;
        t = a01; a01 = a11; a11 = t;
        t = b0; b0 = b1; b1 = t;
    }
    const double m = a10 / a00;
    const double u11 = a11 - m * a01;
    const double y1 = b1 - m * b0;
    x[1] = y1 / u11;
    x[0] = (b0 - a01 * x[1]) / a00;
}

// The rate kernels run 91 Boltzmann factors and 51 Voigt profiles per site and are bound by that arithmetic: the per-site
// divisors (Δλ_D, T) are inverted once, 1/λ comes from the hardware estimate + two Newton steps, exp(-hc/λkT) from the
// table-driven exp_neg_tab -- last-bit differences from the oracle's divisions and libm (contract of this row: 1e-12).
__device__ __forceinline__ double rcp_newton(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ double boltzmann(double hc_over_kT, double lam)      // exp(-hc / (λ k T)), rates.jl:473
{
    return exp_neg_tab(fmin(hc_over_kT * rcp_newton(lam), 745.0));
}

// R of site i out, its statistical equilibrium solved (n_levels = 2, populations.jl:191-221)
__device__ __forceinline__ void populations_from_rates(const RatesArgs &ra, int64_t i, const double R[9])
{
    const int64_t n = ra.n;
    double P[9];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        ra.R[9 * (size_t)i + q] = R[q];
        P[q] = R[q] + ra.C[9 * (size_t)i + q];                                           // populations.jl:195
    }
    // statistical equilibrium, n_levels = 2 (populations.jl:205-219)
#define PP(r, c) P[((r) - 1) + 3 * ((c) - 1)]
    const double N = ra.atom_density[i];
    double A[4], b[2], x[2];
    A[0] = PP(1, 2) + PP(2, 1) + PP(2, 3);
    A[2] = PP(1, 2) - PP(3, 2);
    A[3] = PP(1, 3) + PP(3, 1) + PP(3, 2);
    A[1] = PP(1, 3) - PP(2, 3);
    b[0] = N * PP(1, 2);
    b[1] = N * PP(1, 3);
#undef PP
    solve2(A, b, x);
    ra.populations[i + n] = x[0];
    ra.populations[i + 2 * n] = x[1];
    ra.populations[i] = N - (x[0] + x[1]);
}

// NATIVE: J comes per sweep direction in sweep order; a thread takes the site at up position blockIdx * 256 + tid, so that
// the J_up loads of a wave are contiguous (and the J_down loads piecewise contiguous: layers are the units of both orders) --
// the per-site inputs become gathers instead, 25 values against the 2 x nlam of J.  The arithmetic of a site is the same.
template <bool NATIVE>
__global__ void __launch_bounds__(256)
k_rates_populations(RatesArgs ra)
{
    exp2_table_fill();
    __syncthreads();
    const int tid = threadIdx.x;
    const int64_t n = ra.n, site0 = (int64_t)blockIdx.x * 256;
    const bool valid = site0 + tid < n;
    const int64_t slot = valid ? site0 + tid : n - 1;               // (a thread past the end works on the last site and stores nothing)
    const int64_t i = NATIVE ? (int64_t)ra.store_up[slot] : slot;
    const int64_t pdn = NATIVE ? (int64_t)ra.rank_down[i] : 0;
    int64_t pair_have = -1;
    double2 pair_J = make_double2(0.0, 0.0);
    auto J_native = [&](int64_t l) -> double {
        // J = J_up + J_down as k_combine_J forms it; the two wavelengths of a pair come with one 16-byte load per direction
        const int64_t q = l >> 1;
        if (q != pair_have) {
            const size_t o = (size_t)q * (size_t)n;
            double2 v = make_double2(0.0, 0.0);
            if (ra.J_up) v = reinterpret_cast<const double2 *>(ra.J_up)[o + (size_t)slot];
            if (ra.J_down) {
                const double2 u = reinterpret_cast<const double2 *>(ra.J_down)[o + (size_t)pdn];
                v.x = v.x + u.x; v.y = v.y + u.y;
            }
            pair_J = v;
            pair_have = q;
        }
        return (l & 1) ? pair_J.y : pair_J.x;
    };
    // J is (n, ld) with the wavelength fastest: a thread walking its own row makes every load of the wave touch 64
    // different lines.  The block's 256 rows are staged through LDS 16 wavelengths at a time instead (a site's 128
    // bytes by 16 neighbouring lanes), and a thread picks its row's values up from there (row stride 17: no conflicts)
    __shared__ double Jt[256 * 17];
    auto stage = [&](int64_t l0c, int64_t hi) {
        __syncthreads();                                            // the previous chunk has been consumed
        for (int idx = tid; idx < 256 * 16; idx += 256) {
            const int sr = idx >> 4, c = idx & 15;
            const int64_t site = site0 + sr, l = l0c + c;
            Jt[sr * 17 + c] = (site < n && l < hi) ? ra.J[(size_t)site * (size_t)ra.ld + (size_t)l] : 0.0;
        }
        __syncthreads();
    };
    const double hT = ra.hc_over_kB / ra.temperature[i];
    double R[9];
#pragma unroll
    for (int q = 0; q < 9; q++) R[q] = 0.0;
    // ionisation / recombination: levels 1, 2 <-> continuum (rates.jl:170-178)
    for (int level = 1; level <= 2; level++) {
        const int64_t lo = ra.blocks[2 * level], hi = ra.blocks[2 * level + 1];
        const double *__restrict__ sig = level == 1 ? ra.sigma_bf1 : ra.sigma_bf2;
        const double n_ratio = ra.lte[i + n * (level - 1)] / ra.lte[i + n * 2];
        double rij = 0.0, rji = 0.0, s_prev = 0.0, G_prev = 0.0, J_prev = 0.0;
        for (int64_t l = lo; l < hi; l++) {
            if (!NATIVE && ((l - lo) & 15) == 0) stage(l, hi);
            const double lam = ra.lambda[l], s = sig[l - lo], Jl = NATIVE ? J_native(l) : Jt[tid * 17 + (int)((l - lo) & 15)];
            const double G = n_ratio * boltzmann(hT, lam);                               // Gij, rates.jl:473
            if (l > lo) {
                const double lp = ra.lambda[l - 1], dl = lam - lp;
                rij += ra.pref_ij * ((lp * s_prev * J_prev + lam * s * Jl) * dl);      // :262-263
                rji += ra.pref_ji * ((s_prev * G_prev * lp * (ra.planck2[l - 1] + J_prev) +
                                      s * G * lam * (ra.planck2[l] + Jl)) * dl);        // :357-358
            }
            s_prev = s; G_prev = G; J_prev = Jl;
        }
        R[(level - 1) + 3 * 2] = rij;
        R[2 + 3 * (level - 1)] = rji;
    }
    // bound-bound 1 <-> 2 (rates.jl:183-191): σij carries the static Voigt profile of the site
    {
        const int64_t lo = ra.blocks[0], hi = ra.blocks[1];
        const double n_ratio = ra.lte[i] / ra.lte[i + n];
        const double dD = ra.doppler[i], gm = ra.gamma[i];
        const double r_dD = 1.0 / dD, r_a = 1.0 / (4.0 * kPi * ra.c0 * dD), r_prof = 1.0 / (sqrt(kPi) * dD);
        double rij = 0.0, rji = 0.0, s_prev = 0.0, G_prev = 0.0, J_prev = 0.0;
        for (int64_t l = lo; l < hi; l++) {
            if (!NATIVE && ((l - lo) & 15) == 0) stage(l, hi);
            const double lam = ra.lambda[l], Jl = NATIVE ? J_native(l) : Jt[tid * 17 + (int)((l - lo) & 15)];
            const double a = gm * (lam * lam) * r_a;
            const double v = (lam - ra.lambda0) * r_dD;                                  // rates.jl:408
            const double s = ra.sigma_bb_const * (humlicek_w4_re(v, a) * r_prof);
            const double G = n_ratio * boltzmann(hT, lam);
            if (l > lo) {
                const double lp = ra.lambda[l - 1], dl = lam - lp;
                rij += ra.pref_ij * ((lp * s_prev * J_prev + lam * s * Jl) * dl);      // :236-237
                rji += ra.pref_ji * ((s_prev * G_prev * lp * (ra.planck2[l - 1] + J_prev) +
                                      s * G * lam * (ra.planck2[l] + Jl)) * dl);        // :312-313
            }
            s_prev = s; G_prev = G; J_prev = Jl;
        }
        R[0 + 3 * 1] = rij;
        R[1 + 3 * 0] = rji;
    }
    if (valid) populations_from_rates(ra, i, R);
}

// ---- the same split over wavelength blocks (several devices: vrt_multi_lambda_*, vrt_multi.cpp) -------------------------
// A device that owns the wavelengths [l0, l1) forms ITS share of the six λ-integrals of a site: the trapezoid sums of
// k_rates_populations regrouped per wavelength, Σ_l W_l f_l with W_l = (λ_l - λ_l-1) [l > lo] + (λ_l+1 - λ_l) [l < hi - 1]
// inside each block [lo, hi) -- the same terms, summed in another order (1e-15 relative per term).  The shares are
// summed across the devices (ONE all-reduce of 6 n doubles: no J travels), then every device solves
// the statistical equilibrium of every site itself (k_populations_from_shares).
// shares[q][i], q = (bf1 ij, bf1 ji, bf2 ij, bf2 ji, bb ij, bb ji); J: this device's columns, (l1 - l0) per site
// NATIVE: this device's columns of J as sweep-order plane sets (local wavelength l - l0), threads on up positions
template <bool NATIVE>
__global__ void __launch_bounds__(256)
k_rates_partial(RatesArgs ra, int64_t l0, int64_t l1, double *__restrict__ shares)
{
    exp2_table_fill();
    __syncthreads();
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = ra.n;
    if (slot >= n) return;
    const int64_t i = NATIVE ? (int64_t)ra.store_up[slot] : slot;
    const int64_t pdn = NATIVE ? (int64_t)ra.rank_down[i] : 0;
    const double *__restrict__ Ji = NATIVE ? nullptr : ra.J + (size_t)i * (size_t)ra.ld - l0;      // indexed by the GLOBAL wavelength
    int64_t pair_have = -1;
    double2 pair_J = make_double2(0.0, 0.0);
    auto J_at = [&](int64_t l) -> double {
        if constexpr (!NATIVE) return Ji[l];
        const int64_t ll = l - l0, q = ll >> 1;
        if (q != pair_have) {                                   // J = J_up + J_down as k_combine_J forms it
            const size_t o = (size_t)q * (size_t)n;
            double2 v = make_double2(0.0, 0.0);
            if (ra.J_up) v = reinterpret_cast<const double2 *>(ra.J_up)[o + (size_t)slot];
            if (ra.J_down) {
                const double2 u = reinterpret_cast<const double2 *>(ra.J_down)[o + (size_t)pdn];
                v.x = v.x + u.x; v.y = v.y + u.y;
            }
            pair_J = v;
            pair_have = q;
        }
        return (ll & 1) ? pair_J.y : pair_J.x;
    };
    const double hT = ra.hc_over_kB / ra.temperature[i];
    for (int tr = 0; tr < 3; tr++) {                                              // bf level 1, bf level 2, bb
        const int64_t lo = ra.blocks[tr < 2 ? 2 * (tr + 1) : 0], hi = ra.blocks[tr < 2 ? 2 * (tr + 1) + 1 : 1];
        const double n_ratio = tr < 2 ? ra.lte[i + n * tr] / ra.lte[i + n * 2] : ra.lte[i] / ra.lte[i + n];
        const double *__restrict__ sig = tr == 0 ? ra.sigma_bf1 : ra.sigma_bf2;
        const double dD = ra.doppler[i], gm = ra.gamma[i];
        const double r_dD = 1.0 / dD, r_a = 1.0 / (4.0 * kPi * ra.c0 * dD), r_prof = 1.0 / (sqrt(kPi) * dD);
        double rij = 0.0, rji = 0.0;
        for (int64_t l = lo > l0 ? lo : l0; l < (hi < l1 ? hi : l1); l++) {
            const double lam = ra.lambda[l], Jl = J_at(l);
            double s;
            if (tr < 2) s = sig[l - lo];
            else {
                const double a = gm * (lam * lam) * r_a;
                const double v = (lam - ra.lambda0) * r_dD;                              // rates.jl:408
                s = ra.sigma_bb_const * (humlicek_w4_re(v, a) * r_prof);
            }
            const double G = n_ratio * boltzmann(hT, lam);                               // Gij, rates.jl:473
            double W = 0.0;
            if (l > lo) W += lam - ra.lambda[l - 1];
            if (l < hi - 1) W += ra.lambda[l + 1] - lam;
            rij += (lam * s * Jl) * W;                                                   // rates.jl:236-237, :262-263
            rji += (s * G * lam * (ra.planck2[l] + Jl)) * W;                             // :312-313, :357-358
        }
        shares[(size_t)(2 * tr) * (size_t)n + (size_t)i] = ra.pref_ij * rij;
        shares[(size_t)(2 * tr + 1) * (size_t)n + (size_t)i] = ra.pref_ji * rji;
    }
}

__global__ void __launch_bounds__(256)
k_populations_from_shares(RatesArgs ra, const double *__restrict__ shares)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = ra.n;
    if (i >= n) return;
    double R[9];
#pragma unroll
    for (int q = 0; q < 9; q++) R[q] = 0.0;
    for (int level = 1; level <= 2; level++) {
        R[(level - 1) + 3 * 2] = shares[(size_t)(2 * (level - 1)) * (size_t)n + (size_t)i];
        R[2 + 3 * (level - 1)] = shares[(size_t)(2 * (level - 1) + 1) * (size_t)n + (size_t)i];
    }
    R[0 + 3 * 1] = shares[(size_t)4 * (size_t)n + (size_t)i];
    R[1 + 3 * 0] = shares[(size_t)5 * (size_t)n + (size_t)i];
    populations_from_rates(ra, i, R);
}

static void fill_rates_args(RatesArgs &ra, vrt_grid *g, int64_t nlam, int64_t ld, const int64_t blocks[6], const double *d_small,
                            const double *dJ, double lambda0, double c0, const double *d_doppler, const double *d_gamma,
                            double sigma_bb_const, const double *d_temperature, const double *d_lte, double hc_over_kB,
                            double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density, double *d_R,
                            double *d_populations)
{
    ra.n = g->n; ra.nlam = nlam; ra.ld = ld;
    for (int q = 0; q < 6; q++) ra.blocks[q] = blocks[q];
    ra.lambda = d_small;
    ra.planck2 = d_small + nlam;
    ra.sigma_bf1 = d_small + 2 * nlam;
    ra.sigma_bf2 = ra.sigma_bf1 + (blocks[3] - blocks[2]);
    ra.J = dJ;
    ra.lambda0 = lambda0; ra.c0 = c0; ra.sigma_bb_const = sigma_bb_const; ra.hc_over_kB = hc_over_kB;
    ra.pref_ij = pref_ij; ra.pref_ji = pref_ji;
    ra.doppler = d_doppler; ra.gamma = d_gamma; ra.temperature = d_temperature; ra.lte = d_lte;
    ra.C = d_C; ra.atom_density = d_atom_density; ra.R = d_R; ra.populations = d_populations;
}

// this device's share of the rate integrals: wavelengths [l0, l1) of the FULL wavelength arrays in d_small
// (λ | planck2 | σ_bf1 | σ_bf2 for all nlam wavelengths); dJ holds those columns only (ld values per site)
int launch_rates_partial(vrt_grid *g, int64_t nlam, int64_t l0, int64_t l1, int64_t ld, const int64_t blocks[6],
                         const double *d_small, const double *dJ, double lambda0, double c0, const double *d_doppler,
                         const double *d_gamma, double sigma_bb_const, const double *d_temperature, const double *d_lte,
                         double hc_over_kB, double pref_ij, double pref_ji, double *d_shares, hipStream_t st,
                         const double *dJ_up, const double *dJ_down)
{
    RatesArgs ra;
    fill_rates_args(ra, g, nlam, ld, blocks, d_small, dJ, lambda0, c0, d_doppler, d_gamma, sigma_bb_const, d_temperature, d_lte,
                    hc_over_kB, pref_ij, pref_ji, nullptr, nullptr, nullptr, nullptr);
    if (dJ_up || dJ_down) {
        ra.J_up = dJ_up; ra.J_down = dJ_down;
        ra.store_up = g->up.d_store; ra.rank_down = g->down.d_srank;
        hipLaunchKernelGGL(k_rates_partial<true>, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, st, ra, l0, l1, d_shares);
    } else
    hipLaunchKernelGGL(k_rates_partial<false>, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, st, ra, l0, l1, d_shares);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// R (9 n) and the populations (3 n) from the summed shares
int launch_populations_from_shares(vrt_grid *g, const double *d_shares, const double *d_C, const double *d_atom_density,
                                   double *d_R, double *d_populations, hipStream_t st)
{
    RatesArgs ra;
    const int64_t zero[6] = {0, 0, 0, 0, 0, 0};
    fill_rates_args(ra, g, 0, 0, zero, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, d_C, d_atom_density,
                    d_R, d_populations);
    hipLaunchKernelGGL(k_populations_from_shares, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, st, ra, d_shares);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int launch_rates_populations(vrt_grid *g, int64_t nlam, int64_t ld, const int64_t blocks[6],
                             const double *d_small /* lambda | planck2 | sigma_bf1 | sigma_bf2 */,
                             const double *dJ, double lambda0, double c0, const double *d_doppler,
                             const double *d_gamma, double sigma_bb_const, const double *d_temperature,
                             const double *d_lte, double hc_over_kB, double pref_ij, double pref_ji,
                             const double *d_C, const double *d_atom_density, double *d_R,
                             double *d_populations, hipStream_t st, const double *dJ_up, const double *dJ_down)
{
    RatesArgs ra;
    fill_rates_args(ra, g, nlam, ld, blocks, d_small, dJ, lambda0, c0, d_doppler, d_gamma, sigma_bb_const, d_temperature, d_lte,
                    hc_over_kB, pref_ij, pref_ji, d_C, d_atom_density, d_R, d_populations);
    if (dJ_up || dJ_down) {
        ra.J_up = dJ_up; ra.J_down = dJ_down;
        ra.store_up = g->up.d_store; ra.rank_down = g->down.d_srank;
        hipLaunchKernelGGL(k_rates_populations<true>, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, st, ra);
    } else
    hipLaunchKernelGGL(k_rates_populations<false>, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, st, ra);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

}  // namespace vrt
