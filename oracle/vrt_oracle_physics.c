/*
 * vrt_oracle_physics.c -- CPU ORACLE for the two steps either side of the formal solve that the
 * library also runs on the device: the per-angle opacity prologue and
 * the rates / populations epilogue of a Λ-iteration.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see vrt_oracle.c): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY STATUS: "parity unpinned".  The reference's arithmetic for these steps goes through
 * Transparency.jl (`voigt_profile`, `humlicek`; version unpinned, source not under
 * /root/reference: call sites src/line.jl:133, src/rates.jl:388,408) and through Unitful unit
 * conversions whose floating-point sequence is not visible in the source, and the reference holds
 * no test or fixture for them.  What is restated here is
 *   - the reference's own formulas around those calls, line by line (citations below), with every
 *     quantity stripped to a plain number in ONE consistent unit system chosen by the caller,
 *   - for `voigt_profile`: the PUBLISHED algorithm Transparency.jl documents it uses, Humlíček's
 *     w4 (J. Humlíček 1982, JQSRT 27, 437), H(a, v) = Re w4(v + i a), profile = H / (sqrt(π) ΔλD).
 *     w4's own relative accuracy is 1e-4; tests/test_physics.py checks it against
 *     scipy.special.wofz to that tolerance on committed fixtures.
 * The HIP kernels are compared with THIS restatement at 1e-12 relative.
 *
 * Layouts are Julia's column-major ones, as in vrt_oracle.c: (nλ, n) matrices x[l + nλ i].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef int64_t i64;

/* Humlíček (1982) w4: w(z) = exp(-z^2) erfc(-i z), z = x + i y, y >= 0; four regions, rational
 * approximations in t = y - i x.  Complex arithmetic written out on (re, im) pairs. */
typedef struct { double re, im; } cplx;
static cplx c_mul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static cplx c_add(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static cplx c_sub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
static cplx c_real(double x) { cplx r = {x, 0.0}; return r; }
static cplx c_scale(cplx a, double s) { cplx r = {a.re * s, a.im * s}; return r; }
static cplx c_div(cplx a, cplx b)
{
    const double d = b.re * b.re + b.im * b.im;
    cplx r = {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
    return r;
}
static cplx c_exp(cplx a)
{
    const double e = exp(a.re);
    cplx r = {e * cos(a.im), e * sin(a.im)};
    return r;
}

void orc_humlicek_w4(double x, double y, double *re, double *im)
{
    const cplx t = {y, -x};
    const double s = fabs(x) + y;
    cplx w;
    if (s >= 15.0) {                                   /* region I */
        w = c_div(c_scale(t, 0.5641896), c_add(c_real(0.5), c_mul(t, t)));
    } else if (s >= 5.5) {                             /* region II */
        const cplx u = c_mul(t, t);
        w = c_div(c_mul(t, c_add(c_real(1.410474), c_scale(u, 0.5641896))),
                  c_add(c_real(0.75), c_mul(u, c_add(c_real(3.0), u))));
    } else if (y >= 0.195 * fabs(x) - 0.176) {         /* region III */
        cplx num = c_add(c_real(3.778987), c_scale(t, 0.5642236));
        num = c_add(c_real(11.96482), c_mul(t, num));
        num = c_add(c_real(20.20933), c_mul(t, num));
        num = c_add(c_real(16.4955), c_mul(t, num));
        cplx den = c_add(c_real(6.699398), t);
        den = c_add(c_real(21.69274), c_mul(t, den));
        den = c_add(c_real(39.27121), c_mul(t, den));
        den = c_add(c_real(38.82363), c_mul(t, den));
        den = c_add(c_real(16.4955), c_mul(t, den));
        w = c_div(num, den);
    } else {                                           /* region IV */
        const cplx u = c_mul(t, t);
        cplx num = c_sub(c_real(1.320522), c_scale(u, 0.56419));
        num = c_sub(c_real(35.76683), c_mul(u, num));
        num = c_sub(c_real(219.0313), c_mul(u, num));
        num = c_sub(c_real(1540.787), c_mul(u, num));
        num = c_sub(c_real(3321.9905), c_mul(u, num));
        num = c_sub(c_real(36183.31), c_mul(u, num));
        cplx den = c_sub(c_real(1.841439), u);
        den = c_sub(c_real(61.57037), c_mul(u, den));
        den = c_sub(c_real(364.2191), c_mul(u, den));
        den = c_sub(c_real(2186.181), c_mul(u, den));
        den = c_sub(c_real(9022.228), c_mul(u, den));
        den = c_sub(c_real(24322.84), c_mul(u, den));
        den = c_sub(c_real(32066.6), c_mul(u, den));
        w = c_sub(c_exp(u), c_div(c_mul(t, num), den));
    }
    *re = w.re;
    *im = w.im;
}

/* voigt_profile(a, v, ΔλD) = H(a, v) / (sqrt(π) ΔλD)   (Transparency.jl; used at line.jl:133) */
double orc_voigt_profile(double a, double v, double dD)
{
    double re, im;
    orc_humlicek_w4(v, a, &re, &im);
    return re / (sqrt(3.14159265358979323846) * dD);
}

/* damping (src/broadening.jl:87-89): γ λ² / (4 π c_0 ΔλD) */
double orc_damping(double gamma, double lambda, double dD, double c0)
{
    return gamma * (lambda * lambda) / (4.0 * 3.14159265358979323846 * c0 * dD);
}

/* ------------------------------------------------------------------------------------------
 * α_tot for one direction k: the part of J_λ_voronoi between the angle loop header and the formal
 * solve, src/lambda_iteration.jl:72-80 (damping_λ), :89 (compute_voigt_profile, src/line.jl:121-137
 * with line_of_sight_velocity, :198-208, evaluated for -k), :93-96 (αline_λ, src/line.jl:219-225,
 * + α_cont).
 *   velocity (3, n) rows z,x,y;  doppler ΔλD[n];  gamma γ[n];
 *   line_strength[n] = h c_0/(4π λ0) (n_i B_ij - n_j B_ji)  (the λ-independent factor of αline_λ)
 *   alpha (nλ, n) out.
 * ------------------------------------------------------------------------------------------ */
void orc_line_opacity(const double *k, i64 n, i64 nlam, const double *lambda, double lambda0, double c0,
                      const double *velocity, const double *doppler, const double *gamma,
                      const double *line_strength, const double *alpha_cont, double *alpha)
{
    for (i64 i = 0; i < n; i++) {
        /* v_los = dot(velocity, -k)   (line.jl:126, :205) */
        const double v_los = velocity[3 * i] * (-k[0]) + velocity[3 * i + 1] * (-k[1]) + velocity[3 * i + 2] * (-k[2]);
        for (i64 l = 0; l < nlam; l++) {
            const double a = orc_damping(gamma[i], lambda[l], doppler[i], c0);            /* :79 */
            const double v = (lambda[l] - lambda0 + lambda0 * v_los / c0) / doppler[i];     /* line.jl:132 */
            const double profile = orc_voigt_profile(a, v, doppler[i]);                     /* :133 */
            alpha[l + nlam * i] = line_strength[i] * profile + alpha_cont[i];              /* :93-96 */
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Radiative rates of the two-level-plus-continuum atom -- calculate_R, src/rates.jl:154-201 with
 * Rij / Rji (:226-364), σij (:374-416, static profile: v = (λ - λ0)/ΔλD), Gij (:459-476).
 * Wavelength blocks [lo, hi) (0-based) of the line: bb, bf of level 1, bf of level 2
 * (line.λidx, src/line.jl:59).  Every quantity is a plain number; the unit factors the reference
 * gets from Unitful (and its explicit /1000 in Rij, :237) are folded into pref_ij / pref_ji by the
 * caller:
 *   R_ij[i] = Σ_l pref_ij ((λ_l σ_l J_l + λ_{l+1} σ_{l+1} J_{l+1}) (λ_{l+1} - λ_l))
 *   R_ji[i] = Σ_l pref_ji ((σ_l G_l λ_l (P_l + J_l) + σ_{l+1} G_{l+1} λ_{l+1} (P_{l+1} + J_{l+1})) (λ_{l+1} - λ_l))
 * with P_l = 2 h c_0² / λ_l^5 (planck2[l], in J's unit), G_l = n_ratio exp(-hc_over_kB/(λ_l T)),
 * n_ratio = LTE[i] / LTE[j].
 *   J (nλ, n);  sigma_bf1[nλ_bf1], sigma_bf2[nλ_bf2] (σic, :425-442);  sigma_bb_const = h c_0/(4π λ0) B_ij
 *   lte (n, 3) column-major: lte[i + n*(level-1)];  R (3, 3, n): R[(r-1) + 3*(c-1) + 9*i]
 * ------------------------------------------------------------------------------------------ */
static double orc_G(double n_ratio, double hc_over_kB, double lambda, double T)
{
    return n_ratio * exp(-hc_over_kB / (lambda * T));                       /* rates.jl:473 */
}

void orc_calculate_R(i64 n, i64 nlam, const double *lambda, const i64 *blocks /* 6: lo,hi x (bb,bf1,bf2) */,
                     const double *J, const double *planck2, double lambda0, double c0,
                     const double *doppler, const double *gamma, double sigma_bb_const,
                     const double *sigma_bf1, const double *sigma_bf2, const double *temperature,
                     const double *lte, double hc_over_kB, double pref_ij, double pref_ji, double *R)
{
    for (i64 i = 0; i < n; i++) {
        double *Ri = R + 9 * i;
        for (int q = 0; q < 9; q++) Ri[q] = 0.0;                             /* diagonal zeroed, :195-197 */
        /* ionisation / recombination, levels 1 and 2 <-> continuum (3)   :170-178 */
        for (int level = 1; level <= 2; level++) {
            const i64 lo = blocks[2 * level], hi = blocks[2 * level + 1];
            const double *sig = level == 1 ? sigma_bf1 : sigma_bf2;
            const double n_ratio = lte[i + n * (level - 1)] / lte[i + n * 2];
            double rij = 0.0, rji = 0.0;
            for (i64 l = lo; l + 1 < hi; l++) {
                const double dl = lambda[l + 1] - lambda[l];
                const double s0 = sig[l - lo], s1 = sig[l + 1 - lo];
                const double J0 = J[l + nlam * i], J1 = J[l + 1 + nlam * i];
                rij += pref_ij * ((lambda[l] * s0 * J0 + lambda[l + 1] * s1 * J1) * dl);          /* :262-263 */
                const double G0 = orc_G(n_ratio, hc_over_kB, lambda[l], temperature[i]);
                const double G1 = orc_G(n_ratio, hc_over_kB, lambda[l + 1], temperature[i]);
                rji += pref_ji * ((s0 * G0 * lambda[l] * (planck2[l] + J0) +
                                   s1 * G1 * lambda[l + 1] * (planck2[l + 1] + J1)) * dl);         /* :357-358 */
            }
            Ri[(level - 1) + 3 * 2] = rij;      /* R[level, 3] */
            Ri[2 + 3 * (level - 1)] = rji;      /* R[3, level] */
        }
        /* bound-bound 1 <-> 2   :183-191 */
        {
            const i64 lo = blocks[0], hi = blocks[1];
            const double n_ratio = lte[i] / lte[i + n];
            double rij = 0.0, rji = 0.0;
            double s_prev = 0.0, G_prev = 0.0;
            for (i64 l = lo; l < hi; l++) {
                /* σij (:395-414): σ_constant * voigt_profile(damping_λ[l, i], (λ - λ0)/ΔλD, ΔλD) */
                const double a = gamma[i] * (lambda[l] * lambda[l]) / (4.0 * 3.14159265358979323846 * c0 * doppler[i]);
                const double v = (lambda[l] - lambda0) / doppler[i];
                double re, im;
                orc_humlicek_w4(v, a, &re, &im);
                const double s = sigma_bb_const * (re / (sqrt(3.14159265358979323846) * doppler[i]));
                const double G = orc_G(n_ratio, hc_over_kB, lambda[l], temperature[i]);
                if (l > lo) {
                    const double dl = lambda[l] - lambda[l - 1];
                    const double J0 = J[l - 1 + nlam * i], J1 = J[l + nlam * i];
                    rij += pref_ij * ((lambda[l - 1] * s_prev * J0 + lambda[l] * s * J1) * dl);   /* :236-237 */
                    rji += pref_ji * ((s_prev * G_prev * lambda[l - 1] * (planck2[l - 1] + J0) +
                                       s * G * lambda[l] * (planck2[l] + J1)) * dl);              /* :312-313 */
                }
                s_prev = s;
                G_prev = G;
            }
            Ri[0 + 3 * 1] = rij;                /* R[1, 2] */
            Ri[1 + 3 * 0] = rji;                /* R[2, 1] */
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * get_revised_populations -- src/populations.jl:191-221, n_levels = 2 (+ continuum).
 * P = R + C; A (2 x 2) and b as :205-214; populations[i, 2:3] = inv(A) b (:216-218), here as a
 * 2 x 2 solve with partial pivoting; populations[i, 1] = atom_density - Σ (:219).
 *   R, C (3, 3, n);  atom_density[n];  populations (n, 3) column-major out.
 * ------------------------------------------------------------------------------------------ */
void orc_solve2(const double A[4] /* column-major */, const double b[2], double x[2])
{
    double a00 = A[0], a10 = A[1], a01 = A[2], a11 = A[3], b0 = b[0], b1 = b[1];
    if (fabs(a10) > fabs(a00)) {                    /* partial pivoting */
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

void orc_revised_populations(i64 n, const double *R, const double *C, const double *atom_density,
                             double *populations)
{
    for (i64 i = 0; i < n; i++) {
        double P[9];
        for (int q = 0; q < 9; q++) P[q] = R[9 * i + q] + C[9 * i + q];
#define PP(r, c) P[((r) - 1) + 3 * ((c) - 1)]
        double A[4], b[2], x[2];
        for (int r = 1; r <= 2; r++) {
            const int c = 3 - r;                                 /* setdiff(1:2, r) */
            double arr = PP(1, r + 1) + PP(r + 1, 1);            /* :206 */
            A[(r - 1) + 2 * (c - 1)] = PP(1, r + 1) - PP(c + 1, r + 1);   /* :208 */
            arr += PP(r + 1, c + 1);                             /* :209 */
            A[(r - 1) + 2 * (r - 1)] = arr;
            b[r - 1] = atom_density[i] * PP(1, r + 1);           /* :212 */
        }
#undef PP
        orc_solve2(A, b, x);
        populations[i + n * 1] = x[0];
        populations[i + n * 2] = x[1];
        populations[i] = atom_density[i] - (x[0] + x[1]);        /* :219 */
    }
}

/* ------------------------------------------------------------------------------------------
 * γ of the current populations and the λ-independent factor of αline_λ:
 *   γ_constant, src/broadening.jl:63-82:  γ = γ_unsold(const, T, n_HI) + 4.702e8 + γ_linear_stark(n_e, 2, 1)
 *   + γ_quadratic_stark(n_e, T), which J_λ_voronoi evaluates every iteration with
 *   n_HI = populations[:, 1] .+ populations[:, 2] (src/lambda_iteration.jl:72-75).  The three broadening
 *   functions live in Transparency.jl (absent, unpinned); what the loop needs of them is their dependence on
 *   the populations: only the van der Waals term has one, and it is linear in the neutral density.  So
 *     γ[i] = gamma_static[i] + gamma_unsold[i] (n_1[i] + n_2[i])
 *   with gamma_static = natural + Stark widths and gamma_unsold = γ_unsold per unit density, both per site
 *   from the caller.
 *   αline_λ, src/line.jl:219-225:  strength[i] = strength_const (n_1 B_ij - n_2 B_ji)
 * populations (n, 3) column-major.
 * ------------------------------------------------------------------------------------------ */
void orc_line_terms(i64 n, const double *gamma_static, const double *gamma_unsold, const double *populations,
                    double strength_const, double Bij, double Bji, double *gamma, double *strength)
{
    for (i64 i = 0; i < n; i++) {
        const double n1 = populations[i], n2 = populations[i + n];
        gamma[i] = gamma_static[i] + gamma_unsold[i] * (n1 + n2);
        strength[i] = strength_const * (n1 * Bij - n2 * Bji);
    }
}
