/*
 * vrt_oracle.c -- CPU ORACLE for the Voronoi short-characteristics formal solve.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (voronoirt_amd/, libvrt_hip.so)
 * never links, imports or calls anything in oracle/.
 *
 * It is a literal, loop-for-loop restatement in plain C99 of the reference algorithm
 * (meudnaes/VoronoiRT, Julia), keeping every quirk.  Each function cites the reference
 * file:line it follows (paths relative to the reference checkout).
 *
 * PARITY STATUS, Voronoi path: "parity unpinned" by the reference's own tests -- the reference
 * holds no golden vector, known-answer test or fixture for the Voronoi path (SURVEY.md section
 * 4/8c), and no Julia toolchain exists in the build image to run it.  That part is pinned only
 * by analytic known answers that follow from the reference code itself and by an independent
 * second transcription (tests/test_oracle.py, oracle/pyref.py), and -- statistically -- by the
 * reference's committed searchlight rasters (beam centroid, width and peak of
 * data/searchlight_data/I_*_voronoi.npy; tests/test_searchlight_reference.py).
 * PARITY STATUS, regular-grid solver (end of this file) and the numerics it shares with the
 * Voronoi path (linear_weights, trapezoidal): PINNED by outputs of the reference itself -- the
 * data files data/searchlight_data/I_160_45_regular.npy and I_20_15_regular.npy are reproduced
 * to a few ulp (tests/test_regular.py).
 *
 * Memory layouts mirror Julia's column-major arrays so the same buffers can be handed to the
 * C-ABI product library:
 *   positions      (3, n)      -> pos[3*i + c], c = 0:z 1:x 2:y            (voronoi_utils.jl:8)
 *   neighbours     (n, D+1)    -> nbr[i + n*j], j = 0: count, j>=1: ids     (voronoi_utils.jl:9,60-61)
 *                                 ids are 1-based; ids <= 0 are walls (-5 bottom, -6 top)
 *   Delaunay_lines (3, D, n)   -> lines[c + 3*(j + D*i)], j 0-based slot    (voronoi_utils.jl:195,239)
 *   layers (reduced), perm     -> 1-based int64, exactly Julia's values
 *   S_lambda       (nlam, n)   -> S[l + nlam*i]  (wavelength fastest)        (lambda_iteration.jl:60)
 *
 * Floating-point contract: compiled with -O2 -ffp-contract=off (no FMA contraction), every
 * expression evaluated left to right as written in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

#define NBR(i, j) nbr[(i) + n * (j)] /* 0-based i, column j (0 = count) */

/* ------------------------------------------------------------------------------------------
 * read_cell, file-parsing part -- src/voronoi_utils.jl:42-63
 * One line per cell: "ID nb1 nb2 ... nbN" (voro++ "%i %n", rt_preprocessing/output_sites.cc:49).
 * M must be zero-initialised, n x (max_guess+1) column-major.  Returns 0 ok, <0 on error.
 * *max_nb receives maximum(NeighbourMatrix[:,1]) (voronoi_utils.jl:65).
 * ------------------------------------------------------------------------------------------ */
int orc_read_neighbours(const char *fname, i64 n, i64 max_guess, i64 *M, i64 *max_nb)
{
    FILE *f = fopen(fname, "r");
    if (!f) return -1;
    size_t cap = 1 << 16;
    char *line = (char *)malloc(cap);
    i64 mx = 0;
    int rc = 0;
    while (fgets(line, (int)cap, f)) {
        /* grow for very long lines */
        size_t len = strlen(line);
        while (len == cap - 1 && line[len - 1] != '\n') {
            cap *= 2;
            line = (char *)realloc(line, cap);
            if (!fgets(line + len, (int)(cap - len), f)) break;
            len = strlen(line);
        }
        char *p = line, *end;
        /* split(l): whitespace-separated tokens; first token = ID (voronoi_utils.jl:51) */
        long long id = strtoll(p, &end, 10);
        if (end == p) continue; /* blank line: enumerate(eachline) would fail in Julia; skip */
        p = end;
        if (id < 1 || id > n) { rc = -2; break; }
        i64 N = 0;
        for (;;) {
            long long v = strtoll(p, &end, 10);
            if (end == p) break;
            p = end;
            if (N >= max_guess) { rc = -3; break; } /* Julia: BoundsError */
            M[(id - 1) + n * (N + 1)] = (i64)v;    /* NeighbourMatrix[ID, 2:N+1] (:61) */
            N++;
        }
        if (rc) break;
        M[(id - 1)] = N;                            /* NeighbourMatrix[ID, 1] = N (:60) */
        if (N > mx) mx = N;
    }
    free(line);
    fclose(f);
    *max_nb = mx;
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * _sort_by_layer_up / _sort_by_layer_down -- src/voronoi_utils.jl:93-130 / :138-174
 * boundary = -5 (up, bottom wall) or -6 (down, top wall).  layers[n] out (1-based layer ids).
 * Literal level-synchronous scan.  Returns the number of layers, or -1 if a pass assigns
 * nothing while unassigned cells remain (the reference would loop forever).
 * ------------------------------------------------------------------------------------------ */
i64 orc_sort_by_layer(const i64 *nbr, i64 n, i64 boundary, i64 *layers)
{
    for (i64 i = 0; i < n; i++) layers[i] = 0;
    for (i64 i = 0; i < n; i++) {              /* :98-105 */
        i64 nn = NBR(i, 0);
        for (i64 j = 1; j <= nn; j++)
            if (NBR(i, j) == boundary) layers[i] = 1;
    }
    i64 lower = 1;
    for (;;) {                                 /* :108 while true */
        i64 assigned = 0;
        for (i64 i = 0; i < n; i++) {          /* :109 */
            if (layers[i] == 0) {
                i64 nn = NBR(i, 0);
                for (i64 j = 1; j <= nn; j++) {
                    i64 nb = NBR(i, j);
                    if (nb > 0 && layers[nb - 1] == lower) { /* :114 */
                        layers[i] = lower + 1;
                        assigned++;
                        break;
                    }
                }
            }
        }
        int any0 = 0;                          /* :122 */
        for (i64 i = 0; i < n; i++) if (layers[i] == 0) { any0 = 1; break; }
        if (!any0) break;
        if (assigned == 0) return -1;          /* reference: infinite loop */
        lower++;
    }
    i64 mx = 0;
    for (i64 i = 0; i < n; i++) if (layers[i] > mx) mx = layers[i];
    return mx;
}

/* sortperm(layers) -- src/voronoi_utils.jl:72,77.  Julia's default sortperm is stable
 * (ties broken by index), so a counting sort by layer reproduces it.  perm is 1-based. */
void orc_sortperm_stable(const i64 *layers, i64 n, i64 *perm)
{
    i64 mx = 0;
    for (i64 i = 0; i < n; i++) if (layers[i] > mx) mx = layers[i];
    i64 *cnt = (i64 *)calloc((size_t)mx + 2, sizeof(i64));
    for (i64 i = 0; i < n; i++) cnt[layers[i] + 1]++;
    for (i64 l = 1; l <= mx + 1; l++) cnt[l] += cnt[l - 1];
    for (i64 i = 0; i < n; i++) perm[cnt[layers[i]]++] = i + 1;
    free(cnt);
}

/* reduce_layers -- src/voronoi_utils.jl:253-269.  `sorted` = layers[perm] (ascending).
 * reduced has length maximum(layers)+1; reduced[end] = n (NOT n+1: the reference quirk that
 * leaves the last site of the sweep order unsolved).  Returns that length. */
i64 orc_reduce_layers(const i64 *sorted, i64 n, i64 *reduced)
{
    i64 mx = 0;
    for (i64 i = 0; i < n; i++) if (sorted[i] > mx) mx = sorted[i];
    i64 len = mx + 1;
    for (i64 i = 0; i < len; i++) reduced[i] = 0; /* Julia: undef */
    reduced[0] = 1;                                /* :256 */
    i64 layer = 2;
    for (i64 i = 0; i < n; i++) {                  /* :259 */
        if (sorted[i] == layer) {
            reduced[layer - 1] = i + 1;
            layer++;
        }
    }
    reduced[len - 1] = n;                          /* :266 */
    return len;
}

/* ------------------------------------------------------------------------------------------
 * calc_Delaunay_lines -- src/voronoi_utils.jl:186-245
 * Unit vectors site -> neighbour with the reference's periodic-image rule (shift to the
 * right, MIRROR to the left -- :219-222, :229-232).  Wall slots are left untouched.
 * norm(p_d) = sqrt((a*a + b*b) + c*c)  (LinearAlgebra.generic_norm2, unscaled branch).
 * ------------------------------------------------------------------------------------------ */
void orc_delaunay_lines(const double *pos, const i64 *nbr, i64 n, i64 D,
                        double x_min, double x_max, double y_min, double y_max, double *lines)
{
    for (i64 i = 0; i < n; i++) {
        const double pz = pos[3 * i + 0], px = pos[3 * i + 1], py = pos[3 * i + 2];
        const double x_r_r = x_max - px;  /* :200 */
        const double x_r_l = px - x_min;  /* :201 */
        const double y_r_r = y_max - py;  /* :203 */
        const double y_r_l = py - y_min;  /* :204 */
        i64 nn = NBR(i, 0);
        for (i64 j = 0; j < nn; j++) {
            i64 nb = NBR(i, j + 1);
            if (nb > 0) {
                double qz = pos[3 * (nb - 1) + 0];
                double qx = pos[3 * (nb - 1) + 1];
                double qy = pos[3 * (nb - 1) + 2];
                double x_i_r = fabs(x_max - qx);      /* :215 */
                double x_i_l = fabs(qx - x_min);      /* :216 */
                if (x_r_r + x_i_l < px - qx)          /* :219 */
                    qx = x_max + qx - x_min;          /* :220  (x_max + p) - x_min */
                else if (x_r_l + x_i_r < qx - px)     /* :221 */
                    qx = x_min + x_max - qx;          /* :222  mirror quirk */
                double y_i_r = fabs(y_max - qy);
                double y_i_l = fabs(qy - y_min);
                if (y_r_r + y_i_l < py - qy)          /* :229 */
                    qy = y_max + qy - y_min;
                else if (y_r_l + y_i_r < qy - py)     /* :231 */
                    qy = y_min + y_max - qy;
                double dz = qz - pz, dx = qx - px, dy = qy - py; /* :235 */
                double nrm = sqrt((dz * dz + dx * dx) + dy * dy); /* :237 */
                double *o = lines + 3 * (j + D * i);
                o[0] = dz / nrm;
                o[1] = dx / nrm;
                o[2] = dy / nrm;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * smallest_angle(n::Int, ...) -- src/voronoi_utils.jl:360-396
 * Order-dependent top-2 selection (a new best DISCARDS the old best, :378-386).
 * dot(k, line) = (k1*d1 + k2*d2) + k3*d3, no FMA (build-wide contract).
 * Returns 0, or -1 when no neighbour was ever stored in slot 1 (reference: uninitialised).
 * i is 0-based; idx[] are 1-based ids as in the reference.
 * ------------------------------------------------------------------------------------------ */
int orc_smallest_angle(i64 i, const i64 *nbr, i64 n, i64 D, const double *lines,
                       const double *k, double *dots, i64 *idx)
{
    dots[0] = -1.0; dots[1] = -1.0;  /* :365-366 */
    idx[0] = 0; idx[1] = 0;          /* :368 undef */
    i64 nn = NBR(i, 0);
    for (i64 j = 0; j < nn; j++) {
        i64 nb = NBR(i, j + 1);
        if (nb > 0) {                /* :371 */
            const double *d = lines + 3 * (j + D * i);
            double dp = (k[0] * d[0] + k[1] * d[1]) + k[2] * d[2]; /* :376 */
            if (dp > dots[1]) {      /* :378 */
                if (dp > dots[0]) {  /* :379 */
                    dots[0] = dp; idx[0] = nb;
                } else {
                    dots[1] = dp; idx[1] = nb;
                }
            }
        }
    }
    if (dots[1] <= 0) {              /* :390 */
        dots[1] = 0;
        idx[1] = idx[0];
    }
    return idx[0] > 0 ? 0 : -1;
}

/* linear_weights -- src/functions.jl:484-500.  Returns (alpha, beta, exp) = (a, b, e). */
void orc_linear_weights(double dtau, double *a, double *b, double *e)
{
    if (dtau < 5e-4) {
        *e = 1 - dtau + 0.5 * (dtau * dtau);      /* Δτ^2 == Δτ*Δτ (Base.literal_pow) */
        *a = dtau * (1.0 / 2 - dtau / 3);
        *b = dtau * (1.0 / 2 - dtau / 6);
    } else if (dtau > 50) {
        *e = 0.0;
        *a = 1 / dtau;
        *b = 1.0 - *a;
    } else {
        *e = exp(-dtau);
        *a = (1 - *e) / dtau - *e;
        *b = 1 - *a - *e;
    }
}

/* ------------------------------------------------------------------------------------------
 * Delaunay_upII / Delaunay_downII -- src/irregular_ray_tracing.jl:15-82 / :96-163
 * dir = +1: up (perm_up, ascending in-layer order :41); dir = -1: down (perm_down,
 * descending :122).  layers = reduced offsets (1-based, length nl), perm 1-based.
 * I0 has length layers[1]-1 (:31-33).  I (length n) is the output.
 * Returns 0, or -1 if smallest_angle found no upwind neighbour for a visited site.
 * ------------------------------------------------------------------------------------------ */
int orc_delaunay(int dir, const double *k, const double *S, const double *I0,
                 const double *alpha, const double *pos, const i64 *nbr, i64 n, i64 D,
                 const double *lines, const i64 *layers, i64 nl, const i64 *perm,
                 i64 n_sweeps, double *I)
{
    const double p = 7.0;                                  /* :1 */
    for (i64 i = 0; i < n; i++) I[i] = 0.0;                /* :23 */
    i64 lower = layers[1] - 1;                             /* :31 */
    for (i64 i = 0; i < lower; i++) I[perm[i] - 1] = I0[i]; /* :33 */
    int rc = 0;
    for (i64 layer = 2; layer <= nl - 1; layer++) {        /* :37 */
        i64 lo = layers[layer - 1];                        /* 1-based positions */
        i64 hi = layers[layer];
        for (i64 sweep = 0; sweep < n_sweeps; sweep++) {   /* :40 */
            i64 cnt = hi - lo;                             /* positions lo .. hi-1 */
            for (i64 t = 0; t < cnt; t++) {
                i64 posn = (dir > 0) ? (lo + t) : (hi - 1 - t); /* :41 / :122 */
                i64 idx = perm[posn - 1] - 1;              /* 0-based site */
                double dots[2]; i64 up[2];
                if (orc_smallest_angle(idx, nbr, n, D, lines, k, dots, up)) { rc = -1; continue; }
                double p1 = pow(dots[0], p), p2 = pow(dots[1], p);
                double sum = p1 + p2;                      /* sum(dot_products.^p) */
                double w[2] = { p1 / sum, p2 / sum };      /* :51 */
                I[idx] = 0.0;                              /* :53 */
                for (int rn = 0; rn < 2; rn++) {
                    i64 u = up[rn] - 1;
                    double a_c = alpha[idx], a_u = alpha[u];
                    double dz = pos[3 * idx + 0] - pos[3 * u + 0];
                    double dx = pos[3 * idx + 1] - pos[3 * u + 1];
                    double dy = pos[3 * idx + 2] - pos[3 * u + 2];
                    double r = sqrt((dz * dz + dx * dx) + dy * dy);   /* euclidean :66 (no wrap) */
                    double dtau = r * (a_c + a_u) / 2;                /* trapezoidal functions.jl:393 */
                    double a, b, e;
                    orc_linear_weights(dtau, &a, &b, &e);             /* :73 */
                    double S_c = S[idx], S_u = S[u], I_u = I[u];
                    I[idx] += ((e * I_u + a * S_u) + b * S_c) * w[rn]; /* :76 */
                }
            }
        }
    }
    return rc;
}

/* Upwind table for one direction vector: what the hot loop recomputes per visit
 * (irregular_ray_tracing.jl:50-51,66) hoisted per site.  For parity tests of the product's
 * per-angle table: up (1-based ids), dots, weights, path lengths.  All sites are evaluated
 * (also boundary ones); status[i] = 0 ok / -1 no upwind. */
void orc_upwind_table(const double *k, const double *pos, const i64 *nbr, i64 n, i64 D,
                      const double *lines, i64 *up /*2n*/, double *dots /*2n*/,
                      double *w /*2n*/, double *r /*2n*/, int32_t *status)
{
    for (i64 i = 0; i < n; i++) {
        double d[2]; i64 u[2];
        int rc = orc_smallest_angle(i, nbr, n, D, lines, k, d, u);
        status[i] = rc;
        up[2 * i] = u[0]; up[2 * i + 1] = u[1];
        dots[2 * i] = d[0]; dots[2 * i + 1] = d[1];
        if (rc) { w[2 * i] = w[2 * i + 1] = r[2 * i] = r[2 * i + 1] = 0; continue; }
        double p1 = pow(d[0], 7.0), p2 = pow(d[1], 7.0), sum = p1 + p2;
        w[2 * i] = p1 / sum; w[2 * i + 1] = p2 / sum;
        for (int rn = 0; rn < 2; rn++) {
            i64 q = u[rn] - 1;
            double dz = pos[3 * i + 0] - pos[3 * q + 0];
            double dx = pos[3 * i + 1] - pos[3 * q + 1];
            double dy = pos[3 * i + 2] - pos[3 * q + 2];
            r[2 * i + rn] = sqrt((dz * dz + dx * dx) + dy * dy);
        }
    }
}

/* k = [cos θ, cos ϕ sin θ, sin ϕ sin θ], degrees -> radians as θ*π/180
 * -- src/lambda_iteration.jl:87, src/lambda_continuum.jl:43 */
void orc_direction(double theta_deg, double phi_deg, double *k)
{
    const double pi = 3.14159265358979323846;
    double th = theta_deg * pi / 180, ph = phi_deg * pi / 180;
    k[0] = cos(th);
    k[1] = cos(ph) * sin(th);
    k[2] = sin(ph) * sin(th);
}

/* ------------------------------------------------------------------------------------------
 * J_λ_voronoi -- src/lambda_iteration.jl:60-113 (line, nλ>1) / src/lambda_continuum.jl:27-56.
 * Angles serial (:84), wavelengths split statically over `nthreads` threads (:91,
 * Threads.@threads), sites serial.  For every (angle, λ) the reference materialises the
 * contiguous vectors S_λ[l,:] and α_tot (:93-96,102) -- done here too.
 *   S      [n][nlam]  (λ fastest)
 *   alpha  alpha_mode 0: [n] (same for all λ and angles)
 *                     1: [n][nlam] (same for all angles)
 *                     2: [n_angles][n][nlam]
 *   I0_up  [n1_up][nlam] or NULL (zeros);  I0_down [n1_down][nlam] or NULL (zeros, :105-106)
 *   J      [n][nlam], overwritten with Σ_angles w·I (J_λ = zero(S_λ) :70)
 * θ == 90 is skipped (:98,104).  Returns 0 or -1.
 * ------------------------------------------------------------------------------------------ */
int orc_J_voronoi(i64 n_angles, const double *weights, const double *theta, const double *phi,
                  i64 nlam, const double *S, const double *alpha, int alpha_mode,
                  const double *I0_up, const double *I0_down,
                  const double *pos, const i64 *nbr, i64 n, i64 D, const double *lines,
                  const i64 *layers_up, i64 nl_up, const i64 *perm_up,
                  const i64 *layers_down, i64 nl_down, const i64 *perm_down,
                  i64 n_sweeps, int nthreads, double *J)
{
    for (i64 t = 0; t < n * nlam; t++) J[t] = 0.0;
    int rc_all = 0;
    if (nthreads < 1) nthreads = 1;
    i64 n1_up = layers_up[1] - 1, n1_down = layers_down[1] - 1;
    i64 n1_max = n1_up > n1_down ? n1_up : n1_down;
    for (i64 a = 0; a < n_angles; a++) {               /* :84 serial */
        double k[3];
        orc_direction(theta[a], phi[a], k);
        int up = theta[a] > 90, down = theta[a] < 90;
        if (!up && !down) continue;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
        for (i64 l = 0; l < nlam; l++) {                /* :91 Threads.@threads */
            double *Sl = (double *)malloc(sizeof(double) * (size_t)n);
            double *al = (double *)malloc(sizeof(double) * (size_t)n);
            double *Il = (double *)malloc(sizeof(double) * (size_t)n);
            double *I0 = (double *)calloc((size_t)(n1_max > 0 ? n1_max : 1), sizeof(double));
            for (i64 i = 0; i < n; i++) Sl[i] = S[l + nlam * i];
            if (alpha_mode == 0)      for (i64 i = 0; i < n; i++) al[i] = alpha[i];
            else if (alpha_mode == 1) for (i64 i = 0; i < n; i++) al[i] = alpha[l + nlam * i];
            else for (i64 i = 0; i < n; i++) al[i] = alpha[(size_t)a * n * nlam + l + nlam * i];
            int rc;
            if (up) {
                if (I0_up) for (i64 i = 0; i < n1_up; i++) I0[i] = I0_up[l + nlam * i];
                rc = orc_delaunay(+1, k, Sl, I0, al, pos, nbr, n, D, lines,
                                  layers_up, nl_up, perm_up, n_sweeps, Il);
            } else {
                if (I0_down) for (i64 i = 0; i < n1_down; i++) I0[i] = I0_down[l + nlam * i];
                rc = orc_delaunay(-1, k, Sl, I0, al, pos, nbr, n, D, lines,
                                  layers_down, nl_down, perm_down, n_sweeps, Il);
            }
            if (rc) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                rc_all = -1;
            }
            double w = weights[a];
            for (i64 i = 0; i < n; i++) J[l + nlam * i] += w * Il[i]; /* :102,107 */
            free(Sl); free(al); free(Il); free(I0);
        }
    }
    return rc_all;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ==========================================================================================
 * Regular-grid short characteristics -- src/characteristics.jl.
 * Literal restatement of short_characteristics_up/down and the six per-plane kernels, kept
 * with their quirks (ghost-zone refresh inside the sweep loop only in yz_up_ray :480-482;
 * xz_down_ray takes the centre values from the UPPER plane :794,804; the carried row
 * I_upper/I_lower is not reset between sweeps).
 * Arrays are Julia column-major: S, alpha, I (nz, nx, ny) -> a[iz + nz*(ix + nx*iy)];
 * I_0 and every plane (nx, ny) -> p[ix + nx*iy].  Indices below are 0-based.
 * PINNED by reference output: data/searchlight_data/I_160_45_regular.npy and
 * I_20_15_regular.npy (tests/test_regular.py).
 * ========================================================================================== */

/* bilinear -- src/functions.jl:332-355 */
static double orc_bilinear(double x_mrk, double y_mrk, double x1, double x2, double y1, double y2,
                           double Q11, double Q12, double Q21, double Q22)
{
    double dx = x2 - x1, dy = y2 - y1;
    double f1 = ((x2 - x_mrk) * Q11 + (x_mrk - x1) * Q21) / dx;
    double f2 = ((x2 - x_mrk) * Q12 + (x_mrk - x1) * Q22) / dx;
    return ((y2 - y_mrk) * f1 + (y_mrk - y1) * f2) / dy;
}

/* xy_intersect(k) -- src/functions.jl:430-457 */
void orc_xy_intersect(const double *k, int *sign_x, int *sign_y)
{
    if (k[1] > 0 && k[2] > 0) { *sign_x = -1; *sign_y = -1; }
    else if (k[1] < 0 && k[2] > 0) { *sign_x = 1; *sign_y = -1; }
    else if (k[1] < 0 && k[2] < 0) { *sign_x = 1; *sign_y = 1; }
    else if (k[1] > 0 && k[2] < 0) { *sign_x = -1; *sign_y = 1; }
    else { *sign_x = 1; *sign_y = 1; }
}

#define A3(a, iz, ix, iy) (a)[(iz) + nz * ((ix) + nx * (iy))]
#define P2(p, ix, iy) (p)[(ix) + nx * (iy)]

/* xy_up_ray :191-278 / xy_down_ray :288-372.  up: idz_upwind = idz-1; down: idz+1.
 * I0 = intensity of the upwind plane, I = new plane (nx*ny, zero-initialised here). */
static void orc_xy_ray(int up, const double *k, i64 idz, int sign_x, int sign_y, const double *I0,
                       const double *S, const double *alpha, const double *z, const double *x,
                       const double *y, i64 nz, i64 nx, i64 ny, double *I)
{
    for (i64 t = 0; t < nx * ny; t++) I[t] = 0.0;
    i64 idz_u = up ? idz - 1 : idz + 1;
    double dz = z[idz_u] - z[idz];
    double r = fabs(dz / k[0]);
    double x_inc = r * k[1], y_inc = r * k[2];
    int hx = (sign_x + 1) / 2, hy = (sign_y + 1) / 2;
    for (i64 idx = 1; idx <= nx - 2; idx++) {
        i64 xl = idx - hx, xu = xl + 1;
        double x_up = x[idx] + x_inc;
        for (i64 idy = 1; idy <= ny - 2; idy++) {
            i64 yl = idy - hy, yu = yl + 1;
            double y_up = y[idy] + y_inc;
            double a_c = A3(alpha, idz, idx, idy);
            double a_u = orc_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu],
                                      A3(alpha, idz_u, xl, yl), A3(alpha, idz_u, xl, yu),
                                      A3(alpha, idz_u, xu, yl), A3(alpha, idz_u, xu, yu));
            double dtau = r * (a_c + a_u) / 2;
            double S_c = A3(S, idz, idx, idy);
            double S_u = orc_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu],
                                      A3(S, idz_u, xl, yl), A3(S, idz_u, xl, yu),
                                      A3(S, idz_u, xu, yl), A3(S, idz_u, xu, yu));
            double a, b, e;
            orc_linear_weights(dtau, &a, &b, &e);
            double I_u = orc_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu],
                                      P2(I0, xl, yl), P2(I0, xl, yu), P2(I0, xu, yl), P2(I0, xu, yu));
            P2(I, idx, idy) = (e * I_u + a * S_u) + b * S_c;
        }
        P2(I, idx, 0) = P2(I, idx, ny - 2);       /* ghost zones :270-271 */
        P2(I, idx, ny - 1) = P2(I, idx, 1);
    }
    for (i64 idy = 0; idy < ny; idy++) {          /* :274-275 */
        P2(I, 0, idy) = P2(I, nx - 2, idy);
        P2(I, nx - 1, idy) = P2(I, 1, idy);
    }
}

/* yz_up_ray :383-487 / yz_down_ray :497-604: upwind point on the x = x[idx + sign_x] plane. */
static void orc_yz_ray(int up, const double *k, i64 idz, int sign_x, int sign_y, const double *I0,
                       const double *S, const double *alpha, const double *z, const double *x,
                       const double *y, i64 nz, i64 nx, i64 ny, i64 n_sweeps, double *I)
{
    for (i64 t = 0; t < nx * ny; t++) I[t] = 0.0;
    double *I_row = (double *)calloc((size_t)ny, sizeof(double));  /* I_upper / I_lower :399,514 */
    double dx = x[1] - x[0];
    i64 sx0 = sign_x == 1 ? 1 : nx - 2, sx1 = sign_x == 1 ? nx - 2 : 1;   /* range_bounds */
    i64 sy0 = sign_y == 1 ? 1 : ny - 2, sy1 = sign_y == 1 ? ny - 2 : 1;
    double z_c = z[idz];
    i64 idz_o = up ? idz - 1 : idz + 1;            /* the other plane of the z interval */
    double r = fabs(dx / k[1]);
    double z_inc = r * k[0], y_inc = r * k[2];
    double z_up = z_c + z_inc;
    double zb1 = up ? z[idz_o] : z_c, zb2 = up ? z_c : z[idz_o];      /* z_bounds */
    i64 iz_lo = up ? idz_o : idz, iz_hi = up ? idz : idz_o;             /* α_lower / α_upper planes */
    int hy = (sign_y + 1) / 2;
    for (i64 sweep = 0; sweep < n_sweeps; sweep++) {
        for (i64 idx = sx0; sign_x == 1 ? idx <= sx1 : idx >= sx1; idx += sign_x) {
            i64 xu = idx + sign_x;                 /* idx_upwind */
            for (i64 idy = sy0; sign_y == 1 ? idy <= sy1 : idy >= sy1; idy += sign_y) {
                i64 yl = idy - hy, yu = yl + 1;
                double y_up = y[idy] + y_inc;
                double a_c = A3(alpha, idz, idx, idy);   /* α_upper[idx,idy] (up) / α_lower (down): both = plane idz */
                double a_u = orc_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu],
                                          A3(alpha, iz_lo, xu, yl), A3(alpha, iz_lo, xu, yu),
                                          A3(alpha, iz_hi, xu, yl), A3(alpha, iz_hi, xu, yu));
                double dtau = r * (a_c + a_u) / 2;
                double S_c = A3(S, idz, idx, idy);
                double S_u = orc_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu],
                                          A3(S, iz_lo, xu, yl), A3(S, iz_lo, xu, yu),
                                          A3(S, iz_hi, xu, yl), A3(S, iz_hi, xu, yu));
                double a, b, e;
                orc_linear_weights(dtau, &a, &b, &e);
                double I_u;
                if (up)     /* rows: [I_0 at the lower plane ; carried row at z_centre] :458-459 */
                    I_u = orc_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu],
                                       P2(I0, xu, yl), P2(I0, xu, yu), I_row[yl], I_row[yu]);
                else        /* rows: [carried row at z_centre ; I_0 at the upper plane] :577-578 */
                    I_u = orc_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu],
                                       I_row[yl], I_row[yu], P2(I0, xu, yl), P2(I0, xu, yu));
                P2(I, idx, idy) = (e * I_u + a * S_u) + b * S_c;
            }
            P2(I, idx, 0) = P2(I, idx, ny - 2);
            P2(I, idx, ny - 1) = P2(I, idx, 1);
            for (i64 j = 0; j < ny; j++) I_row[j] = P2(I, idx, j);   /* I_upper = I[idx, :] (copy) */
        }
        if (up)                                     /* yz_up_ray: inside the sweep loop :480-482 */
            for (i64 idy = 0; idy < ny; idy++) {
                P2(I, 0, idy) = P2(I, nx - 2, idy);
                P2(I, nx - 1, idy) = P2(I, 1, idy);
            }
    }
    if (!up)                                        /* yz_down_ray: after the sweeps :599-601 */
        for (i64 idy = 0; idy < ny; idy++) {
            P2(I, 0, idy) = P2(I, nx - 2, idy);
            P2(I, nx - 1, idy) = P2(I, 1, idy);
        }
    free(I_row);
}

/* xz_up_ray :614-716 / xz_down_ray :726-835: upwind point on the y = y[idy + sign_y] plane. */
static void orc_xz_ray(int up, const double *k, i64 idz, int sign_x, int sign_y, const double *I0,
                       const double *S, const double *alpha, const double *z, const double *x,
                       const double *y, i64 nz, i64 nx, i64 ny, i64 n_sweeps, double *I)
{
    for (i64 t = 0; t < nx * ny; t++) I[t] = 0.0;
    /* the reference allocates zero(I_0[end,:]) (length ny) and then assigns I[:, idy] (length
     * nx); before the first assignment only zeros are read, so nx zeros are equivalent when
     * nx == ny (the only case the reference's square grids exercise) */
    double *I_col = (double *)calloc((size_t)(nx > ny ? nx : ny), sizeof(double));
    double dy = y[1] - y[0];
    i64 sx0 = sign_x == 1 ? 1 : nx - 2, sx1 = sign_x == 1 ? nx - 2 : 1;
    i64 sy0 = sign_y == 1 ? 1 : ny - 2, sy1 = sign_y == 1 ? ny - 2 : 1;
    double z_c = z[idz];
    i64 idz_o = up ? idz - 1 : idz + 1;
    double r = fabs(dy / k[2]);
    double z_inc = r * k[0], x_inc = r * k[1];
    double z_up = z_c + z_inc;
    double zb1 = up ? z[idz_o] : z_c, zb2 = up ? z_c : z[idz_o];
    i64 iz_lo = up ? idz_o : idz, iz_hi = up ? idz : idz_o;
    /* centre values: α_upper[idx, idy] in BOTH variants (:672 and :794) -- for the down ray that
     * is the plane idz+1, not the plane being solved (reference quirk) */
    i64 iz_c = iz_hi;
    int hx = (sign_x + 1) / 2;
    for (i64 sweep = 0; sweep < n_sweeps; sweep++) {
        for (i64 idy = sy0; sign_y == 1 ? idy <= sy1 : idy >= sy1; idy += sign_y) {
            i64 yu = idy + sign_y;                 /* idy_upwind */
            for (i64 idx = sx0; sign_x == 1 ? idx <= sx1 : idx >= sx1; idx += sign_x) {
                i64 xl = idx - hx, xu = xl + 1;
                double x_up = x[idx] + x_inc;
                double a_c = A3(alpha, iz_c, idx, idy);
                double a_u = orc_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu],
                                          A3(alpha, iz_lo, xl, yu), A3(alpha, iz_lo, xu, yu),
                                          A3(alpha, iz_hi, xl, yu), A3(alpha, iz_hi, xu, yu));
                double dtau = r * (a_c + a_u) / 2;
                double S_c = A3(S, iz_c, idx, idy);
                double S_u = orc_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu],
                                          A3(S, iz_lo, xl, yu), A3(S, iz_lo, xu, yu),
                                          A3(S, iz_hi, xl, yu), A3(S, iz_hi, xu, yu));
                double a, b, e;
                orc_linear_weights(dtau, &a, &b, &e);
                double I_u;
                if (up)
                    I_u = orc_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu],
                                       P2(I0, xl, yu), P2(I0, xu, yu), I_col[xl], I_col[xu]);
                else
                    I_u = orc_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu],
                                       I_col[xl], I_col[xu], P2(I0, xl, yu), P2(I0, xu, yu));
                P2(I, idx, idy) = (e * I_u + a * S_u) + b * S_c;
            }
            P2(I, 0, idy) = P2(I, nx - 2, idy);    /* :704-705 / :822-823 */
            P2(I, nx - 1, idy) = P2(I, 1, idy);
            for (i64 i = 0; i < nx; i++) I_col[i] = P2(I, i, idy);   /* I_upper = I[:, idy] */
        }
    }
    for (i64 idx = 0; idx < nx; idx++) {           /* after the sweeps :713-714 / :831-832 */
        P2(I, idx, 0) = P2(I, idx, ny - 2);
        P2(I, idx, ny - 1) = P2(I, idx, 1);
    }
    free(I_col);
}

/* short_characteristics_up :19-95 (up = 1) / short_characteristics_down :110-180 (up = 0).
 * plane_kind (optional, nz ints) receives 1/2/3 = xy/yz/xz per solved plane (0 for the boundary). */
int orc_short_characteristics(int up, const double *k, const double *S, const double *I0,
                              const double *alpha, const double *z, const double *x,
                              const double *y, i64 nz, i64 nx, i64 ny, i64 n_sweeps, double *I,
                              int *plane_kind)
{
    if (nz < 2 || nx < 3 || ny < 3) return -1;
    for (i64 t = 0; t < nz * nx * ny; t++) I[t] = 0.0;
    double dx = x[1] - x[0], dy = y[1] - y[0];
    double r_x = fabs(dx / k[1]), r_y = fabs(dy / k[2]);
    int sign_x, sign_y;
    orc_xy_intersect(k, &sign_x, &sign_y);
    double *prev = (double *)malloc(sizeof(double) * (size_t)(nx * ny));
    double *cur = (double *)malloc(sizeof(double) * (size_t)(nx * ny));
    i64 iz_b = up ? 0 : nz - 1;
    for (i64 ix = 0; ix < nx; ix++)
        for (i64 iy = 0; iy < ny; iy++) A3(I, iz_b, ix, iy) = P2(I0, ix, iy);   /* :61 / :146 */
    if (plane_kind) for (i64 t = 0; t < nz; t++) plane_kind[t] = 0;
    for (i64 s = 1; s < nz; s++) {
        i64 idz = up ? s : nz - 1 - s;
        i64 idz_u = up ? idz - 1 : idz + 1;
        double dz = up ? z[idz] - z[idz - 1] : z[idz + 1] - z[idz];
        double r_z = fabs(dz / k[0]);
        int cut = 1;                                 /* argmin([r_z, r_x, r_y]): first minimum */
        double m = r_z;
        if (r_x < m) { m = r_x; cut = 2; }
        if (r_y < m) { m = r_y; cut = 3; }
        for (i64 ix = 0; ix < nx; ix++)
            for (i64 iy = 0; iy < ny; iy++) P2(prev, ix, iy) = A3(I, idz_u, ix, iy);
        if (cut == 1) orc_xy_ray(up, k, idz, sign_x, sign_y, prev, S, alpha, z, x, y, nz, nx, ny, cur);
        else if (cut == 2) orc_yz_ray(up, k, idz, sign_x, sign_y, prev, S, alpha, z, x, y, nz, nx, ny, n_sweeps, cur);
        else orc_xz_ray(up, k, idz, sign_x, sign_y, prev, S, alpha, z, x, y, nz, nx, ny, n_sweeps, cur);
        if (plane_kind) plane_kind[idz] = cut;
        for (i64 ix = 0; ix < nx; ix++)
            for (i64 iy = 0; iy < ny; iy++) A3(I, idz, ix, iy) = P2(cur, ix, iy);
    }
    free(prev);
    free(cur);
    return 0;
}
