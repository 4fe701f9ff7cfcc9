/* Plain-C caller of libvrt_hip.so: plays the role of the reference's Julia host (which cannot be
 * run in the build image).
 *   scenario 1 (default):   a tiny periodic lattice grid, Delaunay_upII through the C ABI, prints
 *                           the intensities -- the direct call sites of compare_searchlight.jl
 *   scenario 2 (argv[1]=2): exactly the caller julia/VoronoiRT_hip.jl's J_line is -- the body of
 *                           J_λ_voronoi (src/lambda_iteration.jl:60-113) with the formal solves
 *                           batched: ul7n12's 12 angles, 5 wavelengths, alpha_tot per (λ, site,
 *                           angle), I_0 of the bottom layer for the up rays, ONE
 *                           vrt_plan_execute, prints J (nλ, n)
 *   scenario 3 (argv[1]=3, argv[2]=inputs file): the caller julia/VoronoiRT_hip.jl's Λ_voronoi is -- the
 *                           reference's Λ_voronoi loop (src/lambda_iteration.jl:205-300) over vrt_lambda_create /
 *                           _iterate / _get with HOST arrays: per iteration only the criterion's scalar
 *                           comes back; prints the history, J, S_new and the populations.  The per-site
 *                           inputs (what Λ_voronoi derives before its loop) are read from a binary file
 *                           the test writes, as the Julia driver would hand over its arrays.
 *   scenario 4 (argv[1]=4, the same inputs file with head[3] = number of device handles): that loop across
 *                           several GPUs of the node (vrt_multi_create + vrt_multi_lambda_*): wavelength blocks per
 *                           device, ONE all-reduce of the rate-integral shares per iteration; devices from
 *                           VRT_DEVICES ("0,1,2,3"), else head[3] handles on device 0 (one-GPU rehearsal).
 * Build:
 *   gcc -std=c99 -I include examples/c_caller.c -o c_caller -L voronoirt_amd -lvrt_hip \
 *       -Wl,-rpath,$PWD/voronoirt_amd -lm
 * Needs a HIP device at run time (the library has no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "voronoirt.h"

/* quadratures/ul7n12.dat: weight, θ [deg], ϕ [deg] */
static const double UL7N12[12][3] = {
    {0.062174023651822, 70.292581108446825, 346.412955051617416},
    {0.062174023651822, 109.707418891553175, 193.587044948382584},
    {0.078304613457687, 152.666292044518485, 315.475247829748128},
    {0.078304613457687, 27.333707955481518, 135.475247829748128},
    {0.090740740740741, 147.207528953818269, 135.743688985642649},
    {0.090740740740741, 67.175739518129632, 155.790538127899197},
    {0.090740740740741, 32.792471046181731, 44.256311014357351},
    {0.090740740740741, 112.824260481870382, 335.790538127899197},
    {0.084923207761833, 101.810709392034880, 235.428463450411130},
    {0.084923207761833, 78.189290607965106, 55.428463450411122},
    {0.093116673647177, 65.132900950498197, 260.165664821292125},
    {0.093116673647177, 114.867099049501803, 80.165664821292154}};

/* the J_λ_voronoi caller: all angles x all wavelengths in one batched execute */
static int scenario_J(vrt_grid *g, int n)
{
    enum { NA = 12, NLAM = 5 };
    double k[3 * NA], w[NA];
    int dirs[NA];
    for (int a = 0; a < NA; a++) {
        w[a] = UL7N12[a][0];
        vrt_direction(UL7N12[a][1], UL7N12[a][2], k + 3 * a);     /* lambda_iteration.jl:87 */
        dirs[a] = UL7N12[a][1] > 90 ? 1 : (UL7N12[a][1] < 90 ? -1 : 0);   /* :98,104 */
    }
    vrt_plan *plan = NULL;
    if (vrt_plan_create_ex(g, NA, k, dirs, 3, &plan)) {
        fprintf(stderr, "vrt_plan_create_ex: %s\n", vrt_last_error());
        return 1;
    }
    int64_t nl = vrt_grid_num_layer_offsets(g, +1);
    int64_t *layers = malloc(sizeof(int64_t) * (size_t)nl);
    vrt_grid_get_layers(g, +1, layers);
    const int64_t n1 = layers[1] - 1;
    double *S = malloc(sizeof(double) * NLAM * (size_t)n), *J = malloc(sizeof(double) * NLAM * (size_t)n);
    double *alpha = malloc(sizeof(double) * NLAM * (size_t)n * NA), *I0 = malloc(sizeof(double) * NLAM * (size_t)n1);
    for (int i = 0; i < n; i++)
        for (int l = 0; l < NLAM; l++) {
            S[l + NLAM * i] = 1.0 + 0.1 * ((i * 7 + l * 3) % 11);          /* S_λ (nλ, n) */
            for (int a = 0; a < NA; a++)                                      /* α_tot (nλ, n, n_angles), :93-96 */
                alpha[l + NLAM * (i + (size_t)n * a)] = 0.5 + 0.05 * ((i + 2 * l + 3 * a) % 13);
        }
    for (int64_t p = 0; p < n1; p++)
        for (int l = 0; l < NLAM; l++) I0[l + NLAM * p] = 2.0 + 0.25 * ((p + l) % 5);   /* B_λ(T) of perm_up[1:n1], :99-101 */
    int rc = vrt_plan_execute(plan, NLAM, NLAM, S, alpha, VRT_ALPHA_ANGLE_SITE_LAM, I0, NULL, w, J, NULL);
    if (rc) fprintf(stderr, "vrt_plan_execute: %s\n", vrt_last_error());
    else
        for (int i = 0; i < n; i++)
            for (int l = 0; l < NLAM; l++) printf("%d %d %.17g\n", i + 1, l + 1, J[l + NLAM * i]);
    free(S); free(J); free(alpha); free(I0); free(layers);
    vrt_plan_destroy(plan);
    return rc ? 1 : 0;
}

/* the Λ_voronoi caller: library-owned device state, one call per iteration */
static double *rd(FILE *f, size_t count)
{
    double *a = malloc(sizeof(double) * (count ? count : 1));
    if (fread(a, sizeof(double), count, f) != count) { fprintf(stderr, "short inputs file\n"); exit(3); }
    return a;
}

static int scenario_lambda(vrt_grid *g, int n, const char *path, const double *pos, const int64_t *nbr, int64_t D1,
                           const double *bounds)
{
    enum { NA = 12 };
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path); return 3; }
    int64_t head[4], blocks[6];
    double sc[9];
    if (fread(head, sizeof(int64_t), 4, f) != 4 || fread(blocks, sizeof(int64_t), 6, f) != 6 ||
        fread(sc, sizeof(double), 9, f) != 9 || head[0] != n) {
        fprintf(stderr, "bad inputs file\n");
        return 3;
    }
    const int64_t nlam = head[1], maxiter = head[2];
    const double eps_conv = 0.0;
    vrt_line_case lc;
    lc.nlam = nlam;
    for (int q = 0; q < 6; q++) lc.blocks[q] = blocks[q];
    lc.lambda0 = sc[0]; lc.c0 = sc[1]; lc.strength_const = sc[2]; lc.Bij = sc[3]; lc.Bji = sc[4];
    lc.sigma_bb_const = sc[5]; lc.hc_over_kB = sc[6]; lc.pref_ij = sc[7]; lc.pref_ji = sc[8];
    lc.lambda = rd(f, (size_t)nlam);
    lc.velocity = rd(f, 3 * (size_t)n);
    lc.doppler_width = rd(f, (size_t)n);
    lc.gamma_static = rd(f, (size_t)n);
    lc.gamma_unsold = rd(f, (size_t)n);
    lc.alpha_cont = rd(f, (size_t)n);
    lc.eps = rd(f, (size_t)n);
    lc.temperature = rd(f, (size_t)n);
    lc.atom_density = rd(f, (size_t)n);
    lc.B0 = rd(f, (size_t)nlam * (size_t)n);
    lc.lte_populations = rd(f, 3 * (size_t)n);
    lc.C = rd(f, 9 * (size_t)n);
    lc.planck2 = rd(f, (size_t)nlam);
    lc.sigma_bf1 = rd(f, (size_t)(blocks[3] - blocks[2]));
    lc.sigma_bf2 = rd(f, (size_t)(blocks[5] - blocks[4]));
    fclose(f);
    double k[3 * NA], w[NA];
    int dirs[NA];
    for (int a = 0; a < NA; a++) {
        w[a] = UL7N12[a][0];
        vrt_direction(UL7N12[a][1], UL7N12[a][2], k + 3 * a);
        dirs[a] = UL7N12[a][1] > 90 ? 1 : (UL7N12[a][1] < 90 ? -1 : 0);
    }
    double *J = malloc(sizeof(double) * (size_t)nlam * (size_t)n), *S = malloc(sizeof(double) * (size_t)nlam * (size_t)n);
    double *P = malloc(sizeof(double) * 3 * (size_t)n);
    double diff = 1.0;
    int64_t i = 0;
    if (head[3] > 0) {
        /* scenario 4: the same loop across several devices of the node (vrt_multi_lambda_*): wavelength blocks per
         * device, per iteration ONE all-reduce of the rate-integral shares.  Devices from VRT_DEVICES ("0,1,2,3"),
         * else head[3] handles on device 0 (a rehearsal on a one-GPU box) */
        int devices[64], nd = 0;
        const char *env = getenv("VRT_DEVICES");
        if (env && *env)
            for (const char *q = env; *q && nd < 64;) {
                devices[nd++] = atoi(q);
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
        else
            for (; nd < head[3] && nd < 64; nd++) devices[nd] = 0;
        vrt_multi *mm = NULL;
        vrt_multi_lambda *ms = NULL;
        if (vrt_multi_create(nd, devices, n, pos, nbr, D1, bounds, NA, k, dirs, 3, &mm) || vrt_multi_lambda_create(mm, &lc, w, &ms)) {
            fprintf(stderr, "vrt_multi_lambda_create: %s\n", vrt_last_error());
            return 1;
        }
        while (diff > eps_conv && i < maxiter) {
            if (vrt_multi_lambda_iterate(ms, &diff)) {
                fprintf(stderr, "vrt_multi_lambda_iterate: %s\n", vrt_last_error());
                return 1;
            }
            printf("hist %lld %.17g\n", (long long)(i + 1), diff);
            i++;
        }
        if (vrt_multi_lambda_get(ms, J, S, P, NULL, NULL)) {
            fprintf(stderr, "vrt_multi_lambda_get: %s\n", vrt_last_error());
            return 1;
        }
        vrt_multi_lambda_destroy(ms);
        vrt_multi_destroy(mm);
    } else {
    vrt_plan *plan = NULL;
    vrt_lambda *ses = NULL;
    if (vrt_plan_create_ex(g, NA, k, dirs, 3, &plan) || vrt_lambda_create(plan, &lc, w, &ses)) {
        fprintf(stderr, "vrt_lambda_create: %s\n", vrt_last_error());
        return 1;
    }
    /* Λ_voronoi: while criterion(S_new, S_old, ϵ, i, maxiter) ... (lambda_iteration.jl:253); the criterion starts at 1 */
    while (diff > eps_conv && i < maxiter) {
        if (vrt_lambda_iterate(ses, &diff)) {
            fprintf(stderr, "vrt_lambda_iterate: %s\n", vrt_last_error());
            return 1;
        }
        printf("hist %lld %.17g\n", (long long)(i + 1), diff);
        i++;
    }
    if (vrt_lambda_get(ses, J, S, P, NULL, NULL)) {
        fprintf(stderr, "vrt_lambda_get: %s\n", vrt_last_error());
        return 1;
    }
    vrt_lambda_destroy(ses);
    vrt_plan_destroy(plan);
    }
    for (int s = 0; s < n; s++)
        for (int64_t l = 0; l < nlam; l++)
            printf("JS %d %lld %.17g %.17g\n", s + 1, (long long)(l + 1), J[l + nlam * s], S[l + nlam * s]);
    for (int s = 0; s < n; s++) printf("P %d %.17g %.17g %.17g\n", s + 1, P[s], P[s + n], P[s + 2 * n]);
    return 0;
}

int main(int argc, char **argv)
{
    /* 4 x 5 x 6 simple-cubic lattice in the unit cube, 6 neighbours per site, x/y periodic,
     * walls -5 (bottom) / -6 (top): the reference's conventions (voronoi_utils.jl:97,141) */
    enum { NX = 4, NY = 5, NZ = 6, N = NX * NY * NZ, D1 = 7 };
    static double pos[3 * N];
    static int64_t nbr[N * D1];
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NY; j++)
            for (int k = 0; k < NZ; k++) {
                int s = (i * NY + j) * NZ + k;
                pos[3 * s + 0] = (k + 0.5) / NZ;
                pos[3 * s + 1] = (i + 0.5) / NX;
                pos[3 * s + 2] = (j + 0.5) / NY;
                int64_t row[6] = {
                    ((i + 1) % NX * NY + j) * NZ + k + 1, ((i + NX - 1) % NX * NY + j) * NZ + k + 1,
                    (i * NY + (j + 1) % NY) * NZ + k + 1, (i * NY + (j + NY - 1) % NY) * NZ + k + 1,
                    k + 1 < NZ ? (int64_t)s + 2 : -6, k > 0 ? (int64_t)s : -5};
                nbr[s] = 6;                                  /* column 0: count */
                for (int c = 0; c < 6; c++) nbr[s + N * (c + 1)] = row[c];
            }
    const double bounds[6] = {0, 1, 0, 1, 0, 1};
    if (vrt_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", "libvrt_hip has no CPU fallback");
        return 2;
    }
    vrt_grid *g = NULL;
    if (vrt_grid_create(N, pos, nbr, D1, bounds, 0, &g)) {
        fprintf(stderr, "vrt_grid_create: %s\n", vrt_last_error());
        return 1;
    }
    if (argc > 1 && atoi(argv[1]) == 2) {
        int rc = scenario_J(g, N);
        vrt_grid_destroy(g);
        return rc;
    }
    if (argc > 2 && (atoi(argv[1]) == 3 || atoi(argv[1]) == 4)) {      /* 4: head[3] of the inputs file > 0 -> several devices */
        int rc = scenario_lambda(g, N, argv[2], pos, nbr, D1, bounds);
        vrt_grid_destroy(g);
        return rc;
    }
    int64_t nl = vrt_grid_num_layer_offsets(g, +1);
    int64_t *layers = malloc(sizeof(int64_t) * (size_t)nl);
    vrt_grid_get_layers(g, +1, layers);
    int64_t n1 = layers[1] - 1;                              /* sites that receive I_0 */
    double k[3], S[N], alpha[N], I[N], *I0 = malloc(sizeof(double) * (size_t)n1);
    vrt_direction(150.0, 30.0, k);
    for (int s = 0; s < N; s++) { S[s] = 1.0; alpha[s] = 2.0; }
    for (int64_t p = 0; p < n1; p++) I0[p] = 3.0;
    if (vrt_delaunay_up(g, k, S, I0, n1, alpha, 3, I)) {
        fprintf(stderr, "vrt_delaunay_up: %s\n", vrt_last_error());
        return 1;
    }
    for (int s = 0; s < N; s++) printf("%d %.17g\n", s + 1, I[s]);
    free(I0);
    free(layers);
    vrt_grid_destroy(g);
    return 0;
}
