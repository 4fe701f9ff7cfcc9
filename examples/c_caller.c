/* Plain-C caller of libvrt_hip.so: plays the role of the reference's Julia host (which cannot be
 * run in the build image) -- builds a tiny periodic lattice grid, runs Delaunay_upII through the
 * C ABI and prints the intensities.  Build:
 *   gcc -std=c99 -I include examples/c_caller.c -o c_caller -L voronoirt_amd -lvrt_hip \
 *       -Wl,-rpath,$PWD/voronoirt_amd -lm
 * Needs a HIP device at run time (the library has no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "voronoirt.h"

int main(void)
{
    /* 3 x 3 x 4 simple-cubic lattice in the unit cube, 6 neighbours per site, x/y periodic,
     * walls -5 (bottom) / -6 (top): the reference's conventions (voronoi_utils.jl:97,141) */
    enum { NX = 3, NY = 3, NZ = 4, N = NX * NY * NZ, D1 = 7 };
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
