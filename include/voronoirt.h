/*
 * voronoirt.h -- C ABI of libvrt_hip.so, the MI355X (gfx950) formal solver that drops in behind
 * VoronoiRT's Julia driver surface.
 *
 * The reference (meudnaes/VoronoiRT) has no FFI of its own: the boundary is the pair of Julia
 * methods `Delaunay_upII` / `Delaunay_downII` (src/irregular_ray_tracing.jl:15-82, :96-163),
 * their caller `J_λ_voronoi` (src/lambda_iteration.jl:60-113, src/lambda_continuum.jl:27-56) and
 * the grid constructor `read_cell` (src/voronoi_utils.jl:36-85).  A maintainer redefines those
 * methods to `ccall` the entry points below (INTEGRATION.md shows the Julia shim).
 *
 * Conventions (identical to the reference's Julia arrays, so Julia buffers pass unconverted):
 *   - all arrays column-major as Julia stores them, ids 1-based, int64 (`Int`), float64
 *   - positions  (3, n): pos[3*i + c], c = 0:z 1:x 2:y                 voronoi_utils.jl:8
 *   - neighbours (n, D1): nbr[i + n*j], column 0 = count, columns 1.. = ids; ids <= 0 are walls
 *     (-5 = z_min / bottom, -6 = z_max / top)                          voronoi_utils.jl:9,60-61,97,141
 *   - S, alpha, J (nlam, n): x[l + ld*i], wavelength fastest, ld >= nlam   lambda_iteration.jl:60-70
 *   - bounds[6] = z_min, z_max, x_min, x_max, y_min, y_max              io.jl:122-124
 * Every function returns 0 on success and a negative VRT_E* code on failure; it never throws
 * or aborts across the boundary.  vrt_last_error() returns a thread-local message.
 * Host entry points take host pointers and copy through PCIe; *_dev entry points take device
 * pointers (hipMalloc'ed or torch tensors) and a hipStream_t passed as void*.
 * There is NO CPU fallback: without a usable HIP device every compute call fails with
 * VRT_ENODEVICE.
 */
#ifndef VORONOIRT_H
#define VORONOIRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_OK 0
#define VRT_EINVAL (-1)     /* bad argument (null pointer, size mismatch, |k| != 1, ...)      */
#define VRT_EGRID (-2)      /* malformed grid: id out of range, unreachable site, D >= cap ... */
#define VRT_ENODEVICE (-3)  /* no HIP device / HIP runtime error                               */
#define VRT_ENOMEM (-4)
#define VRT_EIO (-5)        /* neighbour file unreadable / unparsable                          */

typedef struct vrt_grid vrt_grid;   /* replaces struct VoronoiSites, voronoi_utils.jl:7-28 */
typedef struct vrt_plan vrt_plan;   /* per-(grid, angle set) upwind tables + sweep schedule  */

/* alpha layouts accepted by the batched entry points */
#define VRT_ALPHA_SITE 0        /* alpha[n]                 same for every wavelength and angle  */
#define VRT_ALPHA_SITE_LAM 1    /* alpha[n][ld]             same for every angle (continuum)      */
#define VRT_ALPHA_ANGLE_SITE_LAM 2 /* alpha[n_angles][n][ld] per angle (line: lambda_iteration.jl:89-96) */
/* per angle, already in the library's NATIVE layout (device entry points only): for every angle a
 * block of vrt_plan_native_alpha_count / n_angles doubles holding wavelength PAIRS (an odd nlam is
 * padded with one finite wavelength), B = vrt_plan_native_pair_block(p) pairs of a site side by
 * side: the pairs form blocks of B, then (B not dividing the pair count) one block per set bit of
 * the remainder, widest first; pair q of a block [q0, q0 + w) at storage position pos is pair
 * element q0 * n + pos * w + (q - q0), each element two values (wavelengths 2q, 2q + 1); pos = the
 * site's position in the storage order of the angle's direction (vrt_grid_get_storage_order).
 * B = 1: element (l, pos) at ((l/2) * n + pos) * 2 + l%2.  Written by vrt_plan_alpha_to_native_dev
 * or vrt_line_opacity_dev -- callers need not know the layout; it saves the 2 x 8 B per (site,
 * angle, wavelength) of the layout change on every execute. */
#define VRT_ALPHA_ANGLE_NATIVE 3
/* alpha per (site, wavelength), the same for every angle (VRT_ALPHA_SITE_LAM: the continuum of
 * src/lambda_continuum.jl:27-56, where it does not change from one Λ-iteration to the next), handed over in SWEEP ORDER:
 * two plane sets of vrt_plan_native_plane_count(p, nlam) doubles one behind the other -- the up directions' order, then
 * the down directions' -- as vrt_plan_to_native_dev(p, nlam, ld, alpha, buf, buf + count, ...) writes them, once.
 * Accepted by vrt_plan_execute_native_dev[_f32] only (floats and vrt_plan_to_native_dev_f32 with the float entry): no layout
 * change of any kind is then left inside a step. */
#define VRT_ALPHA_SITE_LAM_NATIVE 4

const char *vrt_last_error(void);
int vrt_version(void);
/* number of visible HIP devices (0 if none); never fails */
int vrt_device_count(void);

/* ---- grid: replaces read_cell, src/voronoi_utils.jl:36-85 ---------------------------------
 * Builds BFS layers from the bottom (-5) and top (-6) walls (:93-174), the stable sort
 * permutations (:72,77), the reduced layer offsets with the reference's r[end] = n quirk
 * (:253-269), CSR-packs the neighbour lists, uploads, and computes the Delaunay lines (:186-245)
 * on the device.  `device` is the HIP device ordinal; device < 0 builds a HOST-ONLY handle (layers,
 * permutations, schedule introspection) on which every compute entry point fails with
 * VRT_ENODEVICE. */
int vrt_grid_create(int64_t n, const double *pos_zxy, const int64_t *nbr, int64_t D1,
                    const double bounds[6], int device, vrt_grid **out);
/* same, parsing the voro++ "%i %n" neighbour file (rt_preprocessing/output_sites.cc:49) the way
 * read_cell does (voronoi_utils.jl:42-70; max_guess = 70) */
int vrt_grid_create_from_file(const char *neighbours_file, int64_t n, const double *pos_zxy,
                              const double bounds[6], int device, vrt_grid **out);
void vrt_grid_destroy(vrt_grid *g);

/* ---- in-process tessellation: replaces the fork/exec of the voro++ wrapper
 * rt_preprocessing/output_sites.cc:35-49 (`voro`, src/functions.jl:13-23), the text round trip
 * (src/io.jl:8-40) and its parse (src/voronoi_utils.jl:42-63).  Voronoi cells of the sites in the
 * box, periodic in x and y, walls at z_min (-5) and z_max (-6); host code, multi-threaded, no GPU.
 *   nbr (n, D1) column-major out: exactly the matrix read_cell builds (column 0 = count, then the
 *   neighbour ids, 1-based); D1 = max_guess + 1 = 71 mirrors voronoi_utils.jl:42; *max_count
 *   receives maximum(nbr[:,1]).
 * The neighbour SET of every cell is the tessellation's; their ORDER inside a row (walls first,
 * then by increasing distance) is not voro++'s, which cannot be known without voro++ -- and the
 * reference's upwind rule depends on it (voronoi_utils.jl:378-386). */
int vrt_tessellate(int64_t n, const double *pos_zxy, const double bounds[6], int64_t D1, int64_t *nbr,
                   int64_t *max_count);
/* writes such a matrix as the voro++ "%i %n" text file read_cell parses (one line per cell) */
int vrt_write_neighbours_file(const char *path, int64_t n, const int64_t *nbr, int64_t D1);

/* introspection (parity tests; all values exactly as the reference's Julia fields, 1-based) */
int64_t vrt_grid_n(const vrt_grid *g);
int64_t vrt_grid_max_neighbours(const vrt_grid *g);             /* D = size(neighbours,2)-1   */
int64_t vrt_grid_num_layer_offsets(const vrt_grid *g, int dir); /* length(layers_up/down)     */
int vrt_grid_get_layers(const vrt_grid *g, int dir, int64_t *out); /* dir > 0 up, < 0 down   */
int vrt_grid_get_perm(const vrt_grid *g, int dir, int64_t *out);   /* n entries               */
/* Delaunay_lines as (3, D, n) column-major; wall slots are filled with 0 */
int vrt_grid_get_delaunay_lines(const vrt_grid *g, double *out);

/* ---- direction helper: k = [cos θ, cos ϕ sin θ, sin ϕ sin θ], lambda_iteration.jl:87 ------ */
void vrt_direction(double theta_deg, double phi_deg, double k[3]);

/* ---- plan: per-angle upwind tables (smallest_angle, voronoi_utils.jl:360-396, + weights and
 * path lengths, irregular_ray_tracing.jl:50-51,66) and the exact Gauss-Seidel dependency
 * schedule of irregular_ray_tracing.jl:37-80 / :118-161.  k is (3, n_angles) column-major; a
 * direction with k[0] < 0 is an "up" ray (θ > 90), k[0] > 0 "down"; k[0] == 0 (θ = 90) is
 * skipped exactly as the reference's callers do (lambda_iteration.jl:98,104). */
int vrt_plan_create(vrt_grid *g, int64_t n_angles, const double *k, int n_sweeps, vrt_plan **out);
/* same with an explicit sweep direction per angle (dirs[i] > 0 up / perm_up, < 0 down / perm_down,
 * 0 = skip), as Delaunay_upII / Delaunay_downII use whatever k they are handed; dirs == NULL
 * infers the direction from the sign of k[0] */
int vrt_plan_create_ex(vrt_grid *g, int64_t n_angles, const double *k, const int *dirs,
                       int n_sweeps, vrt_plan **out);
void vrt_plan_destroy(vrt_plan *p);
int64_t vrt_plan_num_levels(const vrt_plan *p);
int64_t vrt_plan_num_nodes(const vrt_plan *p);   /* live (site, sweep) updates per wavelength */
/* upwind table of one angle: up (2, n) 1-based ids, dots/w/r (2, n); any pointer may be NULL */
int vrt_plan_get_upwind(const vrt_plan *p, int64_t angle, int64_t *up, double *dots, double *w,
                        double *r);

/* ---- formal solve, all angles x all wavelengths: replaces the loop body of J_λ_voronoi ------
 *   S      (nlam, n) with leading dimension ld
 *   alpha  see VRT_ALPHA_*
 *   I0_up  (nlam, n1_up)   boundary intensity for up rays, ordered like perm_up[1:n1_up]
 *          (lambda_iteration.jl:99-101); NULL = zeros.  I0_down likewise (:105-106).
 *   RESERVED VALUE: where a step of one or two wavelength pairs runs as ONE chained launch (option VRT_CHAIN_DATAFLAG,
 *          default auto), an intensity is its own "written" flag, and "not yet written" is the quiet NaN whose 32-bit
 *          halves are both 0x7FF87FF8 (as a float: 0x7FF87FF8).  An ordinary NaN (0x7FF8000000000000, 0x7FC00000)
 *          propagates like anywhere else; an I0 or S that makes an intensity carry exactly the reserved pattern
 *          stalls its readers until the bounded wait expires (VRT_CHAIN_SPIN, ~2 s) and the call fails with
 *          VRT_ENODEVICE instead of returning it.  VRT_CHAIN_DATAFLAG=0 removes the reservation.
 *   weights[n_angles]      quadrature weights
 *   J      (nlam, n) out:  J = Σ_angles w · I  (lambda_iteration.jl:102,107), may be NULL
 *   I_out  (nlam, n, n_angles) out: per-angle intensities, may be NULL
 */
int vrt_plan_execute(vrt_plan *p, int64_t nlam, int64_t ld, const double *S, const double *alpha,
                     int alpha_mode, const double *I0_up, const double *I0_down,
                     const double *weights, double *J, double *I_out);
/* device-pointer variant; I0 leading dimension is nlam; I_out leading dimension is ld */
int vrt_plan_execute_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dS,
                         const double *dalpha, int alpha_mode, const double *dI0_up,
                         const double *dI0_down, const double *weights_host, double *dJ,
                         double *dI_out, void *stream);
/* ---- native layout of per-angle alpha (VRT_ALPHA_ANGLE_NATIVE) ------------------------------
 * storage order of a direction: out[pos] = 1-based site id at storage position pos (layers
 * contiguous like perm_up / perm_down; inside a layer strips of lattice columns, rows of y inside a strip, a row by x --
 * neighbouring positions are neighbouring sites of a row; VRT_STORE_ORDER=morton at grid creation: a Morton curve over (x, y)). */
int vrt_grid_get_storage_order(const vrt_grid *g, int dir, int64_t *out);
/* number of doubles of the native per-angle alpha buffer for nlam wavelengths */
int64_t vrt_plan_native_alpha_count(const vrt_plan *p, int64_t nlam);
/* pairs per block B of the plan's native layout (1, 2, 4, 8 or 16; option VRT_PAIR_BLOCK at creation) */
int vrt_plan_native_pair_block(const vrt_plan *p);
/* the same for the FLOAT native buffer (vrt_plan_alpha_to_native_dev_f32, vrt_line_opacity_dev_f32): at least 2,
 * two neighbouring pairs being one 16-byte access of the fp32 kernel (option VRT_PATCH_QUAD, default on) */
int vrt_plan_native_pair_block_f32(const vrt_plan *p);
/* converts the caller's (nlam, n, n_angles) alpha (VRT_ALPHA_ANGLE_SITE_LAM) once, e.g. per
 * Λ-iteration when alpha changes, so that every execute of that iteration reads it in place */
int vrt_plan_alpha_to_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dalpha,
                                 double *dalpha_native, void *stream);
/* the same for the fp32 VALUE path: float in, vrt_plan_native_alpha_count(p, nlam) FLOATS out, for
 * vrt_plan_execute_dev_f32 with VRT_ALPHA_ANGLE_NATIVE */
int vrt_plan_alpha_to_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dalpha,
                                     float *dalpha_native, void *stream);

/* ---- S and J in SWEEP ORDER (device entry points): what a device-resident Λ-iteration keeps between its steps ----
 * The sweep reads S and reduces J per sweep direction in that direction's storage order; a caller that produces S and
 * consumes J on the device (S_new = (1 - ε) J + ε B, the rate integrals: src/lambda_iteration.jl:261-263,
 * src/rates.jl:154-201) can keep both in that form and save the two layout changes of every execute (1 M sites x 51
 * wavelengths: 0.75 of 8.3 ms).  Per direction (up: θ > 90, down: θ < 90) ONE plane set of
 * vrt_plan_native_plane_count(p, nlam) doubles: element (l, pos) at ((l/2) * n + pos) * 2 + l%2, pos = the site's
 * position in vrt_grid_get_storage_order of the direction; an odd nlam is padded with one wavelength (S: any finite
 * value, J: unspecified).  Needs a plan on a layer path with VRT_PAIR_BLOCK=1 (the default for doubles). */
int64_t vrt_plan_native_plane_count(const vrt_plan *p, int64_t nlam);
/* caller's (nlam, n) array with leading dimension ld -> the plane sets of both directions (either may be NULL) */
int vrt_plan_to_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *d_in, double *d_up, double *d_down,
                           void *stream);
/* the plane set of ONE direction (dir > 0: up) -> the caller's (nlam, n) array */
int vrt_plan_from_native_dev(vrt_plan *p, int dir, int64_t nlam, int64_t ld, const double *d_native, double *d_out,
                             void *stream);
/* J[site][l] = J_up + J_down (a direction without angles: NULL), the sum vrt_plan_execute_dev forms itself */
int vrt_plan_j_from_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dJ_up, const double *dJ_down,
                               double *dJ, void *stream);
/* vrt_plan_execute_dev with S read from and J reduced into sweep-order plane sets, in place.  alpha_mode:
 * VRT_ALPHA_SITE, VRT_ALPHA_ANGLE_NATIVE or VRT_ALPHA_SITE_LAM_NATIVE.  dJ_up / dJ_down: both or neither; the plane set of a direction without
 * angles comes back zeroed.  The results are those of vrt_plan_execute_dev bit for bit (J = J_up + J_down). */
int vrt_plan_execute_native_dev(vrt_plan *p, int64_t nlam, const double *dS_up, const double *dS_down,
                                const double *dalpha, int alpha_mode, const double *dI0_up, const double *dI0_down,
                                const double *weights_host, double *dJ_up, double *dJ_down, void *stream);
/* The same four with FLOAT storage (vrt_plan_execute_dev_f32's values; arithmetic in double): the plane sets hold
 * vrt_plan_native_plane_count(p, nlam) floats per direction, wavelength pairs in blocks of
 * vrt_plan_native_pair_block_f32(p) as in the float native alpha; the patch path only.  J_up + J_down (vrt_plan_j_from_native_dev_f32)
 * equals the J of vrt_plan_execute_dev_f32 bit for bit. */
int vrt_plan_to_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *d_in, float *d_up, float *d_down,
                               void *stream);
int vrt_plan_from_native_dev_f32(vrt_plan *p, int dir, int64_t nlam, int64_t ld, const float *d_native, float *d_out,
                                 void *stream);
int vrt_plan_j_from_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dJ_up, const float *dJ_down,
                                   float *dJ, void *stream);
int vrt_plan_execute_native_dev_f32(vrt_plan *p, int64_t nlam, const float *dS_up, const float *dS_down,
                                    const float *dalpha, int alpha_mode, const float *dI0_up, const float *dI0_down,
                                    const double *weights_host, float *dJ_up, float *dJ_down, void *stream);

/* The two steps either side of the sweep in a device-resident Λ-iteration, on the same sweep-order plane sets (what
 * vrt_lambda_iterate runs internally): vrt_lambda_update_dev -- S_new = (1 - ε) J + ε B and the criterion's maximum
 * (src/lambda_iteration.jl:261-263, :325-349) -- with J = J_up + J_down, B in the UP order (vrt_plan_to_native_dev, once),
 * the OLD S read from dS_up and overwritten there, its down-order copy written to dS_down; and vrt_rates_populations_dev
 * (src/rates.jl:154-201, src/populations.jl:191-221) reading J from the two plane sets.  Arguments otherwise as their
 * caller-layout counterparts; results bit for bit theirs. */
int vrt_lambda_update_native_dev(vrt_grid *g, int64_t nlam, const double *dJ_up, const double *dJ_down,
                                 const double *dB_up, const double *deps, double *dS_up, double *dS_down,
                                 double *max_rel_change, void *stream);
int vrt_rates_populations_native_dev(vrt_grid *g, int64_t nlam, const double *lambda, const int64_t blocks[6],
                                     const double *dJ_up, const double *dJ_down, const double *planck2,
                                     double lambda0, double c0, const double *d_doppler_width, const double *d_gamma,
                                     double sigma_bb_const, const double *sigma_bf1, const double *sigma_bf2,
                                     const double *d_temperature, const double *d_lte_populations, double hc_over_kB,
                                     double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                                     double *d_R, double *d_populations, void *stream);

/* fp32 VALUE path (BASELINE config C5): S, alpha, I_0, J and the per-angle intensities are stored
 * as float, halving the bytes of this bandwidth-bound path; the geometry tables and all
 * arithmetic stay fp64.  Results agree with the fp64 solve to fp32 storage rounding (~1e-6
 * relative).  Runs on the layer-step kernels (coefficients handed over and level tiles held as
 * float) when no layer exceeds 18 432 sites, on the level kernels otherwise. */
int vrt_plan_execute_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dS,
                             const float *dalpha, int alpha_mode, const float *dI0_up,
                             const float *dI0_down, const double *weights_host, float *dJ,
                             float *dI_out, void *stream);
/* time (ms, HIP events on the launch stream) the sweep kernels of the last execute took, and the
 * number of sweep-kernel launches it made (waits for the sweep; VRT_ENODEVICE if its chained patch launch gave up
 * waiting for a dependency -- a bounded spin expired: the results of that execute are invalid; also reported by the
 * next execute of the plan) */
int vrt_plan_last_sweep_timing(const vrt_plan *p, double *ms, int64_t *launches);
/* VRT_OK, or the give-up of a chained patch launch that has finished since the last check (VRT_ENODEVICE: the results of
 * that execute are invalid).  The entry points that synchronise themselves (vrt_plan_execute, vrt_plan_execute_line,
 * vrt_lambda_iterate, vrt_multi_*) report it from the call it spoiled; a caller of the ASYNCHRONOUS entry points
 * (vrt_plan_execute_dev, _native_dev, _f32) calls this -- or vrt_plan_last_sweep_timing -- after it has synchronised
 * its stream.  No synchronisation here. */
int vrt_plan_check(vrt_plan *p);
/* which device path the last execute took: 1 = "levels" (one launch per dependency level),
 * 2 = "tiles" (one persistent launch), 3 = "steps" (two launches per BFS layer), 4 = "patches" (layers cut
 * into patches, the default: ONE chained launch for every layer, or one fused launch per BFS layer); 0 = none yet.  The paths give the
 * same results; option VRT_PATH selects one (performance experiments, the parity tests' cross-checks). */
int vrt_plan_last_path(const vrt_plan *p);
/* Tuning options (performance experiments and the tests' cross-checks; results never depend on them).
 * Every option has a default; an environment variable of the option's name presets it and is read ONCE,
 * when a plan is created -- an execute reads no environment.  vrt_plan_set_option changes an option of a
 * live plan; vrt_grid_set_option sets it for the plans vrt_delaunay_up / _down cache inside the grid.
 *   VRT_PATH = auto | levels | tiles | steps | patches
 *   VRT_PATCH_Q, VRT_PATCH_TARGET              wavelength pairs at a time / workgroups per launch of the patch kernel
 *   VRT_PATCH_K, VRT_PATCH_NT, VRT_PATCH_OWN   entries per thread, threads, owned sites per patch (creation only)
 *   VRT_PATCH_LEAN = 0 | 1                     the 64-register patch kernel, four workgroups per CU (default 1)
 *   VRT_PATCH_CHAIN = 0 | 1 | 2                every layer inside ONE launch, ordered by the data's own dependencies: never
 *                                              (one launch per layer and direction), wherever the kernel exists, auto
 *                                              (default: where a layer alone cannot fill the chip)
 *   VRT_CHAIN_PAIRS, VRT_CHAIN_SPIN            wavelength-pair blocks per item of the chained launch at most (default 9); polls
 *                                              (x 1024) after which a waiting workgroup gives up (default 2048)
 *   VRT_CHAIN_DATAFLAG = 0 | 1 | 2             chained launch, fp64: the stored intensities are their own flags (the planes are
 *                                              filled with a NaN pattern before the launch; a gather that still holds it
 *                                              is repeated) instead of progress words: never, always, auto (default: for
 *                                              one or two wavelength pairs, where the fill is cheaper than the words)
 *   VRT_PATCH_QUAD = 0 | 1                     fp32 storage: four wavelengths per lane (creation only; default 1)
 *   VRT_PAIR_BLOCK = 1 | 2 | 4 | 8 | 16        wavelength pairs of a site side by side in the patch path's planes and
 *                                              in the plan's native alpha (creation only; default 1)
 *   VRT_STEP_K, VRT_STEP_SINGLE, VRT_STEP_PAIRS, VRT_STEP_XCD, VRT_STEP_STREAMS, VRT_STEP_LEVEL_MAP,
 *   VRT_STEP_GROUP_DIR, VRT_TILE_WIDE, VRT_TILE_PRE, VRT_GRAPH     variants of the older paths
 * VRT_EINVAL for an unknown name, a value out of range, or a creation-only option on a live plan. */
int vrt_plan_set_option(vrt_plan *p, const char *name, const char *value);
int vrt_grid_set_option(vrt_grid *g, const char *name, const char *value);

/* ---- schedule introspection (host only, works on a device < 0 grid handle) -----------------
 * The dependency schedule of one direction given an upwind table `up` (2, n), 1-based ids as
 * returned by vrt_plan_get_upwind (0 = none).  Nodes are the surviving (site, sweep) visits of
 * irregular_ray_tracing.jl:37-80 sorted by level; zflags bit 0/1 = the read of I[upwind 1/2]
 * sees the initial zero.  level_off has num_levels + 1 entries. */
typedef struct vrt_schedule vrt_schedule;
int vrt_schedule_build(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, vrt_schedule **out);
int64_t vrt_schedule_num_nodes(const vrt_schedule *s);
int64_t vrt_schedule_num_levels(const vrt_schedule *s);
int vrt_schedule_get(const vrt_schedule *s, int64_t *site, int32_t *zflags, int64_t *level_off);
void vrt_schedule_destroy(vrt_schedule *s);
/* The layer-local form of the same schedule used by the LDS layer-tile kernel: vis[n] packs up to
 * four 8-bit in-layer visit levels per site (0 = none), nlev[num_layer_offsets] the level count
 * of each layer (index = 1-based layer).  VRT_EINVAL if it does not fit that encoding. */
int vrt_layer_schedule(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, uint32_t *vis,
                       int32_t *nlev, int64_t *n_visits);
/* Thread assignment of the layer-step level kernel for such a `vis` (introspection, host only):
 * store[n] = 1-based site id at each storage position (layers contiguous, the storage order inside);
 * self[n] = 0-based storage position held by each sorted index: inside every layer the
 * positions are sorted (stably) by visit pattern -- first, then second, ... visit level. */
int vrt_layer_sorted_slots(const vrt_grid *g, int dir, const uint32_t *vis, int64_t *store, int64_t *self);

/* Patch schedule of the fused layer kernel ("patches" path; introspection, host only): every BFS layer
 * is cut into ranges of about `own_target` consecutive storage positions, and each range carries
 * the in-layer dependency cone of its sites (own sites + a halo of neighbouring sites' visits it
 * recomputes), at most `entry_cap` entries.  Step 1: counts[0..5] = patches, entries, visits
 * executed over all patches, live visits of the unsplit schedule, largest entry count of a patch,
 * layer offsets (= vrt_grid_num_layer_offsets + 1).  Step 2 (vrt_patch_schedule_get): layer_patch_off
 * [counts[5]], patch_own_lo / patch_own_cnt / patch_nlev [patches], patch_ent_off [patches + 1],
 * entry_pos (0-based storage position) / entry_vis / entry_loc [entries]; any pointer may be NULL. */
typedef struct vrt_patch_schedule vrt_patch_schedule;
int vrt_patch_schedule_build(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, int own_target,
                             int entry_cap, vrt_patch_schedule **out, int64_t counts[6]);
int vrt_patch_schedule_get(const vrt_patch_schedule *s, int32_t *layer_patch_off, int32_t *patch_own_lo,
                           int32_t *patch_own_cnt, int32_t *patch_nlev, int64_t *patch_ent_off,
                           int32_t *entry_pos, uint32_t *entry_vis, uint32_t *entry_loc);
/* Dependencies BETWEEN patches (what the chained launch of the patch path waits on, vrt_patch.hip: k_patch_chain):
 * patch q gathers, as upwind intensities of EARLIER layers (irregular_ray_tracing.jl:75), values that the
 * patches dep_list[dep_off[q] .. dep_off[q+1]) store (sorted patch indices of this schedule; the boundary layer
 * and the never-visited last site have no owner).  dep_off [patches + 1] first, then dep_list [dep_off[patches]];
 * either pointer may be NULL. */
int vrt_patch_schedule_get_deps(const vrt_patch_schedule *s, int64_t *dep_off, int32_t *dep_list);
/* The angle's layer schedule as the patch builder derives it on the way (plan creation uses this one): same contents as
 * vrt_layer_schedule's outputs (vis[n], nlev[vrt_grid_num_layer_offsets], *n_visits); any pointer may be NULL. */
int vrt_patch_schedule_get_layers(const vrt_patch_schedule *s, uint32_t *vis, int32_t *nlev, int64_t *n_visits);
void vrt_patch_schedule_destroy(vrt_patch_schedule *s);

/* ---- single solves: drop-in bodies for Delaunay_upII / Delaunay_downII --------------------
 * (src/irregular_ray_tracing.jl:15-20,96-101).  nI0 must equal layers[2]-1 of the direction.
 * The plan for k is cached inside the grid. */
int vrt_delaunay_up(vrt_grid *g, const double k[3], const double *S, const double *I0,
                    int64_t nI0, const double *alpha, int n_sweeps, double *I_out);
int vrt_delaunay_down(vrt_grid *g, const double k[3], const double *S, const double *I0,
                      int64_t nI0, const double *alpha, int n_sweeps, double *I_out);

/* ---- Λ-iteration epilogue on the device (the physics-free part) -----------
 * S_new[l,i] = (1 - eps[i]) J[l,i] + eps[i] B[l,i]           (src/lambda_iteration.jl:261-263)
 * *max_rel_change = max |1 - S_old/S_new|, NaN if any term is NaN  (criterion, :325-349)
 * All arrays are device pointers, (nlam, n) with leading dimension ld, eps[n]; S_new may alias
 * neither S_old nor J.  Synchronises `stream` (the scalar comes back to the host), so a whole
 * Λ-iteration J -> S_new -> J ... stays device-resident. */
int vrt_lambda_update_dev(vrt_grid *g, int64_t nlam, int64_t ld, const double *dJ, const double *dB,
                          const double *deps, const double *dS_old, double *dS_new,
                          double *max_rel_change, void *stream);

/* ---- fused per-angle opacity prologue -------------------------------------
 * alpha_tot = alphaline_lambda + alpha_cont for EVERY angle of the plan from per-site line parameters,
 * written directly in the native layout of VRT_ALPHA_ANGLE_NATIVE -- replaces the per-angle part of
 * J_λ_voronoi between the angle loop header and the formal solve: damping_λ
 * (src/lambda_iteration.jl:72-80, src/broadening.jl:87-89), compute_voigt_profile with the
 * line-of-sight velocity of -k (src/line.jl:121-137, :198-208), αline_λ (src/line.jl:219-225) and
 * α_tot = αline + α_cont (src/lambda_iteration.jl:93-96):
 *   v_los = dot(velocity, -k);  a = γ λ²/(4π c0 ΔλD);  v = (λ - λ0 + λ0 v_los/c0)/ΔλD
 *   alpha = line_strength * H(a, v)/(sqrt(π) ΔλD) + alpha_cont
 * H = Re w4(v + i a), Humlíček's w4 as Transparency.jl's voigt_profile (absent from the reference
 * checkout; tolerance-based parity).  Plain numbers in one unit system of the caller's choice:
 *   lambda[nlam] (host), lambda0, c0;  device: velocity (3, n) rows z,x,y; doppler_width ΔλD[n];
 *   gamma[n]; line_strength[n] = h c0/(4π λ0) (n_i B_ij - n_j B_ji); alpha_cont[n];
 *   alpha_native: vrt_plan_native_alpha_count(p, nlam) doubles, out. */
int vrt_line_opacity_dev(vrt_plan *p, int64_t nlam, const double *lambda, double lambda0, double c0,
                         const double *d_velocity, const double *d_doppler_width, const double *d_gamma,
                         const double *d_line_strength, const double *d_alpha_cont, double *d_alpha_native,
                         void *stream);
/* the same with the result STORED as float (fp32 value path; per-site inputs and arithmetic stay fp64):
 * vrt_plan_native_alpha_count(p, nlam) floats for vrt_plan_execute_dev_f32 with VRT_ALPHA_ANGLE_NATIVE */
int vrt_line_opacity_dev_f32(vrt_plan *p, int64_t nlam, const double *lambda, double lambda0, double c0,
                             const double *d_velocity, const double *d_doppler_width, const double *d_gamma,
                             const double *d_line_strength, const double *d_alpha_cont, float *d_alpha_native,
                             void *stream);

/* ---- the same from HOST arrays: the whole body of J_λ_voronoi, line case, in one call --------------------
 * src/lambda_iteration.jl:72-111 for a host without device arrays of its own (the reference's Julia driver):
 * uploads the seven per-site line vectors, λ and S (about (nλ + 7) n doubles -- NOT α_tot (nλ, n, n_angles),
 * which is nλ n_angles n), makes α_tot of every angle on the device in the native layout, solves every
 * angle x wavelength and returns J.  Arguments as vrt_line_opacity_dev / vrt_plan_execute, all host pointers;
 * gamma[n] is γ_constant of the current populations (or vrt_lambda_* below keeps the whole loop on the device). */
int vrt_plan_execute_line(vrt_plan *p, int64_t nlam, int64_t ld, const double *lambda, double lambda0, double c0,
                          const double *velocity, const double *doppler_width, const double *gamma,
                          const double *line_strength, const double *alpha_cont, const double *S, const double *I0_up,
                          const double *I0_down, const double *weights, double *J);

/* ---- population-dependent line terms on the device ------------------------------------------------------
 *   gamma[i]         = gamma_static[i] + gamma_unsold[i] (n_1 + n_2)     γ_constant, src/broadening.jl:63-82, which
 *                      J_λ_voronoi evaluates with populations[:,1] .+ populations[:,2] in EVERY iteration
 *                      (lambda_iteration.jl:72-75).  gamma_static = natural width + linear + quadratic Stark
 *                      (electron density and temperature only: fixed), gamma_unsold = γ_unsold per unit
 *                      neutral-hydrogen density (van der Waals; linear in the density) -- both come from the
 *                      caller, whose Transparency.jl holds the constants
 *   line_strength[i] = strength_const (n_1 B_ij - n_2 B_ji)              αline_λ, src/line.jl:219-225
 * populations (n, 3) as Julia's; gamma or line_strength may be NULL.  Device pointers. */
int vrt_line_terms_dev(vrt_grid *g, const double *d_gamma_static, const double *d_gamma_unsold,
                       const double *d_populations, double strength_const, double Bij, double Bji, double *d_gamma,
                       double *d_line_strength, void *stream);

/* ---- Λ_voronoi's loop with library-owned device state (src/lambda_iteration.jl:205-300) -----------------
 * For a single-process host with HOST arrays: create uploads the per-site inputs Λ_voronoi derives before its
 * loop, starts in LTE with S = B_0 (:232-240); every vrt_lambda_iterate is one pass of the loop body --
 *   γ, line strength of the current populations -> α_tot of every angle (:72-96) -> J_λ (:84-111) ->
 *   S_new = (1 - ε) J + ε B_0 and max |1 - S_old/S_new| (:261-263, :325-349) -> R, populations (:269, :274)
 * -- and returns that maximum (NaN like Julia's); vrt_lambda_get downloads what the caller wants to keep
 * (the reference checkpoints populations and S_new each iteration, :280-281).  All arrays host, (…) = Julia
 * dims; plain numbers in one unit system, unit factors folded into the constants as in
 * vrt_rates_populations_dev.  weights[n_angles] of the plan's quadrature. */
typedef struct vrt_lambda vrt_lambda;
typedef struct vrt_line_case {
    int64_t nlam;
    const double *lambda;            /* [nlam]: bound-bound block first, then the two bound-free blocks */
    int64_t blocks[6];               /* their [lo, hi) offsets, 0-based (line.λidx) */
    double lambda0, c0;
    const double *velocity;          /* (3, n) rows z, x, y */
    const double *doppler_width;     /* [n] ΔλD */
    const double *gamma_static;      /* [n] see vrt_line_terms_dev */
    const double *gamma_unsold;      /* [n] */
    const double *alpha_cont;        /* [n] */
    const double *eps;               /* [n] destruction probability ε */
    const double *temperature;       /* [n] */
    const double *atom_density;      /* [n] */
    const double *B0;                /* (nlam, n) Planck function */
    const double *lte_populations;   /* (n, 3) */
    const double *C;                 /* (3, 3, n) collisional rates */
    const double *planck2;           /* [nlam] 2 h c0² / λ⁵ in J's unit */
    const double *sigma_bf1, *sigma_bf2;   /* σic per wavelength of the two bound-free blocks */
    double strength_const, Bij, Bji;       /* αline_λ = strength_const (n_1 B_ij - n_2 B_ji) φ */
    double sigma_bb_const, hc_over_kB, pref_ij, pref_ji;
} vrt_line_case;
int vrt_lambda_create(vrt_plan *p, const vrt_line_case *lc, const double *weights, vrt_lambda **out);
int vrt_lambda_iterate(vrt_lambda *s, double *max_rel_change);
/* J, S (nlam, n), populations (n, 3), R (3, 3, n), gamma [n] of the last iteration; any pointer may be NULL */
int vrt_lambda_get(vrt_lambda *s, double *J, double *S, double *populations, double *R, double *gamma);
void vrt_lambda_destroy(vrt_lambda *s);

/* ---- several GPUs of a node from ONE host process -------------------------------------
 * The object owns a grid + plan per device and -- when the devices are distinct -- an in-process RCCL
 * communicator (ncclCommInitAll; librccl is dlopen'ed on first use).  vrt_multi_execute is vrt_plan_execute for
 * the node: the angle x wavelength loop of J_λ_voronoi (src/lambda_iteration.jl:84-111) sharded
 *   by wavelength blocks when nlam >= devices (51 over 8 -> 7,7,7,6,6,6,6,6; every device owns whole rows of J,
 *   which go home by strided copies: no exchange), or
 *   by angles otherwise (ups and downs dealt separately), the partial J's summed by ONE RCCL all-reduce.
 * Arguments as vrt_grid_create + vrt_plan_create_ex / vrt_plan_execute (host pointers; alpha_mode 0..2).
 * The same device listed twice is a rehearsal on a one-GPU box: no RCCL, the partial sums are added by a
 * kernel.  vrt_multi_set_shard: "auto" | "lambda" | "angle"; vrt_multi_last_shard: 1 lambda, 2 angle. */
typedef struct vrt_multi vrt_multi;
int vrt_multi_create(int n_devices, const int *devices, int64_t n, const double *pos_zxy, const int64_t *nbr, int64_t D1,
                     const double bounds[6], int64_t n_angles, const double *k, const int *dirs, int n_sweeps,
                     vrt_multi **out);
int vrt_multi_execute(vrt_multi *m, int64_t nlam, int64_t ld, const double *S, const double *alpha, int alpha_mode,
                      const double *I0_up, const double *I0_down, const double *weights, double *J);
int vrt_multi_set_shard(vrt_multi *m, const char *mode);
int vrt_multi_last_shard(const vrt_multi *m);
int vrt_multi_uses_rccl(const vrt_multi *m);
void vrt_multi_destroy(vrt_multi *m);
/* vrt_plan_execute_line for the node (the body of J_λ_voronoi, line method, src/lambda_iteration.jl:72-111, from host
 * arrays): the wavelengths go to the devices in contiguous blocks, every device makes the per-angle α_tot of ITS
 * wavelengths itself (it never exists on the host and never crosses PCIe) and owns whole rows J[l, :]: no exchange.
 * Arguments as vrt_plan_execute_line. */
int vrt_multi_execute_line(vrt_multi *m, int64_t nlam, int64_t ld, const double *lambda, double lambda0, double c0,
                           const double *velocity, const double *doppler_width, const double *gamma,
                           const double *line_strength, const double *alpha_cont, const double *S, const double *I0_up,
                           const double *I0_down, const double *weights, double *J);
/* The Λ-iteration session of vrt_lambda_* across the devices of a vrt_multi (BASELINE configs[3];
 * src/lambda_iteration.jl:205-300).  Every device owns a contiguous block of the wavelengths and a copy of the per-site
 * state; per iteration it runs, for its wavelengths, the line terms, α_tot of every angle, the sweep, S_new with its
 * share of the convergence maximum, and its share of the six λ-integrals of the radiative rates (src/rates.jl:154-201,
 * regrouped per wavelength: Σ_l W_l f_l).  The shares are summed by ONE RCCL all-reduce of 6 n doubles -- J never
 * travels -- and every device solves the statistical equilibrium of every site itself
 * (src/populations.jl:191-221).  Results equal the one-device session's to the rounding of that regrouping (1e-12).
 * vrt_multi_lambda_get assembles J and S_new (n, nlam) from the devices' blocks; populations (n, 3), R (3, 3, n) and γ [n]
 * come from the first device.  LIFETIME: a session uses its vrt_multi in every call but vrt_multi_lambda_destroy --
 * destroy the session FIRST; vrt_multi_lambda_iterate / _get on a session whose vrt_multi is gone are undefined. */
typedef struct vrt_multi_lambda vrt_multi_lambda;
int vrt_multi_lambda_create(vrt_multi *m, const vrt_line_case *lc, const double *weights, vrt_multi_lambda **out);
int vrt_multi_lambda_iterate(vrt_multi_lambda *s, double *max_rel_change);
int vrt_multi_lambda_get(vrt_multi_lambda *s, double *J, double *S, double *populations, double *R, double *gamma);
void vrt_multi_lambda_destroy(vrt_multi_lambda *s);

/* ---- rates + populations of the Λ-iteration epilogue on the device --------
 * calculate_R (src/rates.jl:154-201: Rij / Rji λ-trapezoids :226-364, σij with the site's static
 * Voigt profile :374-416, Gij :459-476) and get_revised_populations (src/populations.jl:191-221,
 * the per-site 2 x 2 solve) for the reference's 2-level + continuum atom, from J in place:
 *   R_ij = Σ_l pref_ij ((λ_l σ_l J_l + λ_l+1 σ_l+1 J_l+1)(λ_l+1 - λ_l))
 *   R_ji = Σ_l pref_ji ((σ_l G_l λ_l (P_l + J_l) + σ_l+1 G_l+1 λ_l+1 (P_l+1 + J_l+1))(λ_l+1 - λ_l))
 *   G_l = LTE[i]/LTE[j] exp(-hc_over_kB/(λ_l T)),  P_l = planck2[l] = 2 h c0²/λ_l⁵ in J's unit
 * pref_ij / pref_ji = 2π/(h c0) times the unit factors the reference gets from Unitful (and its
 * explicit /1000 in Rij, rates.jl:237,263).  blocks = [lo, hi) (0-based) of the bound-bound, level-1
 * bound-free and level-2 bound-free wavelength blocks (line.λidx).  Host: lambda[nlam],
 * planck2[nlam], sigma_bf1 / sigma_bf2 (σic per wavelength of the block).  Device: J (nlam, n) with
 * leading dimension ld, doppler_width[n], gamma[n], temperature[n], lte_populations (n, 3),
 * C (3, 3, n) collisional rates, atom_density[n]; out: R (3, 3, n), populations (n, 3). */
int vrt_rates_populations_dev(vrt_grid *g, int64_t nlam, int64_t ld, const double *lambda,
                              const int64_t blocks[6], const double *dJ, const double *planck2,
                              double lambda0, double c0, const double *d_doppler_width, const double *d_gamma,
                              double sigma_bb_const, const double *sigma_bf1, const double *sigma_bf2,
                              const double *d_temperature, const double *d_lte_populations, double hc_over_kB,
                              double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                              double *d_R, double *d_populations, void *stream);

/* ---- regular-grid short characteristics ---------------------------------
 * Drop-in bodies for short_characteristics_up / short_characteristics_down
 * (src/characteristics.jl:19-95, :110-180), batched over independent solves the way
 * J_λ_regular loops over angles and wavelengths (src/lambda_iteration.jl:1-58).
 *   z[nz], x[nx], y[ny]   grid axes (x, y carry the reference's one-cell periodic ghost border)
 *   k (3, n_solve), up[n_solve] (1: rays travel up, boundary at z[1]; 0: down, boundary at z[end])
 *   S, alpha (nz, nx, ny[, n_solve]) Julia order a[iz + nz*(ix + nx*iy)]; stride 0 = shared by all
 *   solves, nz*nx*ny = one array per solve
 *   I0 (nx, ny, n_solve), I_out (nz, nx, ny, n_solve)
 * Host pointers; the arrays are staged through the device. */
int vrt_short_characteristics(int64_t nz, int64_t nx, int64_t ny, const double *z, const double *x,
                              const double *y, int64_t n_solve, const double *k, const int *up,
                              const double *S, int64_t S_stride, const double *alpha,
                              int64_t alpha_stride, const double *I0, int n_sweeps, int device,
                              double *I_out);

/* Device-resident form: a handle owns the grid axes and the workspaces; dS, dalpha, dI0, dI_out are
 * device pointers in the layouts above, k and up stay on the host; asynchronous on `stream`.
 * field_period > 0: solve s reads the S / alpha array number s % field_period (solves ordered
 * direction-major with the wavelength fastest share one array per wavelength); 0: array s.
 * vrt_regular_last_solve_ms: HIP-event time of the last execute's solve kernel(s) (synchronise first).
 * A handle serves one caller at a time (its workspaces are reused by every execute).
 * A batch in which every ray cuts the horizontal plane first on every plane (steep rays: xy_up_ray / xy_down_ray
 * only, src/characteristics.jl:72, :191-372) runs as a chip-wide coefficient kernel + one light march per solve,
 * with 3 doubles per point, plane and solve of extra workspace (chunks of at most 2 GiB); results are bit-identical
 * to the single kernel's.  Environment, read at vrt_regular_create (diagnostics and tests): VRT_REG_XY = 0 keeps the
 * single kernel for such batches, 2 reads the upwind plane back from memory instead of LDS; VRT_REG_THREADS forces
 * the workgroup size. */
typedef struct vrt_regular vrt_regular;
int vrt_regular_create(int64_t nz, int64_t nx, int64_t ny, const double *z, const double *x,
                       const double *y, int device, vrt_regular **out);
void vrt_regular_destroy(vrt_regular *r);
int vrt_regular_execute_dev(vrt_regular *r, int64_t n_solve, const double *k, const int *up,
                            const double *dS, int64_t S_stride, const double *dalpha,
                            int64_t alpha_stride, int64_t field_period, const double *dI0,
                            int n_sweeps, double *dI_out, void *stream);
int vrt_regular_last_solve_ms(const vrt_regular *r, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* VORONOIRT_H */
