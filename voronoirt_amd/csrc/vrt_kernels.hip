// HIP kernels for gfx950 (CDNA4, wave64) and their launchers.
//
// Floating-point contract (shared with oracle/vrt_oracle.c): this file is compiled with
// -ffp-contract=off, every expression is evaluated left to right as the reference writes it, so
// the integer results (upwind neighbour ids) agree bit for bit with the CPU oracle; fp64
// division and sqrt are correctly rounded on the device, exp/pow are within 1 ulp.
//
// No MFMA anywhere: the path is gather + exp + a handful of multiplies per (site, wavelength).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "vrt_internal.h"

namespace vrt {

// --------------------------------------------------------------------------------------------
// Delaunay lines, src/voronoi_utils.jl:186-245: unit vectors site -> neighbour with the
// reference's periodic-image rule (shift on the right edge :219-220, MIRROR on the left
// :221-222).  CSR-packed SoA (lz, lx, ly), one entry per neighbour slot; wall slots get 0.
// One 16-lane group per site so the CSR row is read/written contiguously.
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_delaunay_lines(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                 const double *__restrict__ pos, double x_min, double x_max, double y_min,
                 double y_max, double *__restrict__ lz, double *__restrict__ lx,
                 double *__restrict__ ly)
{
    const int lane16 = threadIdx.x & 15;
    const int64_t site = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (site >= n) return;
    const double pz = pos[3 * site + 0], px = pos[3 * site + 1], py = pos[3 * site + 2];
    const double x_r_r = x_max - px, x_r_l = px - x_min;   // :200-201
    const double y_r_r = y_max - py, y_r_l = py - y_min;   // :203-204
    const int beg = rowptr[site], end = rowptr[site + 1];
    for (int e = beg + lane16; e < end; e += 16) {
        const int id = col[e];
        double oz = 0.0, ox = 0.0, oy = 0.0;
        if (id > 0) {
            const int64_t q = id - 1;
            const double qz = pos[3 * q + 0];
            double qx = pos[3 * q + 1], qy = pos[3 * q + 2];
            const double x_i_r = fabs(x_max - qx), x_i_l = fabs(qx - x_min);   // :215-216
            if (x_r_r + x_i_l < px - qx) qx = x_max + qx - x_min;              // :219-220
            else if (x_r_l + x_i_r < qx - px) qx = x_min + x_max - qx;         // :221-222
            const double y_i_r = fabs(y_max - qy), y_i_l = fabs(qy - y_min);
            if (y_r_r + y_i_l < py - qy) qy = y_max + qy - y_min;              // :229-230
            else if (y_r_l + y_i_r < qy - py) qy = y_min + y_max - qy;         // :231-232
            const double dz = qz - pz, dx = qx - px, dy = qy - py;             // :235
            const double nrm = sqrt((dz * dz + dx * dx) + dy * dy);            // :237
            oz = dz / nrm;
            ox = dx / nrm;
            oy = dy / nrm;
        }
        lz[e] = oz;
        lx[e] = ox;
        ly[e] = oy;
    }
}

int launch_delaunay_lines(vrt_grid *g)
{
    const int64_t threads = g->n * 16;
    const int64_t blocks = (threads + 255) / 256;
    hipLaunchKernelGGL(k_delaunay_lines, dim3((unsigned)blocks), dim3(256), 0, g->stream, g->n,
                       g->d_rowptr, g->d_col, g->d_pos, g->bounds[2], g->bounds[3], g->bounds[4],
                       g->bounds[5], g->d_lz, g->d_lx, g->d_ly);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// --------------------------------------------------------------------------------------------
// Upwind ("smallest angle") search, src/voronoi_utils.jl:360-396, hoisted out of the sweep:
// it depends on (site, direction) only.  One 16-lane group per site (4 sites per wave64); the
// neighbour row is visited 16 slots at a time IN FILE ORDER and the reference's order-dependent
// rule is evaluated in closed form per chunk:
//   running best before slot j:  R_j = max(D1, d_0 .. d_{j-1})          (16-lane prefix max)
//   slot j is a record  <=>  d_j > R_j   -> the LAST record (= first arg-max, found with a
//                                           wave ballot) becomes slot 1; earlier records are
//                                           discarded, not demoted (:379-381)
//   non-records compete for slot 2 with strict '>', so the FIRST arg-max among them that beats
//                                           the incoming D2 wins (:382-385)
// then `dots[2] <= 0 -> 0, indices[2] = indices[1]` (:390-393), the weights
// dots^7 / sum(dots^7) (irregular_ray_tracing.jl:51) and the un-wrapped path lengths (:66).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned group_bits(unsigned long long ballot, int group)
{
    return (unsigned)((ballot >> (group * 16)) & 0xFFFFull);
}

__global__ void __launch_bounds__(256)
k_upwind_table(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
               const double *__restrict__ lz, const double *__restrict__ lx,
               const double *__restrict__ ly, const double *__restrict__ pos, double k0, double k1,
               double k2, int32_t *__restrict__ up1, int32_t *__restrict__ up2,
               double *__restrict__ d1o, double *__restrict__ d2o, double *__restrict__ w1o,
               double *__restrict__ w2o, double *__restrict__ r1o, double *__restrict__ r2o)
{
    const int lane16 = threadIdx.x & 15;
    const int group = (threadIdx.x & 63) >> 4;
    const int64_t gsite = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool in_range = gsite < n;
    const int64_t site = in_range ? gsite : n - 1;   // keep every lane in the shuffles
    const int beg = rowptr[site], end = rowptr[site + 1];
    // all four groups of the wave iterate the same number of chunks (wave-uniform loop bound)
    int chunks = (end - beg + 15) >> 4;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) chunks = max(chunks, __shfl_xor(chunks, off, 64));

    double D1 = -1.0, D2 = -1.0;   // :365-366
    int i1 = 0, i2 = 0;            // 1-based ids, 0 = unset (:368 undef)
    for (int c = 0; c < chunks; c++) {
        const int e = beg + c * 16 + lane16;
        int id = 0;
        double d = -INFINITY;
        if (e < end) {
            id = col[e];
            if (id > 0) {                                                  // :371
                d = (k0 * lz[e] + k1 * lx[e]) + k2 * ly[e];                // :376, no FMA
                if (!(d == d)) d = -INFINITY;                              // NaN never passes '>'
            }
        }
        const bool valid = d > -INFINITY;
        // inclusive prefix max over the 16 lanes of the group
        double incl = d;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const double t = __shfl_up(incl, off, 16);
            if (lane16 >= off) incl = fmax(incl, t);
        }
        double excl = __shfl_up(incl, 1, 16);
        if (lane16 == 0) excl = -INFINITY;
        const double R = fmax(D1, excl);
        const bool rec = valid && (d > R);
        const bool nonrec = valid && !rec;
        const double cmax = __shfl(incl, 15, 16);          // chunk maximum
        // slot 2 candidates: maximum over the non-records
        double m2 = nonrec ? d : -INFINITY;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) m2 = fmax(m2, __shfl_xor(m2, off, 16));
        const unsigned b1 = group_bits(__ballot(valid && d == cmax), group);
        const unsigned b2 = group_bits(__ballot(nonrec && d == m2), group);
        // shuffles stay outside the (group-divergent) branches so every lane takes part
        const int id1 = __shfl(id, b1 ? __ffs(b1) - 1 : 0, 16);   // first arg-max = last record
        const int id2 = __shfl(id, b2 ? __ffs(b2) - 1 : 0, 16);
        if (cmax > D1) {                                   // at least one record in this chunk
            D1 = cmax;
            i1 = id1;
        }
        if (m2 > D2) {                                     // strict: earlier slot-2 holder stays
            D2 = m2;
            i2 = id2;
        }
    }
    if (D2 <= 0.0) {        // :390-393
        D2 = 0.0;
        i2 = i1;
    }
    if (!in_range || lane16 != 0) return;
    if (i1 <= 0) {          // no neighbour with dot > -1: the reference reads garbage here
        up1[site] = kNoUpwind;
        up2[site] = kNoUpwind;
        d1o[site] = D1; d2o[site] = D2;
        w1o[site] = 0.0; w2o[site] = 0.0; r1o[site] = 0.0; r2o[site] = 0.0;
        return;
    }
    const double p1 = pow(D1, 7.0), p2 = pow(D2, 7.0);     // irregular_ray_tracing.jl:1,51
    const double sum = p1 + p2;
    const double pz = pos[3 * site + 0], px = pos[3 * site + 1], py = pos[3 * site + 2];
    const int64_t a = i1 - 1, b = i2 - 1;
    double dz = pz - pos[3 * a + 0], dx = px - pos[3 * a + 1], dy = py - pos[3 * a + 2];
    const double r1 = sqrt((dz * dz + dx * dx) + dy * dy); // euclidean, no periodic wrap (:66)
    dz = pz - pos[3 * b + 0]; dx = px - pos[3 * b + 1]; dy = py - pos[3 * b + 2];
    const double r2 = sqrt((dz * dz + dx * dx) + dy * dy);
    up1[site] = (int32_t)a;
    up2[site] = (int32_t)b;
    d1o[site] = D1;
    d2o[site] = D2;
    w1o[site] = p1 / sum;
    w2o[site] = p2 / sum;
    r1o[site] = r1;
    r2o[site] = r2;
}

int launch_upwind_table(vrt_plan *p, int a)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int64_t blocks = (n * 16 + 255) / 256;
    const size_t o = (size_t)a * (size_t)n;
    hipLaunchKernelGGL(k_upwind_table, dim3((unsigned)blocks), dim3(256), 0, g->stream, n,
                       g->d_rowptr, g->d_col, g->d_lz, g->d_lx, g->d_ly, g->d_pos,
                       p->k[3 * a + 0], p->k[3 * a + 1], p->k[3 * a + 2], p->d_up1 + o,
                       p->d_up2 + o, p->d_d1 + o, p->d_d2 + o, p->d_w1 + o, p->d_w2 + o,
                       p->d_r1 + o, p->d_r2 + o);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// --------------------------------------------------------------------------------------------
// Boundary: I[perm[1:n1]] = I_0 (irregular_ray_tracing.jl:31-35) and I = 0 for the one site of
// the last layer the reference never visits (voronoi_utils.jl:266).  blockIdx.y = angle slot.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_boundary(int64_t n, int64_t nlam, int64_t ldI, int64_t n1, const int32_t *__restrict__ order,
           const int32_t *__restrict__ angles, const T *__restrict__ I0, T *__restrict__ I)
{
    const int a = angles[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (n1 + 1) * nlam;
    if (t >= total) return;
    const int64_t p = t / nlam;
    const int64_t l = t - p * nlam;
    T *Ia = I + (size_t)a * (size_t)n * (size_t)ldI;
    if (p < n1) {
        const int64_t site = order[p];
        Ia[(size_t)site * ldI + l] = I0 ? I0[(size_t)p * nlam + l] : (T)0;
    } else {
        const int64_t site = order[n - 1];
        Ia[(size_t)site * ldI + l] = (T)0;
    }
}

int launch_boundary(vrt_plan *p, const SweepArgs &sa, const void *dI0_up, const void *dI0_down,
                    hipStream_t st)
{
    vrt_grid *g = p->g;
    for (int d = 0; d < 2; d++) {
        const Direction &dir = d == 0 ? g->up : g->down;
        const int cnt = d == 0 ? p->n_up : p->n_down;
        if (cnt == 0) continue;
        // when the grid has a single layer every site but the last is boundary (n1 = n - 1)
        const int64_t total = (dir.n1 + 1) * sa.nlam;
        const int64_t blocks = (total + 255) / 256;
        const int32_t *ang = d == 0 ? p->d_angles_up : p->d_angles_down;
        const void *I0 = d == 0 ? dI0_up : dI0_down;
        if (sa.f32)
            hipLaunchKernelGGL(k_boundary<float>, dim3((unsigned)blocks, (unsigned)cnt), dim3(256), 0, st, sa.n,
                               sa.nlam, sa.ldI, dir.n1, dir.d_order, ang, (const float *)I0, (float *)sa.I);
        else
            hipLaunchKernelGGL(k_boundary<double>, dim3((unsigned)blocks, (unsigned)cnt), dim3(256), 0, st, sa.n,
                               sa.nlam, sa.ldI, dir.n1, dir.d_order, ang, (const double *)I0, (double *)sa.I);
        VRT_HIP_TRY(hipGetLastError());
    }
    return VRT_OK;
}

// --------------------------------------------------------------------------------------------
// One level of the sweep schedule: every (node, wavelength) pair of the level is independent.
// Per pair, irregular_ray_tracing.jl:53-76:
//   Δτ_r = r_r (α_c + α_u) / 2              trapezoidal, functions.jl:392-395
//   (a, b, e) = linear_weights(Δτ_r)         functions.jl:484-500
//   I_c = (0 + ((e I_u1 + a S_u1) + b S_c) w_1) + ((e I_u2 + a S_u2) + b S_c) w_2
// Wavelength is the fastest index of S, α and I, so consecutive lanes read consecutive doubles
// of a site row.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_weights(double dtau, double &a, double &b, double &e)
{
    if (dtau < 5e-4) {
        e = 1.0 - dtau + 0.5 * (dtau * dtau);
        a = dtau * (0.5 - dtau / 3.0);
        b = dtau * (0.5 - dtau / 6.0);
    } else if (dtau > 50.0) {
        e = 0.0;
        a = 1.0 / dtau;
        b = 1.0 - a;
    } else {
        e = exp(-dtau);
        a = (1.0 - e) / dtau - e;
        b = 1.0 - a - e;
    }
}

// T = storage type of S, α and I (double, or float for the fp32 value path of BASELINE config C5);
// the arithmetic is always fp64.
template <typename T, int ALPHA_MODE>
__global__ void __launch_bounds__(256)
k_sweep_level(int64_t first, int count, int nlam, int64_t n, int64_t ldS, int64_t ldA, int64_t ldI,
              const uint32_t *__restrict__ node_site, const uint32_t *__restrict__ node_meta,
              const int32_t *__restrict__ node_u1, const int32_t *__restrict__ node_u2,
              const double *__restrict__ w1, const double *__restrict__ w2,
              const double *__restrict__ r1, const double *__restrict__ r2,
              const T *__restrict__ S, const T *__restrict__ alpha, T *I)
{
    // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so
    // blocks b, b+8, b+16, ... -- one XCD -- take a CONTIGUOUS eighth of the Morton-sorted node
    // list; neighbouring sites then share upwind rows through that XCD's L2.  Bijective remap
    // (cdna_hip_programming.md, "XCD swizzle must be bijective"); placement only affects speed.
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned qd = nb >> 3, rm = nb & 7u;
    const unsigned chunk = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    const unsigned t = chunk * blockDim.x + threadIdx.x;
    const unsigned q = nlam == 1 ? t : t / (unsigned)nlam;
    if (q >= (unsigned)count) return;
    const unsigned l = nlam == 1 ? 0u : t - q * (unsigned)nlam;
    const uint32_t site = node_site[first + q];
    const uint32_t meta = node_meta[first + q];
    const unsigned a = meta & 0xFFu;
    const size_t row = (size_t)a * (size_t)n + site;
    const int32_t u1 = node_u1[first + q], u2 = node_u2[first + q];   // (= up1[row], up2[row]: no dependent lookup)
    const double W1 = w1[row], W2 = w2[row], R1 = r1[row], R2 = r2[row];

    const T *Aa = alpha;
    if (ALPHA_MODE == VRT_ALPHA_ANGLE_SITE_LAM) Aa += (size_t)a * (size_t)n * (size_t)ldA;
    T *Ia = I + (size_t)a * (size_t)n * (size_t)ldI;

    double a_c, a_1, a_2;
    if (ALPHA_MODE == VRT_ALPHA_SITE) {
        a_c = Aa[site]; a_1 = Aa[u1]; a_2 = Aa[u2];
    } else {
        a_c = Aa[(size_t)site * ldA + l];
        a_1 = Aa[(size_t)u1 * ldA + l];
        a_2 = Aa[(size_t)u2 * ldA + l];
    }
    const double S_c = S[(size_t)site * ldS + l];
    const double S_1 = S[(size_t)u1 * ldS + l];
    const double S_2 = S[(size_t)u2 * ldS + l];
    const double I_1 = (meta & 0x100u) ? 0.0 : (double)Ia[(size_t)u1 * ldI + l];
    const double I_2 = (meta & 0x200u) ? 0.0 : (double)Ia[(size_t)u2 * ldI + l];

    double ca, cb, ce;
    linear_weights(R1 * (a_c + a_1) / 2.0, ca, cb, ce);
    const double t1 = ((ce * I_1 + ca * S_1) + cb * S_c) * W1;
    linear_weights(R2 * (a_c + a_2) / 2.0, ca, cb, ce);
    const double t2 = ((ce * I_2 + ca * S_2) + cb * S_c) * W2;
    Ia[(size_t)site * ldI + l] = (T)((0.0 + t1) + t2);
}

int launch_sweep_levels(vrt_plan *p, const SweepArgs &sa, hipStream_t st, int64_t *launches)
{
    const int64_t nlev = (int64_t)p->level_off.size() - 1;
    int64_t nl = 0;
    // keep the flat (node, λ) index of one launch inside 31 bits
    const int64_t max_nodes = std::max<int64_t>(1, ((int64_t)1 << 30) / sa.nlam);
    for (int64_t lev = 0; lev < nlev; lev++) {
        int64_t first = p->level_off[(size_t)lev];
        const int64_t last = p->level_off[(size_t)lev + 1];
        while (first < last) {
            const int64_t cnt = std::min(max_nodes, last - first);
            const int64_t blocks = (cnt * sa.nlam + 255) / 256;
#define VRT_LAUNCH_SWEEP_T(TT, MODE)                                                               \
    hipLaunchKernelGGL((k_sweep_level<TT, MODE>), dim3((unsigned)blocks), dim3(256), 0, st, first, \
                       (int)cnt, (int)sa.nlam, sa.n, sa.ldS, sa.ldA, sa.ldI, p->d_node_site,       \
                       p->d_node_meta, p->d_node_u1, p->d_node_u2, p->d_w1, p->d_w2, p->d_r1, p->d_r2, \
                       (const TT *)sa.S, (const TT *)sa.alpha, (TT *)sa.I)
#define VRT_LAUNCH_SWEEP(MODE)                                                                     \
    do {                                                                                           \
        if (sa.f32) VRT_LAUNCH_SWEEP_T(float, MODE);                                               \
        else VRT_LAUNCH_SWEEP_T(double, MODE);                                                     \
    } while (0)
            if (sa.alpha_mode == VRT_ALPHA_SITE) VRT_LAUNCH_SWEEP(VRT_ALPHA_SITE);
            else if (sa.alpha_mode == VRT_ALPHA_SITE_LAM) VRT_LAUNCH_SWEEP(VRT_ALPHA_SITE_LAM);
            else VRT_LAUNCH_SWEEP(VRT_ALPHA_ANGLE_SITE_LAM);
#undef VRT_LAUNCH_SWEEP
#undef VRT_LAUNCH_SWEEP_T
            nl++;
            first += cnt;
        }
    }
    VRT_HIP_TRY(hipGetLastError());
    *launches = nl;
    return VRT_OK;
}

// --------------------------------------------------------------------------------------------
// J = Σ_angles w · I, accumulated in the reference's angle order (lambda_iteration.jl:84,102,107)
// --------------------------------------------------------------------------------------------
struct WeightTable {
    double w[kMaxAngles];
};

template <typename T>
__global__ void __launch_bounds__(256)
k_reduce_J(int64_t n, int64_t nlam, int64_t ldI, int64_t ldJ, int A, WeightTable wt,
           const T *__restrict__ I, T *__restrict__ J)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nlam) return;
    const int64_t site = t / nlam;
    const int64_t l = t - site * nlam;
    double acc = 0.0;
    for (int a = 0; a < A; a++)
        acc += wt.w[a] * (double)I[((size_t)a * (size_t)n + (size_t)site) * (size_t)ldI + l];
    J[(size_t)site * ldJ + l] = (T)acc;
}

int launch_reduce_J(vrt_plan *p, const SweepArgs &sa, const double *weights_active, void *dJ,
                    int64_t ldJ, hipStream_t st)
{
    WeightTable wt;
    for (int a = 0; a < kMaxAngles; a++) wt.w[a] = a < p->A ? weights_active[a] : 0.0;
    const int64_t total = sa.n * sa.nlam;
    const int64_t blocks = (total + 255) / 256;
    if (sa.f32)
        hipLaunchKernelGGL(k_reduce_J<float>, dim3((unsigned)blocks), dim3(256), 0, st, sa.n, sa.nlam, sa.ldI,
                           ldJ, p->A, wt, (const float *)sa.I, (float *)dJ);
    else
        hipLaunchKernelGGL(k_reduce_J<double>, dim3((unsigned)blocks), dim3(256), 0, st, sa.n, sa.nlam, sa.ldI,
                           ldJ, p->A, wt, (const double *)sa.I, (double *)dJ);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// per-angle intensities to the caller's (nlam, n, n_angles) array; skipped (θ = 90) angles -> 0
template <typename T>
__global__ void __launch_bounds__(256)
k_copy_I(int64_t n, int64_t nlam, int64_t ldI, int64_t ldO, int src_angle,
         const T *__restrict__ I, T *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nlam) return;
    const int64_t site = t / nlam;
    const int64_t l = t - site * nlam;
    out[(size_t)site * ldO + l] =
        src_angle < 0 ? (T)0 : I[((size_t)src_angle * (size_t)n + (size_t)site) * (size_t)ldI + l];
}

int launch_copy_I_out(vrt_plan *p, const SweepArgs &sa, void *dI_out, int64_t ldO, hipStream_t st)
{
    std::vector<int> active_of_user((size_t)p->n_angles_user, -1);
    for (int a = 0; a < p->A; a++) active_of_user[(size_t)p->user_of_active[(size_t)a]] = a;
    const int64_t total = sa.n * sa.nlam;
    const int64_t blocks = (total + 255) / 256;
    for (int64_t u = 0; u < p->n_angles_user; u++) {
        const size_t off = (size_t)u * (size_t)sa.n * (size_t)ldO;
        if (sa.f32)
            hipLaunchKernelGGL(k_copy_I<float>, dim3((unsigned)blocks), dim3(256), 0, st, sa.n, sa.nlam, sa.ldI,
                               ldO, active_of_user[(size_t)u], (const float *)sa.I, (float *)dI_out + off);
        else
            hipLaunchKernelGGL(k_copy_I<double>, dim3((unsigned)blocks), dim3(256), 0, st, sa.n, sa.nlam, sa.ldI,
                               ldO, active_of_user[(size_t)u], (const double *)sa.I, (double *)dI_out + off);
    }
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// --------------------------------------------------------------------------------------------
// Λ-iteration epilogue (the part that needs no physics library):
//   S_new[l, i] = (1 - ε_i) J[l, i] + ε_i B[l, i]                 lambda_iteration.jl:261-263
//   diff = max_{l,i} |1 - S_old[l, i] / S_new[l, i]|                criterion, :325-349
// One pass over (site, λ); the maximum is reduced per wave with shuffles and merged with an
// integer atomicMax on the IEEE bits (all candidates are >= 0, so the bit pattern orders like
// the value); a NaN anywhere makes the result NaN as Julia's `maximum` does.
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lambda_update(int64_t n, int64_t nlam, int64_t ld, const double *__restrict__ J,
                const double *__restrict__ B, const double *__restrict__ eps,
                const double *__restrict__ S_old, double *__restrict__ S_new,
                unsigned long long *__restrict__ result /* [0] max bits, [1] NaN flag */)
{
    // grid-stride over the n * nlam elements; ONE atomic per workgroup (an atomic per wave on a single
    // address serialises at the memory side: 9.0 ms for C4's 50.8 M elements against 0.5 ms of traffic)
    __shared__ double wmax[4];
    __shared__ int wnan[4];
    const int64_t total = n * nlam;
    double d = 0.0;
    bool isnan_ = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t site = t / nlam, l = t - site * nlam;
        const size_t o = (size_t)site * ld + l;
        const double e = eps[site];
        const double s_new = (1.0 - e) * J[o] + e * B[o];
        S_new[o] = s_new;
        const double dd = fabs(1.0 - S_old[o] / s_new);
        if (!(dd == dd)) isnan_ = true;
        else d = fmax(d, dd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = fmax(d, __shfl_xor(d, off, 64));
    const unsigned long long any_nan = __ballot(isnan_);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wmax[wave] = d; wnan[wave] = any_nan != 0ull; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        atomicMax(&result[0], (unsigned long long)__double_as_longlong(m));     // m >= 0: bit order = value order
        if (wnan[0] | wnan[1] | wnan[2] | wnan[3]) atomicMax(&result[1], 1ull);
    }
}

// The same on sweep-order planes (vrt_plan_execute_native_dev): one thread per (wavelength pair, up position).  J = J_up +
// J_down as k_combine_J forms it, S_new = (1 - ε) J + ε B with B in the up order, the old S read from the plane it is
// written back to, the down-order copy of S_new written beside it -- the operations of k_lambda_update on the same values,
// so S, J and the criterion are those of the caller-layout loop bit for bit, without either layout change.
__global__ void __launch_bounds__(256)
k_lambda_update_native(int64_t n, int npair, int nlam, const int32_t *__restrict__ store_up, const int32_t *__restrict__ rank_down,
                       const double2 *__restrict__ Ju, const double2 *__restrict__ Jd, const double2 *__restrict__ Bu,
                       const double *__restrict__ eps, double2 *__restrict__ Su, double2 *__restrict__ Sd,
                       unsigned long long *__restrict__ result)
{
    // one thread per UP position, walking the wavelength pairs: the site, its down position and its ε are looked up once; per
    // pair four loads and two stores, independent from pair to pair (several in flight).  Neighbouring lanes are neighbouring
    // up positions: J_up, B, S_up are contiguous, the down-order accesses piecewise contiguous (a layer is a layer in both orders)
    __shared__ double wmax[4];
    __shared__ int wnan[4];
    double d = 0.0;
    bool isnan_ = false;
    for (int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += (int64_t)gridDim.x * blockDim.x) {
        const int32_t site = store_up[pos];
        const size_t pd = (size_t)rank_down[site];
        const double e = eps[site];
        constexpr int U = 4;
        for (int q0 = 0; q0 < npair; q0 += U) {
            double2 J[U], B[U], So[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u < npair ? q0 + u : npair - 1;
                const size_t t = (size_t)q * (size_t)n + (size_t)pos;
                J[u] = make_double2(0.0, 0.0);
                if (Ju) J[u] = Ju[t];
                if (Jd) {
                    const double2 v = Jd[(size_t)q * (size_t)n + pd];
                    J[u].x = J[u].x + v.x; J[u].y = J[u].y + v.y;
                }
                B[u] = Bu[t];
                So[u] = Su[t];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u;
                if (q >= npair) break;
                const size_t t = (size_t)q * (size_t)n + (size_t)pos;
                double2 Sn;
                Sn.x = (1.0 - e) * J[u].x + e * B[u].x;
                Sn.y = (1.0 - e) * J[u].y + e * B[u].y;
                const bool second = 2 * q + 1 < nlam;              // (an odd count: the padding wavelength is carried as zeros)
                if (!second) Sn.y = 0.0;
                Su[t] = Sn;
                Sd[(size_t)q * (size_t)n + pd] = Sn;
                const double dx = fabs(1.0 - So[u].x / Sn.x);
                if (!(dx == dx)) isnan_ = true;
                else d = fmax(d, dx);
                if (second) {
                    const double dy = fabs(1.0 - So[u].y / Sn.y);
                    if (!(dy == dy)) isnan_ = true;
                    else d = fmax(d, dy);
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = fmax(d, __shfl_xor(d, off, 64));
    const unsigned long long any_nan = __ballot(isnan_);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wmax[wave] = d; wnan[wave] = any_nan != 0ull; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        atomicMax(&result[0], (unsigned long long)__double_as_longlong(m));
        if (wnan[0] | wnan[1] | wnan[2] | wnan[3]) atomicMax(&result[1], 1ull);
    }
}

int launch_lambda_update_native(vrt_grid *g, int64_t nlam, const double *dJ_up, const double *dJ_down, const double *dB_up,
                                const double *deps, double *dS_up, double *dS_down, unsigned long long *d_result, hipStream_t st)
{
    VRT_HIP_TRY(hipMemsetAsync(d_result, 0, 2 * sizeof(unsigned long long), st));
    const int npair = (int)((nlam + 1) / 2);
    const int64_t blocks = std::min<int64_t>((g->n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_lambda_update_native, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), 0, st, g->n, npair, (int)nlam,
                       g->up.d_store, g->down.d_srank, reinterpret_cast<const double2 *>(dJ_up), reinterpret_cast<const double2 *>(dJ_down),
                       reinterpret_cast<const double2 *>(dB_up), deps, reinterpret_cast<double2 *>(dS_up),
                       reinterpret_cast<double2 *>(dS_down), d_result);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// dst += src (partial J's of two handles on one device: vrt_multi's same-device rehearsal)
__global__ void __launch_bounds__(256)
k_axpy(size_t count, const double *__restrict__ src, double *__restrict__ dst)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) dst[t] += src[t];
}

int launch_axpy(size_t count, const double *d_src, double *d_dst, hipStream_t st)
{
    if (!count) return VRT_OK;
    hipLaunchKernelGGL(k_axpy, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, d_src, d_dst);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// dst[j][l] = src[order[j]][l]: rows of a caller-layout (nlam, n) array picked by a site list (the boundary
// intensity B_λ(T) of the bottom layer in perm_up order, lambda_iteration.jl:99-101)
__global__ void __launch_bounds__(256)
k_gather_rows(int64_t rows, int64_t nlam, int64_t ld, const int32_t *__restrict__ order, const double *__restrict__ src,
              double *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * nlam) return;
    const int64_t j = t / nlam, l = t % nlam;
    dst[t] = src[(size_t)order[j] * (size_t)ld + (size_t)l];
}

int launch_gather_rows(int64_t rows, int64_t nlam, int64_t ld, const int32_t *d_order, const double *d_src, double *d_dst,
                       hipStream_t st)
{
    if (rows * nlam <= 0) return VRT_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((rows * nlam + 255) / 256)), dim3(256), 0, st, rows, nlam, ld, d_order,
                       d_src, d_dst);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int launch_lambda_update(int64_t n, int64_t nlam, int64_t ld, const double *dJ, const double *dB,
                         const double *deps, const double *dS_old, double *dS_new,
                         unsigned long long *d_result, hipStream_t st)
{
    VRT_HIP_TRY(hipMemsetAsync(d_result, 0, 2 * sizeof(unsigned long long), st));
    const int64_t total = n * nlam;
    const int64_t blocks = std::min<int64_t>((total + 255) / 256, 256 * 16);     // 16 workgroups per CU
    hipLaunchKernelGGL(k_lambda_update, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), 0, st, n, nlam, ld,
                       dJ, dB, deps, dS_old, dS_new, d_result);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

}  // namespace vrt
