// Layer-tile formal solver for gfx950: ONE workgroup per (angle, wavelength) problem walks the
// BFS layers itself; the intensities of the layer being solved live in an LDS tile, the
// per-site integration coefficients live in registers across the layer's Gauss-Seidel levels,
// and ordering inside a layer is `s_barrier` instead of a kernel boundary.  No inter-workgroup
// communication exists (every (angle, λ) solve is independent: lambda_iteration.jl:84-111), so
// there is nothing to deadlock and no cross-XCD coherence to manage.
//
// Data is held in SWEEP ORDER (position in perm_up / perm_down) and wavelength-major
// ([λ][pos]), so a layer is a contiguous range of every array: the centre streams (S, α, table)
// are perfectly coalesced and the upwind gathers stay inside the previous few layers' ranges.
// The caller's (nλ, n) arrays are transposed into that layout by LDS-tiled kernels.
//
// Arithmetic: a visit of the reference computes
//     I_c = ((e1 I_u1 + a1 S_u1) + b1 S_c) w1 + ((e2 I_u2 + a2 S_u2) + b2 S_c) w2
// (irregular_ray_tracing.jl:73-76).  Here the I-independent part is folded once per site into
// c and the in-layer couplings into g_r = e_r w_r, so later visits cost two LDS reads and two
// multiply-adds.  The re-association changes results at the 1e-16 level (contract: 1e-10).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vrt_device.h"
#include "vrt_internal.h"

namespace vrt {

// ---- table in sweep order -----------------------------------------------------------------------
// t_u1/t_u2: sweep positions of the upwind sites; everything else copied from the site-order table.
__global__ void __launch_bounds__(256)
k_permute_table(int64_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ rank,
                const int32_t *__restrict__ up1, const int32_t *__restrict__ up2,
                const double *__restrict__ w1, const double *__restrict__ w2,
                const double *__restrict__ r1, const double *__restrict__ r2,
                const uint32_t *__restrict__ vis, const int32_t *__restrict__ lay, int nlayers,
                int32_t *__restrict__ t_u1, int32_t *__restrict__ t_u2, double *__restrict__ t_w1,
                double *__restrict__ t_w2, double *__restrict__ t_r1, double *__restrict__ t_r2,
                uint32_t *__restrict__ t_vis, uint32_t *__restrict__ t_loc)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t s = order[p];
    const int32_t a = up1[s], b = up2[s];
    const int32_t ua = a >= 0 ? rank[a] : -1, ub = b >= 0 ? rank[b] : -1;
    t_u1[p] = ua;
    t_u2[p] = ub;
    // in-layer tile slots of the two upwinds (16 bits each; 0 when the upwind is not in the
    // site's own layer -- its coupling coefficient is 0 then): depends on (angle, site) only
    int lo_i = 0, hi_i = nlayers;                 // layer l = [lay[l-1], lay[l]): find l with p inside
    while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (lay[mid] <= p) lo_i = mid; else hi_i = mid;
    }
    const int lo = lay[lo_i], hi = lay[lo_i + 1];
    // kNoSlot: the kernels read a dedicated zero slot instead (coupling 0 times a finite 0, so an Inf
    // or NaN elsewhere in the layer stays where the reference keeps it)
    const uint32_t l1 = (ua >= lo && ua < hi) ? (uint32_t)(ua - lo) : kNoSlot;
    const uint32_t l2 = (ub >= lo && ub < hi) ? (uint32_t)(ub - lo) : kNoSlot;
    t_loc[p] = l1 | (l2 << 16);
    t_w1[p] = w1[s];
    t_w2[p] = w2[s];
    t_r1[p] = r1[s];
    t_r2[p] = r2[s];
    t_vis[p] = vis[s];
}

int launch_permute_table(vrt_plan *p, int a, const uint32_t *d_vis_site)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
    const size_t o = (size_t)a * (size_t)n;
    hipLaunchKernelGGL(k_permute_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g->stream, n,
                       dir.d_store, dir.d_srank, p->d_up1 + o, p->d_up2 + o, p->d_w1 + o, p->d_w2 + o,
                       p->d_r1 + o, p->d_r2 + o, d_vis_site, dir.d_lay, (int)dir.reduced.size() - 1,
                       p->t_u1 + o, p->t_u2 + o, p->t_w1 + o, p->t_w2 + o, p->t_r1 + o, p->t_r2 + o,
                       p->t_vis + o, p->t_loc + o);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// visit levels and tile slots in the sorted thread order of k_step_levels (build_sorted_slots)
__global__ void __launch_bounds__(256)
k_sorted_tables(int64_t n, const int32_t *__restrict__ self, const uint32_t *__restrict__ t_vis,
                const uint32_t *__restrict__ t_loc, uint32_t *__restrict__ vis_s, uint32_t *__restrict__ loc_s,
                int32_t *__restrict__ rank_s)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = self[i];
    vis_s[i] = t_vis[p];
    loc_s[i] = t_loc[p];
    rank_s[p] = (int32_t)i;              // storage position -> sorted index (absolute)
}

// the single-wavelength level kernel keeps its LDS tile in SORTED order (a thread's write address is
// then its own index, no per-site register): the two upwind tile slots of sorted entry i, also in
// sorted terms
__global__ void __launch_bounds__(256)
k_sorted_loc(int64_t n, const int32_t *__restrict__ lay, int nlayers, const uint32_t *__restrict__ loc_s,
             const int32_t *__restrict__ rank_s, uint32_t *__restrict__ loc_ss)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo_i = 0, hi_i = nlayers;                  // layer of sorted index i (sorting stays inside layers)
    while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (lay[mid] <= i) lo_i = mid; else hi_i = mid;
    }
    const int lo = lay[lo_i];
    const uint32_t l = loc_s[i];
    const uint32_t l1 = l & 0xFFFFu, l2 = l >> 16;
    const uint32_t s1 = l1 == kNoSlot ? kNoSlot : (uint32_t)(rank_s[lo + (int)l1] - lo);
    const uint32_t s2 = l2 == kNoSlot ? kNoSlot : (uint32_t)(rank_s[lo + (int)l2] - lo);
    loc_ss[i] = s1 | (s2 << 16);
}

// upwind slot + kind codes of the two-launch tile path (k_sweep_tiles_pre; layout described there)
__global__ void __launch_bounds__(256)
k_sorted_code(int64_t n, const int32_t *__restrict__ lay, int nlayers, const int32_t *__restrict__ self,
              const int32_t *__restrict__ rank_s, const int32_t *__restrict__ t_u1,
              const int32_t *__restrict__ t_u2, uint32_t *__restrict__ code_ss)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo_i = 0, hi_i = nlayers;                  // layer of sorted index i (sorting stays inside layers)
    while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (lay[mid] <= i) lo_i = mid; else hi_i = mid;
    }
    const int lo = lay[lo_i], hi = lay[lo_i + 1];
    const int lop = lo_i > 0 ? lay[lo_i - 1] : 0;
    const int p = self[i];
    uint32_t code = 0;
    for (int r = 0; r < 2; r++) {
        const int u = r == 0 ? t_u1[p] : t_u2[p];
        uint32_t c = 0;
        if (u >= lo && u < hi) c = (uint32_t)(rank_s[u] - lo) | (1u << 12);
        else if (lo_i > 0 && u >= lop && u < lo) c = (uint32_t)(rank_s[u] - lop) | (2u << 12);
        code |= c << (14 * r);
    }
    code_ss[i] = code;
}

int launch_sorted_tables(vrt_plan *p, int a)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const size_t o = (size_t)a * (size_t)n;
    const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(k_sorted_tables, grid, dim3(256), 0, g->stream, n, p->t_self + o, p->t_vis + o,
                       p->t_loc + o, p->t_vis_s + o, p->t_loc_s + o, p->t_rank_s + o);
    hipLaunchKernelGGL(k_sorted_loc, grid, dim3(256), 0, g->stream, n, dir.d_lay, (int)dir.reduced.size() - 1,
                       p->t_loc_s + o, p->t_rank_s + o, p->t_loc_ss + o);
    if (p->t_code_ss)
        hipLaunchKernelGGL(k_sorted_code, grid, dim3(256), 0, g->stream, n, dir.d_lay, (int)dir.reduced.size() - 1,
                           p->t_self + o, p->t_rank_s + o, p->t_u1 + o, p->t_u2 + o, p->t_code_ss + o);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// Compact list of the in-layer couplings of a layer.  g_r = e_r w_r is nonzero only when upwind r
// lies in the site's own layer (C4: 1.17 of the 2 per site on average), so the layer-step
// kernels exchange the couplings as a dense list per (angle, wavelength pair, layer):
//   t_gpos[a][p] = position of the site's first in-layer coupling in that list (exclusive prefix
//                  count over the layer's storage order) | in1 << 30 | in2 << 31
// one workgroup per (layer, angle); a thread scans ceil(cnt / 1024) consecutive slots.
__global__ void __launch_bounds__(1024)
k_gpos(int64_t n, const int32_t *__restrict__ lay, int nlayers, const int32_t *__restrict__ t_u1,
       const int32_t *__restrict__ t_u2, uint32_t *__restrict__ gpos)
{
    __shared__ int part[1024];
    const int layer = blockIdx.x + 1;              // 1-based; layer 1 (boundary) has no visits
    if (layer > nlayers) return;
    const int lo = lay[layer - 1], hi = lay[layer], cnt = hi - lo;
    const int tid = threadIdx.x;
    const int per = (cnt + 1023) / 1024;
    int sum = 0;
    for (int j = 0; j < per; j++) {
        const int s = tid * per + j;
        if (s < cnt) {
            const int u1 = t_u1[lo + s], u2 = t_u2[lo + s];
            sum += (int)((u1 >= lo) & (u1 < hi)) + (int)((u2 >= lo) & (u2 < hi));
        }
    }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {     // inclusive Hillis-Steele scan of the thread sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int pos = part[tid] - sum;
    for (int j = 0; j < per; j++) {
        const int s = tid * per + j;
        if (s < cnt) {
            const int u1 = t_u1[lo + s], u2 = t_u2[lo + s];
            const uint32_t in1 = (u1 >= lo) & (u1 < hi), in2 = (u2 >= lo) & (u2 < hi);
            gpos[lo + s] = (uint32_t)pos | (in1 << 30) | (in2 << 31);
            pos += (int)(in1 + in2);
        }
    }
}

int launch_gpos(vrt_plan *p, int a)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
    const size_t o = (size_t)a * (size_t)n;
    const int nlayers = (int)dir.reduced.size() - 1;
    VRT_HIP_TRY(hipMemsetAsync(p->t_gpos + o, 0, sizeof(uint32_t) * (size_t)n, g->stream));
    if (nlayers >= 1)
        hipLaunchKernelGGL(k_gpos, dim3((unsigned)nlayers), dim3(1024), 0, g->stream, n, dir.d_lay, nlayers,
                           p->t_u1 + o, p->t_u2 + o, p->t_gpos + o);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// out[l][p] = in[order[p]][l]   (caller's (nλ, n) site-major rows -> wavelength-major sweep order)
template <typename T>
__global__ void __launch_bounds__(256)
k_to_sweep_order(int64_t n, int nlam, int64_t ld, int lb, const int32_t *__restrict__ order,
                 const T *__restrict__ in, T *__restrict__ out)
{
    __shared__ T tile[64][65];
    __shared__ int32_t rows[64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    if (threadIdx.x < 64) rows[threadIdx.x] = p0 + threadIdx.x < n ? order[p0 + threadIdx.x] : -1;
    __syncthreads();
    T v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {          // 16 independent row reads in flight per thread
        const int32_t site = rows[ty + 4 * j];
        v[j] = (site >= 0 && l0 + tx < nlam) ? in[(size_t)site * ld + l0 + tx] : (T)0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) tile[ty + 4 * j][tx] = v[j];
    __syncthreads();
    if (lb == 1) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int l = l0 + ty + 4 * j;
            if (l < nlam && p0 + tx < n) out[(size_t)l * n + p0 + tx] = tile[tx][ty + 4 * j];
        }
    } else {                                   // one 16-byte store per (site, wavelength pair)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = 2 * (ty + 4 * j), l = l0 + c;
            if (l < nlam && p0 + tx < n) {
                typename Pair<T>::type v2;
                v2.x = tile[tx][c];
                v2.y = l + 1 < nlam ? tile[tx][c + 1] : (T)0;
                reinterpret_cast<typename Pair<T>::type *>(out)[(size_t)(l >> 1) * (size_t)n + (size_t)(p0 + tx)] = v2;
            }
        }
    }
}

// out[p] = in[order[p]]   (per-site vector, e.g. wavelength-independent α)
template <typename T>
__global__ void __launch_bounds__(256)
k_gather_vec(int64_t n, const int32_t *__restrict__ order, const T *__restrict__ in,
             T *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[order[p]];
}

// I[a][l][p] = I0[p][l] for p < n1 (boundary layer, already in sweep order), blockIdx.z = angle slot
template <typename T>
__global__ void __launch_bounds__(256)
k_boundary_sweep_order(int64_t n, int nlam, int lb, int64_t n1, const int32_t *__restrict__ angles,
                       const int32_t *__restrict__ order, const int32_t *__restrict__ srank,
                       const T *__restrict__ I0, T *__restrict__ I)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    const int a = angles[blockIdx.z];
    for (int r = ty; r < 64; r += 4) {
        const int64_t p = p0 + r;
        if (p < n1 && l0 + tx < nlam) tile[r][tx] = I0 ? I0[(size_t)p * nlam + l0 + tx] : (T)0;
    }
    __syncthreads();
    const int nl_pad = (nlam + lb - 1) / lb * lb;
    T *Ia = I + (size_t)a * (size_t)nl_pad * (size_t)n;
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        // I_0 is ordered like perm[1:n1] (irregular_ray_tracing.jl:33); storage is Morton order
        if (l < nlam && p0 + tx < n1) {
            const int32_t pos = srank[order[p0 + tx]];
            Ia[sw_index(l, pos, n, lb)] = tile[tx][c];
            if (l == nlam - 1 && nl_pad > nlam) Ia[sw_index(nlam, pos, n, lb)] = (T)0;   // padding wavelength
        }
        // the never-visited site perm[n] (storage position n-1) keeps I = 0 (voronoi_utils.jl:266)
        // -- also on a single-layer grid, where no layer kernel ever runs
        if (blockIdx.x == 0 && tx == 0 && l < nl_pad) Ia[sw_index(l, n - 1, n, lb)] = (T)0;
    }
}


// K sites per thread, phase-1 batches of B sites (their 12 B loads are in flight together), T threads
// (768 = 3 waves per SIMD leaves 168 VGPRs per thread for B = 4; 1024 allows B = 2)
template <int K, int B, int T>
__global__ void __launch_bounds__(T)
k_sweep_tiles(TileArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];   // I of the current layer, then the constant terms
    double *cst = tile + ta.tile_stride;
    const int tid = threadIdx.x;
    const int task = blockIdx.x;
    const int a = ta.task_map[task] & 0xFF;
    const int l = ta.task_map[task] >> 8;
    const int d = ta.angle_dir[a];
    const int64_t n = ta.n;
    const size_t tab = (size_t)a * (size_t)n;
    const int32_t *__restrict__ tu1 = ta.t_u1 + tab;
    const int32_t *__restrict__ tu2 = ta.t_u2 + tab;
    const double *__restrict__ tw1 = ta.t_w1 + tab;
    const double *__restrict__ tw2 = ta.t_w2 + tab;
    const double *__restrict__ tr1 = ta.t_r1 + tab;
    const double *__restrict__ tr2 = ta.t_r2 + tab;
    const uint32_t *__restrict__ tvis = ta.t_vis + tab;
    const double *__restrict__ S = ta.S[d] + (size_t)l * (size_t)n;
    const double *__restrict__ Al =
        ta.alpha_mode == VRT_ALPHA_SITE ? ta.alpha[d]
        : ta.alpha_mode == VRT_ALPHA_SITE_LAM ? ta.alpha[d] + (size_t)l * (size_t)n
                                              : ta.alpha_angle + ((size_t)a * ta.nlam + l) * (size_t)n;
    double *I = ta.I + ((size_t)a * ta.nlam + l) * (size_t)n;   // written and re-read by this WG only
    const int32_t *__restrict__ lay = ta.lay[d];
    const int32_t *__restrict__ nlev = ta.nlev + (size_t)a * (size_t)(ta.max_layers + 1);
    const int L = ta.nlayers[d];

    long long cyc1 = 0, cyc2 = 0, cyc3 = 0;
    const bool timing = kDiag && ta.dbg != nullptr;
    for (int layer = 2; layer <= L; layer++) {          // irregular_ray_tracing.jl:37
        long long t0 = timing ? clock64() : 0;
        const int lo = lay[layer - 1], hi = lay[layer];  // hi of the last layer = n-1: perm[n] is never visited
        const int cnt = hi - lo;
        double g1[K], g2[K];      // in-layer couplings e_r w_r (registers); the constant term c sits in LDS
        uint32_t loc[K];        // in-layer tile slots of the two upwinds, 16 bits each
        uint32_t vis[K];
        // ---- phase 1: coefficients of every site of the layer (global reads -> registers).
        // Straight-line, branch-free batches of B sites so that the 12 B independent loads of a
        // batch are in flight together (the dependent chain table -> gathers is paid per batch,
        // not per site); invalid slots are clamped to the layer's last site and masked via vis.
        // software pipeline: the table entries (upwind positions) of batch b+1 are requested
        // while batch b's data loads are in flight, so only the first batch of a layer pays the
        // dependent table -> gather latency.
        int nu1[B], nu2[B];
        uint32_t nvis[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int slot = tid + j * T;
            const int p = lo + min(slot, cnt - 1);
            nu1[j] = ldi(tu1, p);
            nu2[j] = ldi(tu2, p);
            nvis[j] = slot < cnt ? ldu(tvis, p) : 0u;
        }
#pragma unroll
        for (int kb = 0; kb < K; kb += B) {
            int pp[B], uu1[B], uu2[B];
#pragma unroll
            for (int j = 0; j < B; j++) {
                const int slot = tid + (kb + j) * T;
                pp[j] = lo + min(slot, cnt - 1);
                uu1[j] = nu1[j];
                uu2[j] = nu2[j];
                vis[kb + j] = nvis[j];
            }
            double w1[B], w2[B], r1[B], r2[B], S_c[B], a_c[B], S_1[B], a_1[B], S_2[B], a_2[B], I_1[B], I_2[B];
#pragma unroll
            for (int j = 0; j < B; j++) {
                const int p = pp[j], u1 = uu1[j], u2 = uu2[j];
                w1[j] = ldd(tw1, p); w2[j] = ldd(tw2, p); r1[j] = ldd(tr1, p); r2[j] = ldd(tr2, p);
                S_c[j] = ldd(S, p); a_c[j] = ldd(Al, p);
                S_1[j] = ldd(S, u1); a_1[j] = ldd(Al, u1);
                S_2[j] = ldd(S, u2); a_2[j] = ldd(Al, u2);
                I_1[j] = ldd(I, min(u1, lo - 1));  // only used when u1 < lo (earlier layer: final)
                I_2[j] = ldd(I, min(u2, lo - 1));
            }
            if (kb + B < K) {
#pragma unroll
                for (int j = 0; j < B; j++) {
                    const int slot = tid + (kb + B + j) * T;
                    const int p = lo + min(slot, cnt - 1);
                    nu1[j] = ldi(tu1, p);
                    nu2[j] = ldi(tu2, p);
                    nvis[j] = slot < cnt ? ldu(tvis, p) : 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < B; j++) {
                const int u1 = uu1[j], u2 = uu2[j];
                double ca, cb, ce;
                lin_weights(r1[j] * (a_c[j] + a_1[j]) / 2.0, ca, cb, ce);   // trapezoidal, functions.jl:393
                const bool early1 = u1 < lo, in1 = (u1 >= lo) & (u1 < hi);   // else: later layer / perm[n] reads 0
                const double t1 = early1 ? ((ce * I_1[j] + ca * S_1[j]) + cb * S_c[j]) * w1[j]
                                         : (ca * S_1[j] + cb * S_c[j]) * w1[j];
                const double gg1 = in1 ? ce * w1[j] : 0.0;
                lin_weights(r2[j] * (a_c[j] + a_2[j]) / 2.0, ca, cb, ce);
                const bool early2 = u2 < lo, in2 = (u2 >= lo) & (u2 < hi);
                const double t2 = early2 ? ((ce * I_2[j] + ca * S_2[j]) + cb * S_c[j]) * w2[j]
                                         : (ca * S_2[j] + cb * S_c[j]) * w2[j];
                const double gg2 = in2 ? ce * w2[j] : 0.0;
                g1[kb + j] = gg1;
                g2[kb + j] = gg2;
                // an upwind outside the layer reads the zero slot tile[cnt] (coupling 0 x finite 0)
                loc[kb + j] = (in1 ? (uint32_t)(u1 - lo) : (uint32_t)cnt) | ((in2 ? (uint32_t)(u2 - lo) : (uint32_t)cnt) << 16);
                const int slot = tid + (kb + j) * T;
                if (slot < cnt) {
                    tile[slot] = 0.0;                                        // I = zero(S), :23
                    cst[slot] = t1 + t2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);    // keep the batches apart: hoisting more loads spills
        }
        if (tid == 0) tile[cnt] = 0.0;            // the zero slot
        __syncthreads();
        long long t1c = timing ? clock64() : 0;
        // ---- phase 2: the layer's Gauss-Seidel levels on the LDS tile ------------------------
        const int nl = nlev[layer];
        for (int t = 1; t <= nl; t++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                // a site's visits come at increasing levels: the low byte is the next one
                if ((vis[k] & 0xFFu) == (uint32_t)t) {
                    tile[tid + k * T] = cst[tid + k * T] + g1[k] * tile[loc[k] & 0xFFFFu] + g2[k] * tile[loc[k] >> 16];
                    vis[k] >>= 8;
                }
            }
            __syncthreads();
        }
        long long t2c = timing ? clock64() : 0;
        // ---- phase 3: the layer is final -> global, visible to this workgroup's next layers ---
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = tid + k * T;
            if (slot < cnt) I[lo + slot] = tile[slot];
        }
        __syncthreads();
        if (timing) {
            const long long t3c = clock64();
            cyc1 += t1c - t0; cyc2 += t2c - t1c; cyc3 += t3c - t2c;
        }
    }
    if (timing && tid == 0) {
        ta.dbg[4 * task + 0] = cyc1; ta.dbg[4 * task + 1] = cyc2; ta.dbg[4 * task + 2] = cyc3;
        ta.dbg[4 * task + 3] = a;
    }
    if (tid == 0) I[n - 1] = 0.0;   // the never-visited site perm[n] keeps I = 0 (voronoi_utils.jl:266)
}

// ---------------------------------------------------------------------------------------------
// Two-launch form of the persistent tile path.  Measured on BASELINE config C2 (12 tasks on 12
// CUs): the coefficient phase of k_sweep_tiles is ALU-bound on its one CU (53 % of the critical
// task; staging S, α of two layers in LDS so that every gather is an LDS read changed nothing:
// 1.345 vs 1.338 ms).  But only ONE term of a site's coefficients depends on the sweep's results:
//     I_c = c0 + H1 I_u1 + H2 I_u2,   c0 = Σ_r (a_r S_ur + b_r S_c) w_r,   H_r = e_r w_r
// (irregular_ray_tracing.jl:73-76 re-associated; the reference adds e_r I_ur inside the bracket).
// So a first chip-wide launch (k_tile_coeffs, no dependencies at all: every site x task in
// parallel) computes c0, H1, H2 with all the exponentials, and the persistent workgroup of a task
// (k_sweep_tiles_pre) only streams three doubles + two schedule words per site, adds the couplings
// to the PREVIOUS layer from its LDS copy of that layer's final intensities, and runs the levels.
// Everything is laid out in the SORTED order of the level loop (visit patterns wave-uniform), so
// the workgroup's loads are perfectly coalesced and prefetched one layer ahead.
//   t_code_ss[a][i] (plan time): the two upwinds of sorted entry i, 14 bits each:
//     bits 0-11 slot in sorted terms, bits 12-13 kind (1 = own layer -> tile, 2 = previous layer,
//     0 = neither: later layer / never-visited site, the intensity reads 0)
// ---------------------------------------------------------------------------------------------
constexpr int kPreMaxLayer = 4096;       // 12-bit slots

// launch 1: c0, H1, H2 of every (task, sorted entry); planes [3][ntask][n]
__global__ void __launch_bounds__(256)
k_tile_coeffs(TileArgs ta, double *__restrict__ rec)
{
    const int64_t n = ta.n;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int task = blockIdx.y;
    const int a = ta.task_map[task] & 0xFF;
    const int l = ta.task_map[task] >> 8;
    const int d = ta.angle_dir[a];
    const size_t tab = (size_t)a * (size_t)n;
    const int p = ta.t_self[tab + i];
    const int u1 = ta.t_u1[tab + p], u2 = ta.t_u2[tab + p];
    const size_t ntask = gridDim.y;
    double *c0 = rec + (size_t)task * (size_t)n, *H1 = c0 + ntask * (size_t)n, *H2 = H1 + ntask * (size_t)n;
    if (i < ta.lay[d][1] || u1 < 0 || u2 < 0) {          // boundary layer (no visits) / no upwind
        c0[i] = 0.0; H1[i] = 0.0; H2[i] = 0.0;
        return;
    }
    const double *__restrict__ S = ta.S[d] + (size_t)l * (size_t)n;
    const double *__restrict__ Al =
        ta.alpha_mode == VRT_ALPHA_SITE ? ta.alpha[d]
        : ta.alpha_mode == VRT_ALPHA_SITE_LAM ? ta.alpha[d] + (size_t)l * (size_t)n
                                              : ta.alpha_angle + ((size_t)a * ta.nlam + l) * (size_t)n;
    const double S_c = S[p], a_c = Al[p];
    double ca, cb, ce;
    lin_weights(ta.t_r1[tab + p] * (a_c + Al[u1]) / 2.0, ca, cb, ce);       // trapezoidal, functions.jl:393
    const double w1 = ta.t_w1[tab + p];
    const double t1 = (ca * S[u1] + cb * S_c) * w1;
    H1[i] = ce * w1;
    lin_weights(ta.t_r2[tab + p] * (a_c + Al[u2]) / 2.0, ca, cb, ce);
    const double w2 = ta.t_w2[tab + p];
    const double t2 = (ca * S[u2] + cb * S_c) * w2;
    H2[i] = ce * w2;
    c0[i] = t1 + t2;
}

// launch 2: one persistent workgroup per task; LDS = tile of the current layer, final intensities
// of the previous layer, constant terms (3 x tile_stride doubles), all in sorted order
template <int K, int T>
__global__ void __launch_bounds__(T)
k_sweep_tiles_pre(TileArgs ta, const double *__restrict__ rec, const uint32_t *__restrict__ code_ss,
                  const int32_t *__restrict__ rank_s)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int stride = ta.tile_stride;
    double *cst = lds + 2 * (size_t)stride;
    const int tid = threadIdx.x;
    const int task = blockIdx.x;
    const int a = ta.task_map[task] & 0xFF;
    const int l = ta.task_map[task] >> 8;
    const int d = ta.angle_dir[a];
    const int64_t n = ta.n;
    const size_t tab = (size_t)a * (size_t)n;
    const size_t ntask = gridDim.x;
    const double *__restrict__ c0 = rec + (size_t)task * (size_t)n;
    const double *__restrict__ H1 = c0 + ntask * (size_t)n;
    const double *__restrict__ H2 = H1 + ntask * (size_t)n;
    const uint32_t *__restrict__ code = code_ss + tab;
    const uint32_t *__restrict__ tvis = ta.t_vis_s + tab;
    const int32_t *__restrict__ tself = ta.t_self + tab;
    const int32_t *__restrict__ trank = rank_s + tab;
    double *I = ta.I + ((size_t)a * ta.nlam + l) * (size_t)n;
    const int32_t *__restrict__ lay = ta.lay[d];
    const int32_t *__restrict__ nlev = ta.nlev + (size_t)a * (size_t)(ta.max_layers + 1);
    const int L = ta.nlayers[d];

    struct Entry { double c0, h1, h2; uint32_t code, vis; };
    auto load_entries = [&](int layer, Entry (&e)[K]) {
        const int lo = lay[layer - 1], cnt = lay[layer] - lo;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = tid + k * T;
            const int i = lo + min(slot, max(cnt - 1, 0));
            e[k].c0 = c0[i]; e[k].h1 = H1[i]; e[k].h2 = H2[i];
            e[k].code = code[i];
            e[k].vis = slot < cnt ? tvis[i] : 0u;
        }
    };
    int par = 0;      // lds[par * stride ..]: tile of the current layer, the other one: the previous layer
    {
        // layer 1 (boundary: I = I_0, written by k_boundary_sweep_order) is the first "previous" layer
        const int lo1 = lay[0], cnt1 = lay[1] - lo1;
        double *Ip = lds + (size_t)(par ^ 1) * stride;
        for (int s = tid; s < cnt1; s += T) Ip[s] = I[tself[lo1 + s]];
    }
    Entry cur[K];
    if (L >= 2) load_entries(2, cur);
    __syncthreads();
    for (int layer = 2; layer <= L; layer++) {          // irregular_ray_tracing.jl:37
        const int lo = lay[layer - 1], hi = lay[layer];  // hi of the last layer = n-1: perm[n] is never visited
        const int cnt = hi - lo;
        double *Ic = lds + (size_t)par * stride;
        const double *Ip = lds + (size_t)(par ^ 1) * stride;
        Entry nxt[K];
        if (layer < L) load_entries(layer + 1, nxt);     // lands during the level loop
        double g1[K], g2[K];
        uint32_t loc[K], vis[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t c1 = cur[k].code & 0x3FFFu, c2 = cur[k].code >> 14;
            const uint32_t k1 = c1 >> 12, k2 = c2 >> 12, s1 = c1 & 0xFFFu, s2 = c2 & 0xFFFu;
            double c = cur[k].c0;
            if (k1 == 2u) c += cur[k].h1 * Ip[s1];                   // previous layer: final
            if (k2 == 2u) c += cur[k].h2 * Ip[s2];
            g1[k] = k1 == 1u ? cur[k].h1 : 0.0;                      // own layer: coupling on the tile
            g2[k] = k2 == 1u ? cur[k].h2 : 0.0;
            loc[k] = (k1 == 1u ? s1 : (uint32_t)cnt) | ((k2 == 1u ? s2 : (uint32_t)cnt) << 16);   // else the zero slot
            vis[k] = cur[k].vis;
            const int slot = tid + k * T;
            if (slot < cnt) {
                cst[slot] = c;
                Ic[slot] = 0.0;                                      // I = zero(S), :23
            }
        }
        if (tid == 0) Ic[cnt] = 0.0;                                 // the zero slot
        __syncthreads();
        const int nl = nlev[layer];
        for (int t = 1; t <= nl; t++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if ((vis[k] & 0xFFu) == (uint32_t)t) {               // a site's visits come at increasing levels
                    Ic[tid + k * T] = cst[tid + k * T] + g1[k] * Ic[loc[k] & 0xFFFFu] + g2[k] * Ic[loc[k] >> 16];
                    vis[k] >>= 8;
                }
            }
            __syncthreads();
        }
        // the layer is final: to global in storage order (J reduction / I_out); it stays in LDS, in
        // sorted order, as the next layer's "previous"
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = tid + k * T;
            if (slot < cnt) I[lo + slot] = Ic[trank[lo + slot] - lo];
        }
        par ^= 1;
#pragma unroll
        for (int k = 0; k < K; k++) cur[k] = nxt[k];
        __syncthreads();       // the next layer zeroes what was "previous" until now
    }
    if (tid == 0) I[n - 1] = 0.0;   // the never-visited site perm[n] keeps I = 0 (voronoi_utils.jl:266)
}

// ---------------------------------------------------------------------------------------------
// "Layer-step" variant of the same algorithm: the two phases of a layer become two chip-wide
// launches.  k_step_coeffs has no dependencies inside a layer, so it runs at full occupancy
// (deep memory-level parallelism for the gathers); the coefficients it leaves in a reused
// buffer are consumed immediately by k_step_levels, one workgroup per (angle, wavelength PAIR),
// which only does the LDS Gauss-Seidel levels.  2 launches per BFS layer instead of one per
// dependency level.
//
// Every array of this path holds wavelength PAIRS side by side ([λ/2][pos][2], sw_index with
// lb = 2): each centre read, upwind gather, coefficient store/load and intensity store is one
// 16-byte access per lane serving two wavelengths -- half the vector-memory instructions and
// half the cache lines touched per gathered value of the 8-byte planes (narrow global accesses
// are issue-bound on gfx950: cdna_hip_programming.md, "under-vectorized global reads").  The
// in-layer dependency structure (levels, tile slots) is the same for every wavelength of an
// angle, so a level visit of the pair costs the same LDS instructions (b128) as one wavelength.
// ---------------------------------------------------------------------------------------------

// block = 256 consecutive slots (a Morton-coherent patch: the upwind gathers of neighbouring
// slots share lines through L1) of one angle; each thread keeps its slot's upwind-table entry
// in registers and loops over a group of kStepPairs wavelength pairs, so the 44-byte entry is
// read once per group and the loads of the group's pairs are independent.
// grid: x = slot chunk, y = angle * ceil(npair / kStepPairs) + pair group
constexpr int kStepPairs = 4;   // fewest pairs per thread (the launch picks 4 to 6; VRT_STEP_PAIRS overrides)

// T: storage type of S, α, I.  SPLIT = false: the coefficients go to the pair level kernel as double2
// (c) + a compact double2 list (g); SPLIT = true: to the single-wavelength level kernel as one
// plane of T per wavelength (c[l][slot], compact g[l][..]) -- fp32 on the fp32 value path, which
// halves the hand-off bytes.
template <typename T, bool SPLIT>
__global__ void __launch_bounds__(256)
k_step_coeffs(StepArgs sa)
{
    typedef typename Pair<T>::type T2;
    const TileArgs &ta = sa.ta;
    const int ppt = sa.pairs_per_thread;
    const int ngrp = (sa.npair + ppt - 1) / ppt;
    // 1-D grid of chunks x (angle, pair group).  Workgroups are dealt round-robin to the 8 XCDs
    // (block b and b + 8 share one: MI355X_MICROARCH.md, speed only), so with xcd_map each XCD
    // takes a contiguous range of a layer's chunks for every (angle, pair group): neighbouring
    // Morton patches share their boundary gather lines, and the pair groups of an angle their
    // table entries, through that XCD's L2.
    int chunk, grp;
    if (sa.xcd_map) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int c0 = (sa.chunks * x) >> 3, c1 = (sa.chunks * (x + 1)) >> 3, cx = c1 - c0;
        if (cx == 0) return;
        grp = j / cx;
        chunk = c0 + j % cx;
        if (grp >= sa.n_list * ngrp) return;
    } else {
        chunk = blockIdx.x % sa.chunks;
        grp = blockIdx.x / sa.chunks;
    }
    // angle fastest: the angles of a direction read the same S lines for a (chunk, pair group)
    const int a = sa.xcd_map == 2 ? sa.angle_list[grp % sa.n_list] : sa.angle_list[grp / ngrp];
    const int q0 = (sa.xcd_map == 2 ? grp / sa.n_list : grp % ngrp) * ppt;
    const int d = ta.angle_dir[a];
    if (sa.layer > ta.nlayers[d]) return;
    const int lo = ta.lay[d][sa.layer - 1], hi = ta.lay[d][sa.layer];
    const int slot = chunk * 256 + threadIdx.x;
    if (slot >= hi - lo) return;
    const int64_t n = ta.n;
    const size_t tab = (size_t)a * (size_t)n;
    const int p = lo + slot;
    const int u1 = ta.t_u1[tab + p], u2 = ta.t_u2[tab + p];
    const double w1 = ta.t_w1[tab + p], w2 = ta.t_w2[tab + p], r1 = ta.t_r1[tab + p], r2 = ta.t_r2[tab + p];
    const uint32_t gp = ta.t_gpos[tab + p];
    const bool early1 = u1 < lo, in1 = (gp >> 30) & 1u;      // in = upwind inside [lo, hi)
    const bool early2 = u2 < lo, in2 = gp >> 31;
    int i1 = min(u1, lo - 1), i2 = min(u2, lo - 1);
    const int dbg = kDiag ? sa.debug_flags : 0;
    int v1 = u1, v2 = u2;
    if (dbg & 1) { v1 = p; v2 = p; }              // S/alpha gathers -> coalesced centre re-reads
    if (dbg & 2) { i1 = lo - 1; i2 = lo - 1; }    // I gathers -> one broadcast address
    const int qend = min(q0 + ppt, sa.npair);
    struct PairIn { double2 a_c, a_1, a_2, S_c, S_1, S_2, I_1, I_2; };
    auto load_pair = [&](int q) {
        PairIn in;
        const T2 *__restrict__ S = reinterpret_cast<const T2 *>(ta.S[d]) + (size_t)q * (size_t)n;
        const T2 *__restrict__ I = reinterpret_cast<const T2 *>(ta.I) + ((size_t)a * sa.npair + q) * (size_t)n;
        if (ta.alpha_mode == VRT_ALPHA_SITE) {                      // one opacity per site for every λ
            const T *__restrict__ Al = reinterpret_cast<const T *>(ta.alpha[d]);
            const double c0 = Al[p], c1 = Al[v1], c2 = Al[v2];
            in.a_c = make_double2(c0, c0); in.a_1 = make_double2(c1, c1); in.a_2 = make_double2(c2, c2);
        } else {
            const T2 *__restrict__ Al =
                ta.alpha_mode == VRT_ALPHA_SITE_LAM
                    ? reinterpret_cast<const T2 *>(ta.alpha[d]) + (size_t)q * (size_t)n
                    : reinterpret_cast<const T2 *>(ta.alpha_angle) + ((size_t)a * sa.npair + q) * (size_t)n;
            in.a_c = ld2(Al, p); in.a_1 = ld2(Al, v1); in.a_2 = ld2(Al, v2);
        }
        in.S_c = ld2(S, p); in.S_1 = ld2(S, v1); in.S_2 = ld2(S, v2);
        in.I_1 = ld2(I, i1); in.I_2 = ld2(I, i2);
        return in;
    };
    // software pipeline over the thread's pairs: the 8 loads of pair q + 1 are in flight while pair q
    // is computed (VRT_DEBUG_FLAGS & 128 switches the prefetch off)
    const bool prefetch = !(dbg & 128);
    PairIn cur = load_pair(q0);
    for (int q = q0; q < qend; q++) {
        PairIn nxt = cur;
        if (prefetch && q + 1 < qend) nxt = load_pair(q + 1);
        double2 c, g1, g2;
        double t1, t2;
        const bool cheap = dbg & 32;
        upwind_term(r1, w1, cur.a_c.x, cur.a_1.x, cur.S_c.x, cur.S_1.x, cur.I_1.x, early1, in1, t1, g1.x, cheap);
        upwind_term(r2, w2, cur.a_c.x, cur.a_2.x, cur.S_c.x, cur.S_2.x, cur.I_2.x, early2, in2, t2, g2.x, cheap);
        c.x = t1 + t2;
        upwind_term(r1, w1, cur.a_c.y, cur.a_1.y, cur.S_c.y, cur.S_1.y, cur.I_1.y, early1, in1, t1, g1.y, cheap);
        upwind_term(r2, w2, cur.a_c.y, cur.a_2.y, cur.S_c.y, cur.S_2.y, cur.I_2.y, early2, in2, t2, g2.y, cheap);
        c.y = t1 + t2;
        if (SPLIT) {                                                // one plane of T per wavelength
            T *cc = reinterpret_cast<T *>(sa.cg_c), *gg = reinterpret_cast<T *>(sa.cg_g);
            const size_t o0 = ((size_t)a * (2 * sa.npair) + 2 * q) * (size_t)sa.cg_stride, o1 = o0 + (size_t)sa.cg_stride;
            cc[o0 + slot] = (T)c.x;
            cc[o1 + slot] = (T)c.y;
            T *g0 = gg + 2 * o0 + (gp & 0xFFFFu), *gy = gg + 2 * o1 + (gp & 0xFFFFu);
            if (in1) { g0[0] = (T)g1.x; gy[0] = (T)g1.y; }
            if (in2) { g0[in1 ? 1 : 0] = (T)g2.x; gy[in1 ? 1 : 0] = (T)g2.y; }
        } else if (!((dbg & 4) && c.x != 1.2345e300)) {             // (dbg & 4: no coefficient stores)
            const size_t o = ((size_t)a * sa.npair + q) * (size_t)sa.cg_stride + (size_t)slot;
            sa.cg_c[o] = c;
            double2 *gl = sa.cg_g + 2 * (o - (size_t)slot) + (gp & 0xFFFFu);
            if (in1) gl[0] = g1;
            if (in2) gl[in1 ? 1 : 0] = g2;
        }
        if (!prefetch && q + 1 < qend) nxt = load_pair(q + 1);
        cur = nxt;
    }
}

// Task (angle-major index into angle_list x wavelengths) of a level workgroup.  Workgroups are dealt
// round-robin to the 8 XCDs (block b and b + 8 share one); level_map (build_level_map) gives each
// XCD a contiguous run of the tasks, cut at equal estimated cost: the wavelengths of an angle
// read that angle's tables (16-20 B per site and workgroup) through ONE L2 instead of all eight
// (C5: 7 MB of tables per layer do not fit a 4 MB L2), and the two wavelengths of a pair store
// their halves of the same lines of I through the same L2.  -1: padding block.
__device__ __forceinline__ int level_task(const StepArgs &sa, int ntask)
{
    if (sa.level_map) return sa.level_map[blockIdx.x];
    return (int)blockIdx.x < ntask ? (int)blockIdx.x : -1;
}

template <int K>
__global__ void __launch_bounds__(1024)
k_step_levels(StepArgs sa)
{
    extern __shared__ __attribute__((aligned(16))) double2 tile2[];
    const TileArgs &ta = sa.ta;
    const int T = 1024, tid = threadIdx.x;
    const int lt = level_task(sa, sa.n_list * sa.npair);
    if (lt < 0) return;
    const int a = sa.angle_list[lt / sa.npair], q = lt % sa.npair;
    const int task = a * sa.npair + q;
    const int d = ta.angle_dir[a];
    if (sa.layer > ta.nlayers[d]) return;
    const int lo = ta.lay[d][sa.layer - 1], hi = ta.lay[d][sa.layer];
    const int cnt = hi - lo;
    const int64_t n = ta.n;
    const size_t tab = (size_t)a * (size_t)n;
    const uint32_t *__restrict__ tvis = ta.t_vis_s + tab;
    const uint32_t *__restrict__ tloc = ta.t_loc_s + tab;
    const int32_t *__restrict__ tself = ta.t_self + tab;
    double2 *I = reinterpret_cast<double2 *>(ta.I) + (size_t)task * (size_t)n;
    const size_t o = (size_t)task * (size_t)sa.cg_stride;
    double2 c[K], g1[K], g2[K];
    uint32_t loc[K], vis[K], self[K];   // self: storage slot of the sorted entry this thread owns
    // coefficients arrive in storage order (coalesced 16-byte loads) ...
    const int dbgl = kDiag ? sa.debug_flags : 0;
    const bool sorted = !(dbgl & 64);
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = tid + k * T;
        const bool ok = i < cnt;
        const int ii = ok ? i : cnt - 1;
        if (dbgl & 8) {                  // no coefficient loads
            c[k] = make_double2(1.0 + ii, 2.0 + ii); g1[k] = make_double2(0.25, 0.25); g2[k] = make_double2(0.125, 0.125);
        } else {
            c[k] = sa.cg_c[o + ii];
            const uint32_t gp = ta.t_gpos[tab + lo + ii];
            const double2 *gl = sa.cg_g + 2 * o + (gp & 0xFFFFu);
            const bool in1 = (gp >> 30) & 1u, in2 = gp >> 31;
            g1[k] = in1 ? gl[0] : make_double2(0.0, 0.0);
            g2[k] = in2 ? gl[in1 ? 1 : 0] : make_double2(0.0, 0.0);
        }
        if (sorted) {
            self[k] = (uint32_t)(tself[lo + ii] - lo);
            loc[k] = tloc[lo + ii];
            vis[k] = ok ? tvis[lo + ii] : 0u;
        } else {                                   // diagnostics (VRT_DEBUG_FLAGS & 64): storage-order assignment
            self[k] = (uint32_t)ii;
            loc[k] = ta.t_loc[tab + lo + ii];
            vis[k] = ok ? ta.t_vis[tab + lo + ii] : 0u;
        }
    }
    // ... and are dealt to the threads in visit-pattern order through the (still unused) tile:
    // written at their storage slot (consecutive, conflict-free), read back at the slot of the
    // sorted entry tid + k T this thread owns, whose visit levels are nearly wave-uniform
    if (sorted) {
#pragma unroll
        for (int arr = 0; arr < 3; arr++) {
            double2 *v = arr == 0 ? c : arr == 1 ? g1 : g2;
#pragma unroll
            for (int k = 0; k < K; k++)
                if (tid + k * T < cnt) tile2[tid + k * T] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = tile2[self[k]];
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        if (tid + k * T < cnt) tile2[tid + k * T] = make_double2(0.0, 0.0);   // I = zero(S), irregular_ray_tracing.jl:23
    if (tid == 0) tile2[cnt] = make_double2(0.0, 0.0);                         // the zero slot
    __syncthreads();
    const int nl = (kDiag && sa.debug_skip_levels) ? 0 : ta.nlev[(size_t)a * (size_t)(ta.max_layers + 1) + sa.layer];
    for (int t = 1; t <= nl; t++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((vis[k] & 0xFFu) == (uint32_t)t) {       // a site's visits come at increasing levels
                const uint32_t lx = loc[k] & 0xFFFFu, ly = loc[k] >> 16;   // kNoSlot -> the zero slot
                const double2 x = tile2[lx == kNoSlot ? (uint32_t)cnt : lx], y = tile2[ly == kNoSlot ? (uint32_t)cnt : ly];
                double2 r;
                r.x = c[k].x + g1[k].x * x.x + g2[k].x * y.x;
                r.y = c[k].y + g1[k].y * x.y + g2[k].y * y.y;
                tile2[self[k]] = r;
                vis[k] >>= 8;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int slot = tid + k * T;
        if (slot < cnt && (!(dbgl & 16) || tile2[slot].x == 1.2345e300)) I[lo + slot] = tile2[slot];
    }
    if (tid == 0 && sa.layer == ta.nlayers[d]) I[n - 1] = make_double2(0.0, 0.0);   // never-visited site perm[n]
}

// Single-wavelength level kernel for layers the pair kernel cannot hold (its tile is 16 B per site
// and its register-resident coefficients 15 VGPRs per site, i.e. 8192 sites): one workgroup per
// (angle, wavelength), tile of T (8 B or 4 B per site) kept in SORTED order so a thread's write
// slot is its own index (no per-site register for it), coefficients c, g1, g2 held as T (6 or 3
// VGPRs per site) + packed upwind slots + packed visit levels: up to 12 sites per thread in fp64
// (12 288-site layers), 18 in fp32 (18 432).  The visit arithmetic is done in fp64.
template <typename T, int K>
__global__ void __launch_bounds__(1024)
k_step_levels1(StepArgs sa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    T *tile1 = reinterpret_cast<T *>(tile_raw);
    const TileArgs &ta = sa.ta;
    const int TT = 1024, tid = threadIdx.x;
    const int lt = level_task(sa, sa.n_list * ta.nlam);
    if (lt < 0) return;
    const int a = sa.angle_list[lt / ta.nlam], l = lt % ta.nlam;
    const int d = ta.angle_dir[a];
    if (sa.layer > ta.nlayers[d]) return;
    const int lo = ta.lay[d][sa.layer - 1], hi = ta.lay[d][sa.layer];
    const int cnt = hi - lo;
    const int64_t n = ta.n;
    const size_t tab = (size_t)a * (size_t)n;
    const uint32_t *__restrict__ tvis = ta.t_vis_s + tab;
    const uint32_t *__restrict__ tloc = sa.t_loc_ss + tab;
    const int32_t *__restrict__ tself = ta.t_self + tab;
    const int32_t *__restrict__ trank = sa.t_rank_s + tab;
    // element (l, pos) of the pair planes: ((l / 2) n + pos) 2 + l % 2
    T *I = reinterpret_cast<T *>(ta.I) + (((size_t)a * sa.npair + (size_t)(l >> 1)) * (size_t)n << 1) + (size_t)(l & 1);
    const size_t o = ((size_t)a * (2 * sa.npair) + l) * (size_t)sa.cg_stride;
    const T *__restrict__ cc = reinterpret_cast<const T *>(sa.cg_c) + o;
    const T *__restrict__ gg = reinterpret_cast<const T *>(sa.cg_g) + 2 * o;
    T c[K], g1[K], g2[K];
    uint32_t loc[K], vis[K];
    const int dbgl = kDiag ? sa.debug_flags : 0;     // timing diagnostics (-DVRT_DIAG build only)
    {
        uint32_t self[K];            // live during the permutation only
        // coefficients arrive in storage order (coalesced) ...
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = tid + k * TT;
            const bool ok = i < cnt;
            const int ii = ok ? i : cnt - 1;
            if (dbgl & 8) {              // no coefficient loads
                c[k] = (T)(1.0 + ii); g1[k] = (T)0.25; g2[k] = (T)0.125;
            } else {
            c[k] = cc[ii];
            const uint32_t gp = ta.t_gpos[tab + lo + ii];
            const T *gl = gg + (gp & 0xFFFFu);
            const bool in1 = (gp >> 30) & 1u, in2 = gp >> 31;
            g1[k] = in1 ? gl[0] : (T)0;
            g2[k] = in2 ? gl[in1 ? 1 : 0] : (T)0;
            }
            self[k] = (uint32_t)(tself[lo + ii] - lo);
            loc[k] = tloc[lo + ii];
            vis[k] = ok ? tvis[lo + ii] : 0u;
        }
        // ... and are dealt to the threads in visit-pattern order through the still unused tile
        if (!(dbgl & 256))               // (256: no permutation)
#pragma unroll
        for (int arr = 0; arr < 3; arr++) {
            T *v = arr == 0 ? c : arr == 1 ? g1 : g2;
#pragma unroll
            for (int k = 0; k < K; k++)
                if (tid + k * TT < cnt) tile1[tid + k * TT] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = tile1[self[k]];
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        if (tid + k * TT < cnt) tile1[tid + k * TT] = (T)0;      // I = zero(S), irregular_ray_tracing.jl:23
    if (tid == 0) tile1[cnt] = (T)0;                             // the zero slot
    __syncthreads();
    const int nl = (kDiag && sa.debug_skip_levels) ? 0 : ta.nlev[(size_t)a * (size_t)(ta.max_layers + 1) + sa.layer];
    for (int t = 1; t <= nl; t++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((vis[k] & 0xFFu) == (uint32_t)t) {       // a site's visits come at increasing levels
                if (dbgl & 512) { vis[k] >>= 8; continue; }   // (512: levels polled, no visits)
                uint32_t lk = loc[k];
                asm volatile("" : "+v"(lk));      // keep the two slots packed in ONE register (no hoisted addresses)
                const uint32_t lx = lk & 0xFFFFu, ly = lk >> 16;           // kNoSlot -> the zero slot
                const double x = (double)tile1[lx == kNoSlot ? (uint32_t)cnt : lx];
                const double y = (double)tile1[ly == kNoSlot ? (uint32_t)cnt : ly];
                T ck = c[k], g1k = g1[k], g2k = g2[k];
                if (sizeof(T) == 4)      // keep the state in fp32 registers: without this the compiler hoists
                    asm volatile("" : "+v"(ck), "+v"(g1k), "+v"(g2k));   // the conversions and holds doubles
                tile1[tid + k * TT] = (T)((double)ck + (double)g1k * x + (double)g2k * y);
                vis[k] >>= 8;
            }
        }
        __syncthreads();
    }
    // the tile is in sorted order: storage slot i holds tile1[rank_s[i]] (LDS gather, coalesced store)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int slot = tid + k * TT;
        if (slot < cnt && (!(dbgl & 16) || tile1[slot] == (T)1.2345e30)) {
            if (dbgl & 1024) I[(size_t)(lo + slot) + (size_t)(l & 1) * (size_t)(hi - lo)] = tile1[trank[lo + slot] - lo];   // (1024: contiguous stores)
            else
            I[(size_t)(lo + slot) << 1] = tile1[trank[lo + slot] - lo];
        }
    }
    if (tid == 0 && sa.layer == ta.nlayers[d]) I[(size_t)(n - 1) << 1] = (T)0;   // never-visited site perm[n]
}

// J_d[l][p] = Σ_{angles of direction d} w_a I_a[l][p], reference's angle order within the direction
template <typename T>
__global__ void __launch_bounds__(256)
k_reduce_dir(int64_t total, int64_t stride_angle, DirWeights dw, const T *__restrict__ I,
             T *__restrict__ Jd)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    double acc = 0.0;
    for (int j = 0; j < dw.count; j++) acc += dw.w[j] * (double)I[(size_t)dw.idx[j] * stride_angle + t];
    Jd[t] = (T)acc;
}

// J[site][l] = J_up[l][rank_up[site]] + J_down[l][rank_down[site]], walking sites in up order so
// the J_up reads are coalesced and the J_down reads are piecewise contiguous on stratified grids.
template <typename T>
__global__ void __launch_bounds__(256)
k_combine_J(int64_t n, int nlam, int64_t ldJ, int lb, const int32_t *__restrict__ order_up,
            const int32_t *__restrict__ rank_down, const T *__restrict__ Ju,
            const T *__restrict__ Jdn, T *__restrict__ J)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    const int64_t p = p0 + tx;
    int32_t site = 0, pd = 0;
    if (p < n) {
        site = order_up[p];
        pd = rank_down[site];
    }
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        if (l < nlam && p < n) {
            double v = 0.0;
            if (Ju) v = (double)Ju[sw_index(l, p, n, lb)];
            if (Jdn) v = v + (double)Jdn[sw_index(l, pd, n, lb)];
            tile[tx][c] = (T)v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = p0 + r;
        if (q < n && l0 + tx < nlam) J[(size_t)order_up[q] * ldJ + l0 + tx] = tile[r][tx];
    }
}

// out[order[p]][l] = in[l][p]  (sweep order, wavelength-major -> caller's site-major rows)
template <typename T>
__global__ void __launch_bounds__(256)
k_from_sweep_order(int64_t n, int nlam, int64_t ld, int lb, const int32_t *__restrict__ order,
                   const T *__restrict__ in, T *__restrict__ out)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        if (l < nlam && p0 + tx < n) tile[tx][c] = in ? in[sw_index(l, p0 + tx, n, lb)] : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = p0 + r;
        if (q < n && l0 + tx < nlam) out[(size_t)order[q] * ld + l0 + tx] = tile[r][tx];
    }
}

// ---- host side of one execute on the tile path ---------------------------------------------------
static int ensure_dev(double *&buf, size_t &cap, size_t count)
{
    if (buf && count <= cap) return VRT_OK;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&buf, std::max<size_t>(count, 1) * sizeof(double));
    if (e != hipSuccess) {
        buf = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE,
                    std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    cap = count;
    return VRT_OK;
}

// Block -> (angle, wavelength) map.  Workgroups are dealt round-robin to the 8 XCDs (block b runs
// on the XCD that also runs b + 8, b + 16, ...: MI355X_MICROARCH.md, speed only), and every
// XCD has a private 4 MB L2.  The 44-byte-per-site upwind table of an angle is shared by all
// wavelength tasks of that angle, so each angle's wavelengths are split into two groups and the
// groups are dealt to the XCDs (longest-processing-time first): the tasks that share a table
// run on one XCD, in lockstep, and read it from that XCD's L2 instead of HBM.  A different
// placement would only be slower, never wrong.
static int build_task_map(vrt_plan *p, int nlam, hipStream_t st)
{
    const int A = p->A;
    if (p->task_map_nlam == nlam && p->d_task_map) return VRT_OK;
    const int ntask = A * nlam;
    struct Group { int a, l0, l1; double cost; };
    std::vector<Group> groups;
    const int halves = nlam >= 2 ? 2 : 1;
    for (int a = 0; a < A; a++)
        for (int h = 0; h < halves; h++) {
            const int l0 = h * nlam / halves, l1 = (h + 1) * nlam / halves;
            groups.push_back({a, l0, l1, (double)p->angle_visits[(size_t)a] * (double)(l1 - l0)});
        }
    std::stable_sort(groups.begin(), groups.end(), [](const Group &x, const Group &y) { return x.cost > y.cost; });
    std::vector<std::vector<int>> per_xcd(8);      // task lists, heaviest groups first
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Group &gr : groups) {
        int x = 0;
        for (int q = 1; q < 8; q++)
            if (load[q] < load[x]) x = q;
        load[x] += gr.cost;
        for (int l = gr.l0; l < gr.l1; l++) per_xcd[(size_t)x].push_back(gr.a | (l << 8));
    }
    // interleave: block b takes the next task of XCD b % 8; XCDs that run dry borrow from the fullest
    p->h_task_map.assign((size_t)ntask, 0);
    size_t cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < ntask; b++) {
        int x = b % 8;
        if (cur[x] >= per_xcd[(size_t)x].size()) {
            size_t best = 0;
            for (int q = 0; q < 8; q++) {
                const size_t left = per_xcd[(size_t)q].size() - cur[q];
                if (left > best) { best = left; x = q; }
            }
        }
        p->h_task_map[(size_t)b] = per_xcd[(size_t)x][cur[x]++];
    }
    if (!p->d_task_map || p->task_map_cap < (size_t)ntask) {
        if (p->d_task_map) (void)hipFree(p->d_task_map);
        p->d_task_map = nullptr;
        VRT_HIP_TRY(hipMalloc((void **)&p->d_task_map, sizeof(int32_t) * (size_t)std::max(ntask, 1)));
        p->task_map_cap = (size_t)ntask;
    }
    VRT_HIP_TRY(hipMemcpyAsync(p->d_task_map, p->h_task_map.data(), sizeof(int32_t) * (size_t)ntask,
                               hipMemcpyHostToDevice, st));
    p->task_map_nlam = nlam;
    return VRT_OK;
}

// caller's per-angle alpha (n_angles, n, ld) -> the native layout of VRT_ALPHA_ANGLE_NATIVE
int alpha_to_native(vrt_plan *p, int64_t nlam, int64_t ld, const double *dalpha, double *out, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int64_t nl_pad = (nlam + 1) / 2 * 2;
    const size_t plane = (size_t)nl_pad * (size_t)n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    for (int a = 0; a < p->A; a++) {
        const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
        hipLaunchKernelGGL(k_to_sweep_order<double>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2, dir.d_store,
                           dalpha + (size_t)p->user_of_active[(size_t)a] * (size_t)n * (size_t)ld,
                           out + (size_t)a * plane);
    }
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// internal streams + angle groups of the layer-step path
static int ensure_step_streams(vrt_plan *p, int G)
{
    if (p->step_groups == G && p->d_step_angles) return VRT_OK;
    const int A = p->A;
    std::vector<int32_t> order((size_t)A);
    for (int a = 0; a < A; a++) order[(size_t)a] = a;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return p->angle_visits[(size_t)x] > p->angle_visits[(size_t)y];
    });
    std::vector<int32_t> list;
    p->step_group_off.assign((size_t)G + 1, 0);
    // two streams and as many up as down angles: one direction per stream, whose angles share the
    // S planes and the storage order (C4 11.85 -> 11.70 ms); otherwise dealt heaviest first
    const bool by_dir = G == 2 && p->n_up > 0 && std::abs(p->n_up - p->n_down) <= 1 &&
                        p->tune.step_group_dir != 0;
    for (int gi = 0; gi < G; gi++) {
        p->step_group_off[(size_t)gi] = (int)list.size();
        if (by_dir) {                              // one direction per stream: its angles share the S planes
            for (int j = 0; j < A; j++)
                if ((p->dir_of_active[(size_t)order[(size_t)j]] > 0) == (gi == 0)) list.push_back(order[(size_t)j]);
        } else
        for (int j = gi; j < A; j += G) list.push_back(order[(size_t)j]);
    }
    p->step_group_off[(size_t)G] = (int)list.size();
    if (!p->d_step_angles) VRT_HIP_TRY(hipMalloc((void **)&p->d_step_angles, sizeof(int32_t) * (size_t)std::max(A, 1)));
    VRT_HIP_TRY(hipMemcpy(p->d_step_angles, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice));
    p->h_step_angles = list;
    if (p->d_patch_work) { (void)hipFree(p->d_patch_work); p->d_patch_work = nullptr; }   // work lists follow the groups
    if (!p->step_fork) VRT_HIP_TRY(hipEventCreateWithFlags(&p->step_fork, hipEventDisableTiming));
    for (int gi = 0; gi < 4; gi++) {
        if (gi >= 1 && gi < G && !p->step_stream[gi]) {      // group 0 advances on the caller's stream
            VRT_HIP_TRY(hipStreamCreateWithFlags(&p->step_stream[gi], hipStreamNonBlocking));
            VRT_HIP_TRY(hipEventCreateWithFlags(&p->step_join[gi], hipEventDisableTiming));
        }
    }
    p->step_groups = G;
    return VRT_OK;
}

// block -> task maps of the level kernels, one per stream group (see level_task): the group's tasks
// (angle-major, `units` wavelengths or wavelength pairs per angle) are cut into 8 contiguous runs of
// equal estimated cost -- a fixed part (loads, permutation, stores) plus the angle's mean level
// count -- and XCD x (blocks x, x + 8, ...) walks run x.
static int build_level_map(vrt_plan *p, int G, int units)
{
    if (p->d_level_map && p->level_map_groups == G && p->level_map_units == units) return VRT_OK;
    if (p->d_level_map) { (void)hipFree(p->d_level_map); p->d_level_map = nullptr; }
    double mean_all = 0.0;
    for (double v : p->angle_mean_levels) mean_all += v;
    mean_all = p->angle_mean_levels.empty() ? 1.0 : std::max(1.0, mean_all / (double)p->angle_mean_levels.size());
    std::vector<int32_t> h_angles((size_t)p->A);
    VRT_HIP_TRY(hipMemcpy(h_angles.data(), p->d_step_angles, sizeof(int32_t) * (size_t)p->A, hipMemcpyDeviceToHost));
    std::vector<int32_t> map;
    p->level_map_off.assign((size_t)G + 1, 0);
    for (int gi = 0; gi < G; gi++) {
        p->level_map_off[(size_t)gi] = (int)map.size();
        const int j0 = p->step_group_off[(size_t)gi], n_list = p->step_group_off[(size_t)gi + 1] - j0;
        const int ntask = n_list * units;
        if (ntask == 0) continue;
        std::vector<double> w((size_t)n_list);
        double W = 0.0;
        for (int j = 0; j < n_list; j++) {
            const int a = h_angles[(size_t)(j0 + j)];
            const double lv = (size_t)a < p->angle_mean_levels.size() ? p->angle_mean_levels[(size_t)a] : mean_all;
            w[(size_t)j] = 1.7 * mean_all + lv;          // measured on C5: fixed part : level loop = 231 : 133
            W += w[(size_t)j] * units;
        }
        std::vector<std::vector<int32_t>> runs(8);
        double cum = 0.0;
        for (int t = 0; t < ntask; t++) {
            const double wt = w[(size_t)(t / units)];
            const int x = std::min(7, (int)((cum + 0.5 * wt) * 8.0 / W));
            runs[(size_t)x].push_back(t);
            cum += wt;
        }
        size_t per = 0;
        for (const auto &r : runs) per = std::max(per, r.size());
        for (size_t j = 0; j < per; j++)
            for (int x = 0; x < 8; x++) map.push_back(j < runs[(size_t)x].size() ? runs[(size_t)x][j] : -1);
    }
    p->level_map_off[(size_t)G] = (int)map.size();
    VRT_HIP_TRY(hipMalloc((void **)&p->d_level_map, sizeof(int32_t) * std::max<size_t>(map.size(), 1)));
    VRT_HIP_TRY(hipMemcpy(p->d_level_map, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice));
    p->level_map_groups = G;
    p->level_map_units = units;
    return VRT_OK;
}

// sites per thread the single-wavelength level kernel is instantiated for (even counts)
constexpr int kSingleMaxK64 = 12, kSingleMaxK32 = 18;

template <typename T, int K>
static void launch_levels1(dim3 grid, size_t lds, hipStream_t sg, const StepArgs &sa)
{
    hipLaunchKernelGGL((k_step_levels1<T, K>), grid, dim3(1024), lds, sg, sa);
}

template <typename T>
static void launch_levels1_K(int K, dim3 grid, size_t lds, hipStream_t sg, const StepArgs &sa)
{
    switch ((K + 1) / 2 * 2) {
    case 2: launch_levels1<T, 2>(grid, lds, sg, sa); break;
    case 4: launch_levels1<T, 4>(grid, lds, sg, sa); break;
    case 6: launch_levels1<T, 6>(grid, lds, sg, sa); break;
    case 8: launch_levels1<T, 8>(grid, lds, sg, sa); break;
    case 10: launch_levels1<T, 10>(grid, lds, sg, sa); break;
    case 12: launch_levels1<T, 12>(grid, lds, sg, sa); break;
    default:
        if (sizeof(T) == 4) {
            switch ((K + 1) / 2 * 2) {
            case 14: launch_levels1<float, 14>(grid, lds, sg, sa); break;
            case 16: launch_levels1<float, 16>(grid, lds, sg, sa); break;
            default: launch_levels1<float, 18>(grid, lds, sg, sa); break;
            }
        }
        break;
    }
}

// T = storage type of the caller's arrays and of every workspace plane
template <typename T>
static int execute_tiles_t(vrt_plan *p, int64_t nlam, int64_t ld, const T *dS, const T *dalpha,
                           int alpha_mode, const T *dI0_up, const T *dI0_down,
                           const double *weights_user, T *dJ, T *dI_out, hipStream_t st)
{
    constexpr bool kF32 = sizeof(T) == 4;
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int A = p->A;
    const bool patches = p->last_path == 4;          // fused patch kernel (vrt_patch.hip): same layouts as steps
    const bool steps = p->last_path == 3 || patches;
    // the pair level kernel (fp64 storage, layers <= 8192 sites) or the single-wavelength one
    // (VRT_STEP_SINGLE=1 selects the single-wavelength kernel on any grid: same results, for the tests)
    const bool single = steps && !patches && (kF32 || p->tile_max_layer_size > 8192 ||
                                  p->tune.step_single == 1);
    // storage layout: wavelength pairs side by side on the layer-step path, plain planes on the
    // persistent tile path (sw_index); planes are padded to a whole number of blocks
    const int lb = steps ? 2 : 1;
    const int64_t nl_pad = (nlam + lb - 1) / lb * lb;
    const size_t plane = (size_t)nl_pad * (size_t)n;
    // workspaces are kept as double buffers; a plane of T needs this many doubles
    auto dcount = [](size_t elems) { return (elems * sizeof(T) + 7) / 8; };
    int rc;
    if ((rc = ensure_dev(p->d_I, p->I_cap, dcount((size_t)std::max(1, A) * plane)))) return rc;
    T *wI = reinterpret_cast<T *>(p->d_I);
    const bool use_dir[2] = {p->n_up > 0, p->n_down > 0};
    for (int d = 0; d < 2; d++)
        if (use_dir[d] && (rc = ensure_dev(p->ws_S[d], p->ws_S_cap[d], dcount(plane)))) return rc;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    TileArgs ta;
    ta.n = n;
    ta.nlam = (int)nlam;
    ta.A = A;
    ta.alpha_mode = alpha_mode;
    ta.max_layers = p->tile_max_layers;
    ta.tile_stride = (int)((std::max<int64_t>(p->tile_max_layer_size, 1) + 2) & ~(int64_t)1);   // + the zero slot
    if (!steps && (rc = build_task_map(p, (int)nlam, st))) return rc;
    ta.task_map = p->d_task_map;
    ta.angle_dir = p->d_angle_dir;
    ta.nlev = p->d_nlev;
    ta.t_u1 = p->t_u1; ta.t_u2 = p->t_u2;
    ta.t_w1 = p->t_w1; ta.t_w2 = p->t_w2; ta.t_r1 = p->t_r1; ta.t_r2 = p->t_r2;
    ta.t_vis = p->t_vis;
    ta.t_loc = p->t_loc;
    ta.t_self = p->t_self;
    ta.t_vis_s = p->t_vis_s; ta.t_loc_s = p->t_loc_s;
    ta.t_gpos = p->t_gpos;
    ta.alpha_angle = nullptr;
    ta.I = p->d_I;
    ta.dbg = nullptr;
    for (int d = 0; d < 2; d++) {
        const Direction &dir = d == 0 ? g->up : g->down;
        ta.lay[d] = dir.d_lay;
        ta.nlayers[d] = (int)dir.reduced.size() - 1;
        ta.S[d] = nullptr;
        ta.alpha[d] = nullptr;
        if (!use_dir[d]) continue;
        hipLaunchKernelGGL(k_to_sweep_order<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, dir.d_store, dS,
                           reinterpret_cast<T *>(p->ws_S[d]));
        ta.S[d] = p->ws_S[d];
        if (alpha_mode == VRT_ALPHA_SITE) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], dcount((size_t)n)))) return rc;
            hipLaunchKernelGGL(k_gather_vec<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                               dir.d_store, dalpha, reinterpret_cast<T *>(p->ws_A[d]));
            ta.alpha[d] = p->ws_A[d];
        } else if (alpha_mode == VRT_ALPHA_SITE_LAM) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], dcount(plane)))) return rc;
            hipLaunchKernelGGL(k_to_sweep_order<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, dir.d_store,
                               dalpha, reinterpret_cast<T *>(p->ws_A[d]));
            ta.alpha[d] = p->ws_A[d];
        }
        const int cnt = d == 0 ? p->n_up : p->n_down;
        if (dir.n1 > 0) {
            const dim3 bgrid((unsigned)((dir.n1 + 63) / 64), (unsigned)((nlam + 63) / 64), (unsigned)cnt);
            hipLaunchKernelGGL(k_boundary_sweep_order<T>, bgrid, dim3(256), 0, st, n, (int)nlam, lb, dir.n1,
                               d == 0 ? p->d_angles_up : p->d_angles_down, dir.d_order, dir.d_srank,
                               d == 0 ? dI0_up : dI0_down, wI);
        }
    }
    if (alpha_mode == VRT_ALPHA_ANGLE_NATIVE) {
        // already in storage-pair order per active angle (vrt_plan_alpha_to_native_dev or the
        // opacity prologue wrote it): no transposed copy, the kernels read the caller's buffer
        ta.alpha_mode = VRT_ALPHA_ANGLE_SITE_LAM;
        ta.alpha_angle = reinterpret_cast<const double *>(dalpha);
    } else if (alpha_mode == VRT_ALPHA_ANGLE_SITE_LAM) {
        if ((rc = ensure_dev(p->ws_AA, p->ws_AA_cap, dcount((size_t)A * plane)))) return rc;
        for (int a = 0; a < A; a++) {
            const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
            hipLaunchKernelGGL(k_to_sweep_order<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, dir.d_store,
                               dalpha + (size_t)a * (size_t)n * (size_t)ld,
                               reinterpret_cast<T *>(p->ws_AA) + (size_t)a * plane);
        }
        ta.alpha_angle = p->ws_AA;
    }
    VRT_HIP_TRY(hipGetLastError());

    const bool debug = kDiag && p->tune.tile_debug;
    bool fused_dir[2] = {false, false};     // J_dir of the direction was reduced inside the sweep (patch path)
    long long *d_dbg = nullptr;
    int64_t launches = 1;
    if (steps && A > 0) {
        // ---- layer-step variant: 2 launches per BFS layer -------------------------------------
        const int stride = (int)((std::max<int64_t>(p->tile_max_layer_size, 1) + 63) & ~(int64_t)63);
        const int npair = (int)(nl_pad / 2);
        // hand-off buffers: pair kernel -> double2 per (angle, pair, slot); single-wavelength
        // kernel -> one plane of T per (angle, wavelength)
        const size_t cgn = single ? dcount((size_t)A * (size_t)nl_pad * (size_t)stride)
                                  : (size_t)A * (size_t)nl_pad * (size_t)stride;
        if (!patches) {
            if ((rc = ensure_dev(p->ws_cg[0], p->ws_cg_cap[0], cgn))) return rc;
            if ((rc = ensure_dev(p->ws_cg[1], p->ws_cg_cap[1], 2 * cgn))) return rc;
        }
        StepArgs sa;
        sa.ta = ta;
        sa.cg_stride = stride;
        sa.npair = npair;
        sa.cg_c = reinterpret_cast<double2 *>(p->ws_cg[0]);
        sa.cg_g = reinterpret_cast<double2 *>(p->ws_cg[1]);
        sa.t_rank_s = p->t_rank_s;
        sa.t_loc_ss = p->t_loc_ss;
        // pairs per coefficient thread: 4 to 6, whichever leaves the last group of an angle fullest (10
        // pairs: 5 + 5 instead of 4 + 4 + 2 -- C3 9.58 -> 9.36 ms; C4's 26 pairs stay at 4)
        sa.pairs_per_thread = kStepPairs;
        for (int c = kStepPairs + 1; c <= kStepPairs + 2; c++)
            if ((npair + c - 1) / c * c - npair < (npair + sa.pairs_per_thread - 1) / sa.pairs_per_thread * sa.pairs_per_thread - npair)
                sa.pairs_per_thread = c;
        if (p->tune.step_pairs > 0) sa.pairs_per_thread = p->tune.step_pairs;
        sa.chunks = (int)((p->tile_max_layer_size + 255) / 256);
        sa.xcd_map = p->tune.step_xcd;
        sa.debug_skip_levels = kDiag && p->tune.debug_skip_levels;
        // 1: S/alpha gathers off, 2: I gathers off, 4: coefficient stores off, 8: coefficient loads off,
        // 16: I stores off, 32: no linear_weights arithmetic, 64: level kernel keeps the storage-order thread assignment
        sa.debug_flags = kDiag ? p->tune.debug_flags : 0;
        if ((sa.debug_flags & ~(64 | 128)) || sa.debug_skip_levels) {   // (256, 512, 1024: single-wavelength level kernel)
            static bool warned = false;
            if (!warned) std::fprintf(stderr, "[vrt] VRT_DEBUG_FLAGS / VRT_DEBUG_SKIP_LEVELS set: timing diagnostics, the results are WRONG\n");
            warned = true;
        }
        const int Lmax = std::max(ta.nlayers[0] * (use_dir[0] ? 1 : 0), ta.nlayers[1] * (use_dir[1] ? 1 : 0));
        const int force_K = p->tune.step_K;
        // The angles are dealt (heaviest first) to a few internal streams that advance through
        // the layers independently: the (angle, wavelength) problems of different streams share
        // nothing, so one stream's launches fill the tail of the other's (612 level workgroups
        // are 2.4 rounds of the 256 CUs: a lone launch idles a fifth of the chip in its last round).
        const int G = std::max(1, std::min({p->tune.step_streams, 4, A}));
        if ((rc = ensure_step_streams(p, G))) return rc;
        // level workgroups -> XCDs: contiguous cost-balanced runs (VRT_STEP_LEVEL_MAP=0: round-robin)
        const bool use_map = !patches && p->tune.step_level_map != 0;
        if (use_map && (rc = build_level_map(p, G, single ? (int)nlam : npair))) return rc;
        if (patches && (rc = ensure_patch_work(p, G, p->h_step_angles, p->step_group_off))) return rc;
        // J reduction riding along the patch launches: a stream that holds ALL angles of a direction forms
        // J_dir of layer l - 1 in its launch of layer l (the layer is final, its lines still cache-resident)
        PatchReduce red_tmpl;
        int owner_of_dir[2] = {-1, -1};
        int64_t reduced_upto[2] = {0, 0};
        if (patches && dJ) {
            for (int a = 0; a < A; a++) red_tmpl.w[a] = weights_user[p->user_of_active[(size_t)a]];
            for (int d = 0; d < 2; d++) {
                if (!use_dir[d]) continue;
                if ((rc = ensure_dev(p->ws_J[d], p->ws_J_cap[d], dcount(plane)))) return rc;
                for (int gi = 0; gi < G; gi++) {
                    int have = 0;
                    for (int j = p->step_group_off[(size_t)gi]; j < p->step_group_off[(size_t)gi + 1]; j++)
                        have += (p->dir_of_active[(size_t)p->h_step_angles[(size_t)j]] > 0) == (d == 0);
                    if (have == (d == 0 ? p->n_up : p->n_down)) owner_of_dir[d] = gi;
                }
            }
        }
        fused_dir[0] = owner_of_dir[0] >= 0;
        fused_dir[1] = owner_of_dir[1] >= 0;
        auto make_reduce = [&](int gi, int layer_done, bool final, PatchReduce &red) -> bool {
            // ranges of the directions this group owns that became final with layer `layer_done`
            red = red_tmpl;
            int r = 0;
            for (int d = 0; d < 2; d++) {
                if (owner_of_dir[d] != gi) continue;
                const Direction &dir = d == 0 ? g->up : g->down;
                const int Ld = (int)dir.reduced.size() - 1;
                int64_t upto = reduced_upto[d];
                if (final) upto = n;
                else if (layer_done >= 1 && layer_done <= Ld) upto = dir.reduced[(size_t)layer_done] - 1;
                if (upto <= reduced_upto[d]) continue;
                red.lo[r] = (int)reduced_upto[d];
                red.hi[r] = (int)upto;
                red.Jd[r] = p->ws_J[d];
                red.count[r] = 0;
                for (int a = 0; a < A; a++)
                    if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) red.angles[r][red.count[r]++] = a;
                reduced_upto[d] = upto;
                r++;
            }
            return r > 0;
        };
        sa.level_map = nullptr;
        VRT_HIP_TRY(hipEventRecord(p->ev0, st));
        VRT_HIP_TRY(hipEventRecord(p->step_fork, st));
        launches = 0;
        // Launches are enqueued layer by layer across the streams (not stream by stream): the host
        // needs ~3.5 us per launch, so a stream whose 2 (L - 1) launches were queued behind all of
        // another stream's would start a millisecond late and finish alone.
        const int ngrp = (npair + sa.pairs_per_thread - 1) / sa.pairs_per_thread;
        if (G > 1)
            for (int gi = 1; gi < G; gi++) VRT_HIP_TRY(hipStreamWaitEvent(p->step_stream[gi], p->step_fork, 0));
        for (int layer = 2; layer <= Lmax; layer++) {
            sa.layer = layer;
            // launch geometry from THIS layer's size (the larger of the two directions'): layers
            // of a stratified tessellation differ severalfold
            int64_t cnt_l = 1;
            for (int d = 0; d < 2; d++) {
                const Direction &dir = d == 0 ? g->up : g->down;
                if (use_dir[d] && layer <= ta.nlayers[d])
                    cnt_l = std::max<int64_t>(cnt_l, dir.reduced[(size_t)layer] - dir.reduced[(size_t)layer - 1]);
            }
            sa.chunks = (int)((cnt_l + 255) / 256);
            const int per_xcd = (sa.chunks + 7) / 8;     // largest chunk range of an XCD
            // sites per thread of the level kernel: the fewest that cover the layer (register-
            // resident coefficients); VRT_STEP_K forces more (tests)
            const int need_K = (int)((cnt_l + 1023) / 1024);
            const int step_K = std::max(1, std::min(8, std::max(force_K, need_K)));
            for (int gi = 0; gi < G; gi++) {
                hipStream_t sg = gi == 0 ? st : p->step_stream[gi];     // group 0 on the caller's stream: one hardware queue less
                const int n_list = p->step_group_off[gi + 1] - p->step_group_off[gi];
                if (n_list == 0) continue;
                if (patches) {               // ONE fused launch per layer and stream
                    // pairs per workgroup: the plan's Q, or 1 when that would leave half of every group empty
                    int Q = p->tune.patch_Q;
                    if (!patch_shape_exists(p->patch_K, Q, p->patch_NT) || (npair % Q != 0 && npair < 2 * Q)) Q = 1;
                    PatchReduce red;
                    const bool have_red = make_reduce(gi, layer - 1, false, red);
                    if ((rc = launch_patch_layer(p, sa.ta, npair, layer, gi, Q, sg, kF32, have_red ? &red : nullptr))) return rc;
                    launches += 1;
                    continue;
                }
                sa.angle_list = p->d_step_angles + p->step_group_off[gi];
                sa.n_list = n_list;
                const size_t ntask_l = (size_t)n_list * (size_t)(single ? (int)nlam : npair);
                size_t lblocks = ntask_l;
                // only while a launch is a single round of the chip (<= one workgroup per CU): with several
                // rounds the fixed split costs more in balance than the shared L2 gains (C5: 150 -> 153 ms;
                // C3, 100 workgroups per launch: 9.83 -> 9.50 ms)
                sa.level_map = nullptr;
                if (use_map && ntask_l <= 256) {
                    sa.level_map = p->d_level_map + p->level_map_off[(size_t)gi];
                    lblocks = (size_t)(p->level_map_off[(size_t)gi + 1] - p->level_map_off[(size_t)gi]);
                }
                const dim3 g1(sa.xcd_map ? (unsigned)(8 * per_xcd * n_list * ngrp) : (unsigned)(sa.chunks * n_list * ngrp));
                if (single) {
                    hipLaunchKernelGGL((k_step_coeffs<T, true>), g1, dim3(256), 0, sg, sa);
                    const int K1 = std::max(need_K, std::min(force_K, kF32 ? kSingleMaxK32 : kSingleMaxK64));
                    launch_levels1_K<T>(std::max(K1, 1), dim3((unsigned)lblocks),
                                        (size_t)(cnt_l + 1) * sizeof(T), sg, sa);
                    launches += 2;
                    continue;
                }
                if constexpr (!kF32) {
                    const size_t lds = (size_t)(cnt_l + 1) * sizeof(double2);   // + the zero slot
                    const dim3 g2((unsigned)lblocks);
                    hipLaunchKernelGGL((k_step_coeffs<double, false>), g1, dim3(256), 0, sg, sa);
                    switch (step_K) {
                    case 1: hipLaunchKernelGGL(k_step_levels<1>, g2, dim3(1024), lds, sg, sa); break;
                    case 2: hipLaunchKernelGGL(k_step_levels<2>, g2, dim3(1024), lds, sg, sa); break;
                    case 3: hipLaunchKernelGGL(k_step_levels<3>, g2, dim3(1024), lds, sg, sa); break;
                    case 4: hipLaunchKernelGGL(k_step_levels<4>, g2, dim3(1024), lds, sg, sa); break;
                    case 5: hipLaunchKernelGGL(k_step_levels<5>, g2, dim3(1024), lds, sg, sa); break;
                    case 6: hipLaunchKernelGGL(k_step_levels<6>, g2, dim3(1024), lds, sg, sa); break;
                    case 7: hipLaunchKernelGGL(k_step_levels<7>, g2, dim3(1024), lds, sg, sa); break;
                    default: hipLaunchKernelGGL(k_step_levels<8>, g2, dim3(1024), lds, sg, sa); break;
                    }
                }
                launches += 2;
            }
        }
        if (patches)       // the last layers (and the never-visited site n - 1, whose intensity is 0)
            for (int gi = 0; gi < G; gi++) {
                PatchReduce red;
                if (!make_reduce(gi, 0, true, red)) continue;
                if ((rc = launch_patch_layer(p, sa.ta, npair, p->tile_max_layers + 1, gi, 1, gi == 0 ? st : p->step_stream[gi], kF32, &red)))
                    return rc;
                launches += 1;
            }
        if (G > 1)
            for (int gi = 1; gi < G; gi++) {
                VRT_HIP_TRY(hipEventRecord(p->step_join[gi], p->step_stream[gi]));
                VRT_HIP_TRY(hipStreamWaitEvent(st, p->step_join[gi], 0));
            }
        VRT_HIP_TRY(hipGetLastError());
        VRT_HIP_TRY(hipEventRecord(p->ev1, st));
    } else {
        if constexpr (!kF32) {
            if (debug && hipMalloc((void **)&d_dbg, sizeof(long long) * 4 * (size_t)A * (size_t)nlam) == hipSuccess) ta.dbg = d_dbg;
            VRT_HIP_TRY(hipEventRecord(p->ev0, st));
            if (A > 0) {
                const size_t lds = 2 * (size_t)ta.tile_stride * sizeof(double);
                const dim3 grid((unsigned)((size_t)A * (size_t)nlam));
                // layers of up to 3072 sites: 768 threads x 4 sites in ONE phase-1 batch (the 168 VGPRs of
                // 3 waves per SIMD hold its 48 loads); larger layers: 1024 threads, batches of two
                const bool wide = p->tile_max_layer_size <= 3072 && p->tune.tile_wide != 0;
                // layers of at most 4096 sites: the two-launch form (chip-wide I-independent
                // coefficients, then persistent level workgroups; VRT_TILE_PRE=0: the one-launch kernel)
                const bool pre = p->tile_max_layer_size <= kPreMaxLayer && p->t_code_ss &&
                                 p->tune.tile_pre != 0;
                if (pre) {
                    const size_t ntask = (size_t)A * (size_t)nlam;
                    if ((rc = ensure_dev(p->ws_cg[0], p->ws_cg_cap[0], 3 * ntask * (size_t)n))) return rc;
                    hipLaunchKernelGGL(k_tile_coeffs, dim3((unsigned)((n + 255) / 256), (unsigned)ntask), dim3(256), 0, st,
                                       ta, p->ws_cg[0]);
                    const size_t lds_pre = 3 * (size_t)ta.tile_stride * sizeof(double);
                    if (p->tile_max_layer_size <= 1536)
                        hipLaunchKernelGGL((k_sweep_tiles_pre<2, 768>), grid, dim3(768), lds_pre, st, ta, p->ws_cg[0], p->t_code_ss, p->t_rank_s);
                    else if (p->tile_max_layer_size <= 3072)
                        hipLaunchKernelGGL((k_sweep_tiles_pre<4, 768>), grid, dim3(768), lds_pre, st, ta, p->ws_cg[0], p->t_code_ss, p->t_rank_s);
                    else
                        hipLaunchKernelGGL((k_sweep_tiles_pre<4, 1024>), grid, dim3(1024), lds_pre, st, ta, p->ws_cg[0], p->t_code_ss, p->t_rank_s);
                    launches = 2;
                }
                else if (wide && p->tile_max_layer_size <= 1536)
                    hipLaunchKernelGGL((k_sweep_tiles<2, 2, 768>), grid, dim3(768), lds, st, ta);
                else if (wide)
                    hipLaunchKernelGGL((k_sweep_tiles<4, 4, 768>), grid, dim3(768), lds, st, ta);
                else
                    switch (p->tile_K) {
                    case 2: hipLaunchKernelGGL((k_sweep_tiles<2, 2, 1024>), grid, dim3(1024), lds, st, ta); break;
                    case 4: hipLaunchKernelGGL((k_sweep_tiles<4, 2, 1024>), grid, dim3(1024), lds, st, ta); break;
                    default: hipLaunchKernelGGL((k_sweep_tiles<8, 2, 1024>), grid, dim3(1024), lds, st, ta); break;
                    }
                VRT_HIP_TRY(hipGetLastError());
            }
            VRT_HIP_TRY(hipEventRecord(p->ev1, st));
        } else
            return fail(VRT_EINVAL, "the persistent tile kernel stores fp64 only");
    }
    p->ev_valid = true;
    p->last_launches = launches;
    if (d_dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<long long> h(4 * (size_t)A * (size_t)nlam);
        (void)hipMemcpy(h.data(), d_dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d_dbg);
        double s1 = 0, s2 = 0, s3 = 0;
        for (size_t t = 0; t < (size_t)A * (size_t)nlam; t++) { s1 += h[4 * t]; s2 += h[4 * t + 1]; s3 += h[4 * t + 2]; }
        const double nt = (double)A * (double)nlam;
        std::fprintf(stderr, "[vrt tiles] mean cycles per task (s_memtime, 100 MHz): phase1 %.0f phase2 %.0f phase3 %.0f; first task %lld %lld %lld, last task %lld %lld %lld\n",
                     s1 / nt, s2 / nt, s3 / nt, h[0], h[1], h[2], h[h.size() - 4], h[h.size() - 3], h[h.size() - 2]);
    }

    if (dJ) {
        T *Jd[2] = {nullptr, nullptr};
        for (int d = 0; d < 2; d++) {
            if (!use_dir[d]) continue;
            if ((rc = ensure_dev(p->ws_J[d], p->ws_J_cap[d], dcount(plane)))) return rc;
            DirWeights dw;
            dw.count = 0;
            for (int a = 0; a < A; a++)
                if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) {
                    dw.w[dw.count] = weights_user[p->user_of_active[(size_t)a]];
                    dw.idx[dw.count] = a;
                    dw.count++;
                }
            Jd[d] = reinterpret_cast<T *>(p->ws_J[d]);
            if (fused_dir[d]) continue;                  // formed layer by layer inside the sweep's launches
            hipLaunchKernelGGL(k_reduce_dir<T>, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st,
                               (int64_t)plane, (int64_t)plane, dw, wI, Jd[d]);
        }
        hipLaunchKernelGGL(k_combine_J<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, g->up.d_store,
                           g->down.d_srank, Jd[0], Jd[1], dJ);
        VRT_HIP_TRY(hipGetLastError());
    }
    if (dI_out) {
        std::vector<int> active_of_user((size_t)p->n_angles_user, -1);
        for (int a = 0; a < A; a++) active_of_user[(size_t)p->user_of_active[(size_t)a]] = a;
        for (int64_t u = 0; u < p->n_angles_user; u++) {
            const int a = active_of_user[(size_t)u];
            const Direction &dir = (a >= 0 && p->dir_of_active[(size_t)a] < 0) ? g->down : g->up;
            hipLaunchKernelGGL(k_from_sweep_order<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, dir.d_store,
                               a >= 0 ? wI + (size_t)a * plane : (const T *)nullptr,
                               dI_out + (size_t)u * (size_t)n * (size_t)ld);
        }
        VRT_HIP_TRY(hipGetLastError());
    }
    return VRT_OK;
}

int execute_tiles(vrt_plan *p, int64_t nlam, int64_t ld, const void *dS, const void *dalpha,
                  int alpha_mode, const void *dI0_up, const void *dI0_down,
                  const double *weights_user, void *dJ, void *dI_out, hipStream_t st, bool f32)
{
    if (f32)
        return execute_tiles_t<float>(p, nlam, ld, (const float *)dS, (const float *)dalpha, alpha_mode,
                                      (const float *)dI0_up, (const float *)dI0_down, weights_user, (float *)dJ,
                                      (float *)dI_out, st);
    return execute_tiles_t<double>(p, nlam, ld, (const double *)dS, (const double *)dalpha, alpha_mode,
                                   (const double *)dI0_up, (const double *)dI0_down, weights_user, (double *)dJ,
                                   (double *)dI_out, st);
}

// limits of the layer-step level kernels (sites per layer)
int64_t steps_max_layer(bool f32) { return f32 ? (int64_t)kSingleMaxK32 * 1024 : (int64_t)kSingleMaxK64 * 1024; }

}  // namespace vrt
