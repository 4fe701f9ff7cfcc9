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
//
// Kernel templates of the PERSISTENT tile path ("tiles"), included by vrt_layers.hip.
#pragma once

#include "vrt_device.h"

namespace vrt {

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

}  // namespace vrt
