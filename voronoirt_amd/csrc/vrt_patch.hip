// Fused patch kernel of the layer-ordered formal solve ("patches" path) for gfx950.
//
// One launch per BFS layer and direction stream.  A workgroup solves ONE patch of the layer -- a range
// of consecutive storage positions plus the halo of its in-layer dependency cone
// (vrt_patch.cpp) -- for one angle and a group of Q wavelength pairs, start to finish:
//   1. integration coefficients of every entry (own + halo sites) straight into registers:
//        I_c = c + g1 I_u1 + g2 I_u2,   c = Σ_r ((e_r I_ur [earlier layer] + a_r S_ur) + b_r S_c) w_r,
//        g_r = e_r w_r [upwind r in this layer]            (irregular_ray_tracing.jl:66-76 re-associated)
//   2. the patch's Gauss-Seidel levels on a private LDS tile (s_barrier between levels),
//   3. the final intensities of the sites it owns -> I (storage order, wavelength pairs).
// Nothing is handed from one kernel to another inside a layer (the layer-step path writes and
// re-reads 35 B of coefficients per cell-update), no workgroup waits for another one, and a layer
// may have any number of sites: it just has more patches.  The halo is recomputed, not exchanged:
// every visit performs the same arithmetic on the same values as in the unsplit layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "vrt_device.h"
#include "vrt_internal.h"

namespace vrt {

// ---- timing diagnostics of the -DVRT_DIAG build (tools/gather_flags.sh, chain_flags.sh, traffic_split.sh; WRONG results) ----
// Every switch in the kernels below goes through diag(): in the product build kDiag is false, diag() is the constant
// `false` and the switched statement is gone -- the hot loops read as what runs when every `diag(...)` is read as `false`.
enum : int {
    kDiagNoLevels = 1, kDiagCentreGathers = 2, kDiagNoWeights = 4, kDiagNoStores = 8, kDiagNoGatherI = 16, kDiagNoGatherAlpha = 32,
    kDiagNoGatherS = 64, kDiagNoReduce = 128, kDiagNoWait = 256, kDiagPlainLoadI = 512, kDiagPlainStoreI = 1024, kDiagNoPairLoop = 2048,
};
__host__ __device__ __forceinline__ constexpr bool diag(int dbg, int flag) { return kDiag && (dbg & flag) != 0; }

struct PatchArgs {
    TileArgs ta;              // n, nlam, alpha_mode, angle_dir, lay, nlayers, S, alpha, alpha_angle, I (pair planes)
    int npair;                // ceil(nlam / 2)
    int layer;                // 1-based BFS layer being solved
    int ngrp;                 // workgroups per work item = siblings (1 << lgB) x splits
    int nsplit, Q;            // the pair blocks are dealt to nsplit workgroups per sibling in steps of Q blocks (split_blocks)
    int lgB;                  // log2 of the pairs per block of the storage layout (vrt_device.h: pair_block_at)
    int stride;               // tile slots per pair plane (> largest entry count: + the zero slot)
    int cap;                  // entries per patch at most (K * NT): length of the LDS table arrays
    int quad;                 // fp32 storage: k_patch_quad (two neighbouring pairs of a block per workgroup)
    int lean;                 // k_patch_lean (64 registers: four workgroups per CU)
    int dbg;                  // timing diagnostics (-DVRT_DIAG build only, WRONG results): 1 no levels, 2 gathers ->
                              //   coalesced centre reads, 4 no weights arithmetic, 8 no stores, 16 / 32 / 64 no upwind gathers
                              //   of I / alpha / S, 128 no J reduction
    // J reduction riding along (lagged by one layer): the first nred blocks of the launch do not solve a patch
    // but form J_dir = Σ_a w_a I_a (reference's angle order inside the direction) over storage positions
    // [red_lo, red_hi) of up to two directions -- layers the stream's previous launch has made final
    PatchReduce red;
    // this launch's work list, two int4 per (slot, XCD) -- everything a workgroup needs to know about its item in ONE
    // round trip: {first entry, entries (0: a padding slot), first owned storage position, owned sites},
    // {in-layer levels, active angle | direction << 16, the layer's first storage position, the next layer's}
    const int4 *wrec;
    const int32_t *e_pos, *e_u1, *e_u2;     // per entry: storage position of the site and of its two upwinds
    const uint32_t *e_vis, *e_loc;           //   packed visit levels; patch-local tile slots of the upwinds
    const double *e_w1, *e_w2, *e_r1, *e_r2; //   weights and path lengths (irregular_ray_tracing.jl:51,66)
};

// entry tables of one angle from its storage-order tables
__global__ void __launch_bounds__(256)
k_patch_entries(int64_t count, const int32_t *__restrict__ e_pos, const int32_t *__restrict__ t_u1,
                const int32_t *__restrict__ t_u2, const double *__restrict__ t_w1, const double *__restrict__ t_w2,
                const double *__restrict__ t_r1, const double *__restrict__ t_r2, int32_t *__restrict__ e_u1,
                int32_t *__restrict__ e_u2, double *__restrict__ e_w1, double *__restrict__ e_w2,
                double *__restrict__ e_r1, double *__restrict__ e_r2)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int p = e_pos[e];
    e_u1[e] = t_u1[p]; e_u2[e] = t_u2[p];
    e_w1[e] = t_w1[p]; e_w2[e] = t_w2[p];
    e_r1[e] = t_r1[p]; e_r2[e] = t_r2[p];
}

int launch_patch_entries(vrt_plan *p, int a, int64_t first, int64_t count)
{
    if (count <= 0) return VRT_OK;
    const size_t o = (size_t)a * (size_t)p->g->n;
    hipLaunchKernelGGL(k_patch_entries, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, p->g->stream, count,
                       p->e_pos + first, p->t_u1 + o, p->t_u2 + o, p->t_w1 + o, p->t_w2 + o, p->t_r1 + o,
                       p->t_r2 + o, p->e_u1 + first, p->e_u2 + first, p->e_w1 + first, p->e_w2 + first,
                       p->e_r1 + first, p->e_r2 + first);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// exp(-x) for 5e-4 <= x <= 50: the table-driven exp_neg_tab of vrt_device.h (every patch kernel fills the table)
__device__ __forceinline__ double exp_neg10(double x) { return exp_neg_tab(x); }

// The kernel is bound by its fp64 arithmetic (two exponentials and a dozen weights per site, angle and
// wavelength; MI355X issues a wave's fp64 instruction in 4 cycles), so the weights are written with explicit
// fused multiply-adds -- a third fewer instructions than the reference's expression order, results within a few
// ulp of it (the parity contract is 1e-10; the build-wide -ffp-contract=off stays for the neighbour search).
//
// linear_weights (functions.jl:484-500) without control flow inside a lane; `MODE` is wave-uniform:
//   0  no lane has 5e-4 <= Δτ <= 50: thin or thick only, no exponential (optically thin upper layers and
//      thick bottom layers are most of a stratified atmosphere; a wave's lanes are neighbouring sites of a layer)
//   1  no lane is thin: no Taylor branch
//   2  general
// The thick branch (Δτ > 50: e = 0, a = 1/Δτ, b = 1 - a) needs no select for a and b: with e = exp(-50) = 2e-22
// the general formulas round to exactly those values; only e itself is set to 0.
template <int MODE>
__device__ __forceinline__ void lin_weights_fma(double dtau, double &a, double &b, double &e)
{
    double rc = __builtin_amdgcn_rcp(dtau);                 // only consumed when dtau >= 5e-4
    rc = fma(fma(-dtau, rc, 1.0), rc, rc);                  // v_rcp_f64 is good to ~2^-23: one Newton step -> 1e-14
    double e_thin = 0.0, a_thin = 0.0, b_thin = 0.0;
    if (MODE != 1) {
        e_thin = fma(dtau, fma(0.5, dtau, -1.0), 1.0);
        a_thin = dtau * fma(dtau, -1.0 / 3.0, 0.5);
        b_thin = dtau * fma(dtau, -1.0 / 6.0, 0.5);
    }
    const bool thin = dtau < 5e-4;
    if (MODE == 0) {
        e = thin ? e_thin : 0.0;
        a = thin ? a_thin : rc;
        b = thin ? b_thin : 1.0 - rc;
        return;
    }
    const double ee = exp_neg10(fmin(dtau, 50.0));
    const double a_mid = fma(1.0 - ee, rc, -ee), b_mid = (1.0 - a_mid) - ee;
    const double e_mid = dtau > 50.0 ? 0.0 : ee;
    if (MODE == 1) {
        e = e_mid; a = a_mid; b = b_mid;
    } else {
        e = thin ? e_thin : e_mid;
        a = thin ? a_thin : a_mid;
        b = thin ? b_thin : b_mid;
    }
}

// one wavelength of an entry: both upwinds' shares of a visit, t_r = ((e_r I_ur + a_r S_ur) + b_r S_c) w_r with
// I_ur gathered as 0 unless upwind r lies in an earlier layer; g_r = e_r wg_r, wg_r = w_r if upwind r lies in the
// site's own layer, else 0.  c = t_1 + t_2.  dt_r = r_r (α_c + α_ur) / 2 (trapezoidal, functions.jl:393).
template <int MODE>
__device__ __forceinline__ void entry_terms(double dt1, double dt2, double w1, double w2, double wg1, double wg2,
                                            double S_c, double S_1, double S_2, double I_1, double I_2, double &c,
                                            double &g1, double &g2)
{
    double ca1, cb1, ce1, ca2, cb2, ce2;
    lin_weights_fma<MODE>(dt1, ca1, cb1, ce1);
    lin_weights_fma<MODE>(dt2, ca2, cb2, ce2);
    const double t1 = fma(cb1, S_c, fma(ce1, I_1, ca1 * S_1)) * w1;
    const double t2 = fma(cb2, S_c, fma(ce2, I_2, ca2 * S_2)) * w2;
    c = t1 + t2;
    g1 = ce1 * wg1;
    g2 = ce2 * wg2;
}

// the same with the wave-uniform choice of MODE from the two optical depths of every lane
__device__ __forceinline__ void entry_lambda(double rh1, double rh2, double w1, double w2, double wg1, double wg2,
                                             double a_c, double a_1, double a_2, double S_c, double S_1, double S_2,
                                             double I_1, double I_2, double &c, double &g1, double &g2)
{
    const double d1 = rh1 * (a_c + a_1), d2 = rh2 * (a_c + a_2);
    const bool mid = ((d1 >= 5e-4) & (d1 <= 50.0)) | ((d2 >= 5e-4) & (d2 <= 50.0));
    const bool thin = (d1 < 5e-4) | (d2 < 5e-4);
    if (__ballot(mid) == 0ull) entry_terms<0>(d1, d2, w1, w2, wg1, wg2, S_c, S_1, S_2, I_1, I_2, c, g1, g2);
    else if (__ballot(thin) == 0ull) entry_terms<1>(d1, d2, w1, w2, wg1, wg2, S_c, S_1, S_2, I_1, I_2, c, g1, g2);
    else entry_terms<2>(d1, d2, w1, w2, wg1, wg2, S_c, S_1, S_2, I_1, I_2, c, g1, g2);
}

// The same, one upwind at a time: `next` (the optical depth the following evaluation starts from) is tied to this
// one's results by a compiler fence, so that the four evaluations of an entry's pair follow each other instead of
// being interleaved (the compiler's own order needs 72 registers, this one 64).  The
// weights w_r are read from the thread's LDS slots where they are used (pw1, pw2), not held.
template <int MODE>
__device__ __forceinline__ void entry_terms_seq(double dt1, double dt2, const double *pw1, const double *pw2, bool in1,
                                                bool in2, double S_c, double S_1, double S_2, double I_1, double I_2,
                                                double &c, double &g1, double &g2, double &next)
{
    double ca, cb, ce;
    lin_weights_fma<MODE>(dt1, ca, cb, ce);
    const double w1 = *pw1;
    double t1 = fma(cb, S_c, fma(ce, I_1, ca * S_1)) * w1;
    g1 = in1 ? ce * w1 : 0.0;
    asm volatile("" : "+v"(t1), "+v"(g1), "+v"(dt2));
    lin_weights_fma<MODE>(dt2, ca, cb, ce);
    const double w2 = *pw2;
    const double t2 = fma(cb, S_c, fma(ce, I_2, ca * S_2)) * w2;
    c = t1 + t2;
    g2 = in2 ? ce * w2 : 0.0;
    asm volatile("" : "+v"(c), "+v"(g2), "+v"(next));
}
__device__ __forceinline__ void entry_lambda_seq(double d1, double d2, const double *pw1, const double *pw2, bool in1,
                                                 bool in2, double S_c, double S_1, double S_2, double I_1, double I_2,
                                                 double &c, double &g1, double &g2, double &next)
{
    const bool mid = ((d1 >= 5e-4) & (d1 <= 50.0)) | ((d2 >= 5e-4) & (d2 <= 50.0));
    const bool thin = (d1 < 5e-4) | (d2 < 5e-4);
    if (__ballot(mid) == 0ull) entry_terms_seq<0>(d1, d2, pw1, pw2, in1, in2, S_c, S_1, S_2, I_1, I_2, c, g1, g2, next);
    else if (__ballot(thin) == 0ull) entry_terms_seq<1>(d1, d2, pw1, pw2, in1, in2, S_c, S_1, S_2, I_1, I_2, c, g1, g2, next);
    else entry_terms_seq<2>(d1, d2, pw1, pw2, in1, in2, S_c, S_1, S_2, I_1, I_2, c, g1, g2, next);
}

// The same visit with the upwind intensities applied LAST (the data-as-flag chained launch, where a workgroup waits for
// exactly those): everything that does not need I_1, I_2 -- the four weights, a_r S_ur, the couplings -- is formed while the
// gathers are in flight or repeated, and what is left behind the wait is three dependent operations per upwind.  The
// same operations on the same values in the same association as entry_terms_seq: bit-identical.
struct LateTerms { double ce1, p1, cb1, ce2, p2, cb2; };
template <int MODE>
__device__ __forceinline__ void late_coeffs(double dt1, double dt2, double S_1, double S_2, LateTerms &L, double &next)
{
    double ca, cb, ce;
    lin_weights_fma<MODE>(dt1, ca, cb, ce);
    L.ce1 = ce; L.p1 = ca * S_1; L.cb1 = cb;
    asm volatile("" : "+v"(L.ce1), "+v"(L.p1), "+v"(L.cb1), "+v"(dt2));
    lin_weights_fma<MODE>(dt2, ca, cb, ce);
    L.ce2 = ce; L.p2 = ca * S_2; L.cb2 = cb;
    asm volatile("" : "+v"(L.ce2), "+v"(L.p2), "+v"(L.cb2), "+v"(next));
}
__device__ __forceinline__ void late_lambda(double d1, double d2, double S_1, double S_2, LateTerms &L, double &next)
{
    const bool mid = ((d1 >= 5e-4) & (d1 <= 50.0)) | ((d2 >= 5e-4) & (d2 <= 50.0));
    const bool thin = (d1 < 5e-4) | (d2 < 5e-4);
    if (__ballot(mid) == 0ull) late_coeffs<0>(d1, d2, S_1, S_2, L, next);
    else if (__ballot(thin) == 0ull) late_coeffs<1>(d1, d2, S_1, S_2, L, next);
    else late_coeffs<2>(d1, d2, S_1, S_2, L, next);
}
__device__ __forceinline__ void late_apply(const LateTerms &L, const double *pw1, const double *pw2, bool in1, bool in2,
                                           double S_c, double I_1, double I_2, double &c, double &g1, double &g2)
{
    const double w1 = *pw1, w2 = *pw2;
    const double t1 = fma(L.cb1, S_c, fma(L.ce1, I_1, L.p1)) * w1;
    const double t2 = fma(L.cb2, S_c, fma(L.ce2, I_2, L.p2)) * w2;
    c = t1 + t2;
    g1 = in1 ? L.ce1 * w1 : 0.0;
    g2 = in2 ? L.ce2 * w2 : 0.0;
}

// wavelength pair `idx` of a plane: 32-bit byte offset from a wave-uniform base (planes are < 4 GiB: n < 2^28),
// so the load takes the saddr + voffset form -- one address VGPR, no 64-bit vector arithmetic
template <typename T2>
__device__ __forceinline__ double2 ldpair(const T2 *base, int idx, int sh)    // sh = log2(bytes per site of the block)
{
    const unsigned off = (unsigned)idx << sh;
    return to_d2(*reinterpret_cast<const T2 *>(reinterpret_cast<const char *>(base) + off));
}
template <typename T2> struct Log2Size;
template <> struct Log2Size<double2> { static constexpr int value = 4; };
template <> struct Log2Size<float2> { static constexpr int value = 3; };

// blocks [b0, b1) of split number `split`: the steps (Q blocks each) are dealt evenly, the first (steps % nsplit)
// splits taking one more (26 pairs over 5 splits: 6, 5, 5, 5, 5 -- not 6, 6, 6, 6, 2)
__host__ __device__ __forceinline__ void split_blocks(const PatchArgs &pa, int split, int nblock, int &b0, int &b1)
{
    const int nstep = (nblock + pa.Q - 1) / pa.Q;
    const int base = nstep / pa.nsplit, rem = nstep - base * pa.nsplit;
    const int s0 = split * base + (split < rem ? split : rem), s1 = s0 + base + (split < rem ? 1 : 0);
    b0 = s0 * pa.Q;
    b1 = nblock < s1 * pa.Q ? nblock : s1 * pa.Q;
}

// ---- reduction role of a patch launch: J_dir of a finished layer (NT x ppb pair elements per block) --------------
// which blocks of the launch take it: the first nred (VRT_REDUCE_LAST=0) or the last nred -- behind the patches, whose
// slowest workgroup ends the launch, the short reduction blocks fill the slots the fast patches leave
#ifndef VRT_REDUCE_LAST
#define VRT_REDUCE_LAST 1
#endif
__device__ __forceinline__ int reduce_block_index(const PatchArgs &pa)      // >= 0: this block reduces
{
    return VRT_REDUCE_LAST ? (int)blockIdx.x - ((int)gridDim.x - pa.red.nred) : ((int)blockIdx.x < pa.red.nred ? (int)blockIdx.x : -1);
}
__device__ __forceinline__ int patch_block_index(const PatchArgs &pa)
{
    return VRT_REDUCE_LAST ? (int)blockIdx.x : (int)blockIdx.x - pa.red.nred;
}
template <typename T, int NT>
__device__ __forceinline__ void patch_reduce_role(const PatchArgs &pa)
{
    typedef typename Pair<T>::type T2;
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    int b = reduce_block_index(pa), r = 0;
    if (b >= pa.red.nblk[0]) { b -= pa.red.nblk[0]; r = 1; }
    if (b >= pa.red.nblk[r]) return;                         // padding to a multiple of 8
    if (diag(pa.dbg, kDiagNoReduce)) return;
    // the range's pair elements of a pair block [k0, k0 + 2^lw) are one contiguous run of the plane,
    // (hi - lo) << lw long: blocks are dealt per pair block (size_reduce counts them the same way)
    const int len = pa.red.hi[r] - pa.red.lo[r];
    const int per = NT * pa.red.ppb;
    const int nblock = pair_block_count(pa.npair, pa.lgB);
    int k0 = 0, lw = 0;
    for (int k = 0; k < nblock; k++) {
        pair_block_of(k, pa.npair, pa.lgB, k0, lw);
        const int cnt = (int)((((int64_t)len << lw) + per - 1) / per);
        if (b < cnt) break;
        b -= cnt;
    }
    const int64_t nn = ta.n;
    const size_t run = (size_t)len << lw;
    const size_t base = (size_t)k0 * (size_t)nn + ((size_t)pa.red.lo[r] << lw);
    T2 *Jd = reinterpret_cast<T2 *>(pa.red.Jd[r]);
    const T2 *I0 = reinterpret_cast<const T2 *>(ta.I);
    const size_t plane = (size_t)pa.npair * (size_t)nn;
    if (pa.red.ppb == 4) {
        // four elements of a thread side by side: their loads of an angle's plane are in flight together (one element
        // after the other the block is four dependent round trips long -- and the launch ends with its last block);
        // per element the same sum in the same order
        // (elements of a block are NT apart: one 32-bit offset from a wave-uniform base; a plane set holds < 2^31 pairs)
        const size_t f0 = (size_t)b * 4 * NT + tid;
        const unsigned left = f0 < run ? (unsigned)min((size_t)(run - f0 + NT - 1) / NT, (size_t)4) : 0u;   // elements of this thread
        const size_t e0 = base + (left ? f0 : 0);
        double ax[4], ay[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { ax[i] = 0.0; ay[i] = 0.0; }
        for (int j = 0; j < pa.red.count[r]; j++) {          // the reference's angle order (lambda_iteration.jl:84,102,107)
            const int a = pa.red.angles[r][j];
            const T2 *Ia = I0 + (size_t)a * plane + e0;
            double2 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = to_d2(Ia[(unsigned)i < left ? i * NT : 0]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ax[i] += pa.red.w[a] * v[i].x;
                ay[i] += pa.red.w[a] * v[i].y;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((unsigned)i < left) Jd[e0 + (size_t)i * NT] = from_d2<T>(make_double2(ax[i], ay[i]));
        return;
    }
    for (int i = 0; i < pa.red.ppb; i++) {
        const size_t f = ((size_t)b * pa.red.ppb + i) * NT + tid;
        if (f >= run) break;
        const size_t e = base + f;
        double ax = 0.0, ay = 0.0;
        for (int j = 0; j < pa.red.count[r]; j++) {          // the reference's angle order (lambda_iteration.jl:84,102,107)
            const int a = pa.red.angles[r][j];
            const double2 v = to_d2(I0[(size_t)a * plane + e]);
            ax += pa.red.w[a] * v.x;
            ay += pa.red.w[a] * v.y;
        }
        Jd[e] = from_d2<T>(make_double2(ax, ay));
    }
}

// T: storage type of S, α, I; AM: alpha mode (VRT_ALPHA_SITE, _SITE_LAM, _ANGLE_SITE_LAM); K entries per thread;
// Q wavelength pairs solved at a time; NT threads.  A workgroup walks `ppw` wavelength pairs with ONE read of its
// patch's entry table, Q pairs at a time.
#ifdef VRT_PATCH_WPE            // experiments: force the register budget of that many waves per SIMD
#define VRT_WPE_ATTR __attribute__((amdgpu_waves_per_eu(VRT_PATCH_WPE, VRT_PATCH_WPE)))
#else
#define VRT_WPE_ATTR
#endif
template <typename T, int AM, int K, int Q, int NT>
__global__ void __launch_bounds__(NT) VRT_WPE_ATTR
k_patch_solve(PatchArgs pa)
{
    typedef typename Pair<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    // block -> (work item, pair group): blocks b, b + 8, ... share an XCD (MI355X_MICROARCH.md, speed
    // only); the pair groups of an item follow each other on ONE XCD and read its entry tables
    // through that L2, and consecutive items of an XCD are neighbouring patches / angles of a patch
    if (reduce_block_index(pa) >= 0) {
        patch_reduce_role<T, NT>(pa);
        return;
    }
    const int bid = patch_block_index(pa);
    const int x = bid & 7, rr = bid >> 3;
    const int grp = rr % pa.ngrp, sj = rr / pa.ngrp;
    const int4 rec = pa.wrec[2 * (sj * 8 + x)], rec2 = pa.wrec[2 * (sj * 8 + x) + 1];
    if (rec.y <= 0) return;
    // sibling sib solves pair k0 + sib of every block [k0, k0 + 2^lw) with 2^lw > sib among blocks b0 .. b1-1: the
    // 2^lgB siblings of an item run side by side on one XCD and use a gathered line (one site's pairs) whole
    const int sib = grp & ((1 << pa.lgB) - 1);
    const int nblock = pair_block_count(pa.npair, pa.lgB);
    int b0, b1;
    split_blocks(pa, grp >> pa.lgB, nblock, b0, b1);
    if (b0 >= b1) return;
    {
        int k0, lw;
        pair_block_of(b0, pa.npair, pa.lgB, k0, lw);
        if (sib >= (1 << lw)) return;                           // block widths only shrink: nothing for this sibling
    }
    const int ent_off = rec.x, n_ent = rec.y, own_lo = rec.z, own_cnt = rec.w;
    const int dbg = kDiag ? pa.dbg : 0;
    const int nlev = diag(dbg, kDiagNoLevels) ? 0 : rec2.x, a = rec2.y & 0xFFFF;
    const int d = rec2.y >> 16;
    const int lo = rec2.z, hi = rec2.w;
    const int64_t n = ta.n;
    const int stride = pa.stride;

    // ---- the patch's entry table, kept for every pair of this workgroup: each thread parks the entries it
    // owns in LDS slots only it ever reads (a register file extension: no barrier, no bank conflict) -- the
    // table costs no registers across the gather and level phases, which decides how many workgroups a CU holds
    double *s_w1 = reinterpret_cast<double *>(ptile + Q * stride);
    double *s_w2 = s_w1 + pa.cap, *s_r1 = s_w2 + pa.cap, *s_r2 = s_r1 + pa.cap;
    int *s_pos = reinterpret_cast<int *>(s_r2 + pa.cap);
    int *s_u1 = s_pos + pa.cap, *s_u2 = s_u1 + pa.cap;
    uint32_t *s_vis = reinterpret_cast<uint32_t *>(s_u2 + pa.cap), *s_loc = s_vis + pa.cap;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = tid + k * NT;
        const bool ok = i < n_ent;
        const int e = ent_off + (ok ? i : n_ent - 1);
        s_pos[i] = pa.e_pos[e];
        s_u1[i] = pa.e_u1[e];
        s_u2[i] = pa.e_u2[e];
        s_vis[i] = ok ? pa.e_vis[e] : 0u;
        // an upwind outside the cone reads the zero slot (coupling x a finite 0)
        const uint32_t lc = pa.e_loc[e], l1 = lc & 0xFFFFu, l2 = lc >> 16;
        s_loc[i] = (l1 == 0xFFFFu ? (uint32_t)n_ent : l1) | ((l2 == 0xFFFFu ? (uint32_t)n_ent : l2) << 16);
        s_w1[i] = pa.e_w1[e]; s_w2[i] = pa.e_w2[e]; s_r1[i] = pa.e_r1[e]; s_r2[i] = pa.e_r2[e];
    }
    if (tid == 0) {
#pragma unroll
        for (int qi = 0; qi < Q; qi++) ptile[qi * stride + n_ent] = make_double2(0.0, 0.0);   // the zero slot
    }
    exp2_table_fill();
    __syncthreads();
    constexpr int lgT2 = Log2Size<T2>::value;
    for (int bk = b0; bk < b1; bk += Q) {
        // the Q pairs of this step: (element base of the block + sibling, byte shift of a site); a step past the
        // sibling's last block repeats the previous pair and stores nothing
        size_t qbase[Q];
        int qsh[Q];
        bool qok[Q];
#pragma unroll
        for (int qi = 0; qi < Q; qi++) {
            int k0, lw;
            pair_block_of(min(bk + qi, b1 - 1), pa.npair, pa.lgB, k0, lw);
            qok[qi] = bk + qi < b1 && sib < (1 << lw);
            if (!qok[qi] && qi > 0) { qbase[qi] = qbase[qi - 1]; qsh[qi] = qsh[qi - 1]; continue; }
            qbase[qi] = (size_t)k0 * (size_t)n + (size_t)sib;
            qsh[qi] = lw + lgT2;
        }
        if (!qok[0]) break;
        // ---- integration coefficients of the entries for pairs q0 .. q0 + Q - 1 --------------------------
        double2 c[K][Q], g1[K][Q], g2[K][Q];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = tid + k * NT;
            const int p = s_pos[i];
            int v1 = s_u1[i], v2 = s_u2[i];
            if (diag(dbg, kDiagCentreGathers)) { v1 = p; v2 = p; }
            // the intensity of an upwind counts when it lies in an EARLIER layer (final); an upwind in this
            // layer enters through the tile, one in a later layer reads 0 (:23): those gather the never-visited
            // site perm[n] at storage position n - 1, whose intensity is 0 in every plane
            int i1 = v1 < lo ? v1 : (int)n - 1, i2 = v2 < lo ? v2 : (int)n - 1;
            if (diag(dbg, kDiagNoGatherI)) { i1 = (int)n - 1; i2 = (int)n - 1; }
            const int av1 = diag(dbg, kDiagNoGatherAlpha) ? p : v1, av2 = diag(dbg, kDiagNoGatherAlpha) ? p : v2;       // traffic split (diagnostic build)
            const int sv1 = diag(dbg, kDiagNoGatherS) ? p : v1, sv2 = diag(dbg, kDiagNoGatherS) ? p : v2;
            const bool in1 = (v1 >= lo) & (v1 < hi), in2 = (v2 >= lo) & (v2 < hi);
#pragma unroll
            for (int qi = 0; qi < Q; qi++) {
                const int sh = qsh[qi];
                const T2 *__restrict__ S = reinterpret_cast<const T2 *>(ta.S[d]) + qbase[qi];
                const T2 *__restrict__ I = reinterpret_cast<const T2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n + qbase[qi];
                // every load of the entry first (8 independent 16-byte gathers in flight), arithmetic after
                double2 a_c, a_1, a_2;
                if constexpr (AM == VRT_ALPHA_SITE) {                      // one opacity per site for every λ
                    const T *__restrict__ Al = reinterpret_cast<const T *>(ta.alpha[d]);
                    const double c0 = Al[p], c1 = Al[av1], c2 = Al[av2];
                    (void)sh;
                    a_c = make_double2(c0, c0); a_1 = make_double2(c1, c1); a_2 = make_double2(c2, c2);
                } else {
                    const T2 *__restrict__ Al =
                        AM == VRT_ALPHA_SITE_LAM
                            ? reinterpret_cast<const T2 *>(ta.alpha[d]) + qbase[qi]
                            : reinterpret_cast<const T2 *>(ta.alpha_angle) + (size_t)a * pa.npair * (size_t)n + qbase[qi];
                    a_c = ldpair(Al, p, sh); a_1 = ldpair(Al, av1, sh); a_2 = ldpair(Al, av2, sh);
                }
                const double2 S_c = ldpair(S, p, sh), S_1 = ldpair(S, sv1, sh), S_2 = ldpair(S, sv2, sh);
                const double2 I_1 = ldpair(I, i1, sh), I_2 = ldpair(I, i2, sh);
                const double w1 = s_w1[i], w2 = s_w2[i], r1 = s_r1[i], r2 = s_r2[i];
                const double wg1 = in1 ? w1 : 0.0, wg2 = in2 ? w2 : 0.0;
                const double rh1 = 0.5 * r1, rh2 = 0.5 * r2;               // exact: r (α_c + α_u) / 2 = (r / 2)(α_c + α_u)
                if (diag(dbg, kDiagNoWeights)) {
                    c[k][qi] = make_double2(a_c.x + S_c.x + I_1.x + a_1.x + S_1.x, a_c.y + S_c.y + I_2.y + a_2.y + S_2.y);
                    g1[k][qi] = make_double2(w1, w2); g2[k][qi] = make_double2(r1, r2);
                    continue;
                }
                entry_lambda(rh1, rh2, w1, w2, wg1, wg2, a_c.x, a_1.x, a_2.x, S_c.x, S_1.x, S_2.x, I_1.x, I_2.x,
                             c[k][qi].x, g1[k][qi].x, g2[k][qi].x);
                entry_lambda(rh1, rh2, w1, w2, wg1, wg2, a_c.y, a_1.y, a_2.y, S_c.y, S_1.y, S_2.y, I_1.y, I_2.y,
                             c[k][qi].y, g1[k][qi].y, g2[k][qi].y);
            }
        }
        // ---- the patch's Gauss-Seidel levels on the LDS tile: plane qi at ptile + qi * stride ----------------
        // (a thread only ever WRITES its own slots; the previous step's level loop ended with a barrier, so
        // nobody still reads them)
        uint32_t vis[K], loc[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            vis[k] = s_vis[tid + k * NT];
            loc[k] = s_loc[tid + k * NT];
            if (tid + k * NT < n_ent) {
#pragma unroll
                for (int qi = 0; qi < Q; qi++) ptile[qi * stride + tid + k * NT] = make_double2(0.0, 0.0);   // I = zero(S), :23
            }
        }
        __syncthreads();
        for (int t = 1; t <= nlev; t++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if ((vis[k] & 0xFFu) == (uint32_t)t) {          // a site's visits come at increasing levels
                    const uint32_t l1 = loc[k] & 0xFFFFu, l2 = loc[k] >> 16;
#pragma unroll
                    for (int qi = 0; qi < Q; qi++) {
                        const double2 xv = ptile[qi * stride + l1], yv = ptile[qi * stride + l2];
                        double2 r;
                        r.x = fma(g2[k][qi].x, yv.x, fma(g1[k][qi].x, xv.x, c[k][qi].x));
                        r.y = fma(g2[k][qi].y, yv.y, fma(g1[k][qi].y, xv.y, c[k][qi].y));
                        ptile[qi * stride + tid + k * NT] = r;
                    }
                    vis[k] >>= 8;
                }
            }
            __syncthreads();
        }
        // ---- final intensities of the owned sites (entries 0 .. own_cnt-1 = positions own_lo ..; own slots) ----
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = tid + k * NT;
            if (i < own_cnt) {
#pragma unroll
                for (int qi = 0; qi < Q; qi++) {
                    if (qok[qi] && !(diag(dbg, kDiagNoStores) && ptile[qi * stride + i].x != 1.2345e300)) {
                        T2 *I = reinterpret_cast<T2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n + qbase[qi];
                        const unsigned off = (unsigned)(own_lo + i) << qsh[qi];
                        *reinterpret_cast<T2 *>(reinterpret_cast<char *>(I) + off) = from_d2<T>(ptile[qi * stride + i]);
                    }
                }
            }
        }
    }
}


// ---- what the one-entry-per-thread kernels (k_patch_lean, k_patch_quad, k_patch_chain) share --------------------------
// the work of this workgroup: patch, sibling number inside a pair block, blocks [b0, b1), the patch's record
struct PatchItem {
    int sib, b0, b1;
    int ent_off, n_ent, own_lo, own_cnt, nlev, a, d, lo, hi;
};
// lgS: log2 of the sibling workgroups per pair block (lgB, or lgB - 1 when a workgroup takes two pairs of a block);
// false: a padding slot, or nothing of the item for this workgroup
__device__ __forceinline__ bool patch_item(const PatchArgs &pa, int lgS, PatchItem &it)
{
    const int bid = patch_block_index(pa);
    const int x = bid & 7, rr = bid >> 3;
    const int grp = rr % pa.ngrp, sj = rr / pa.ngrp;
    const int4 rec = pa.wrec[2 * (sj * 8 + x)], rec2 = pa.wrec[2 * (sj * 8 + x) + 1];
    if (rec.y <= 0) return false;
    it.sib = grp & ((1 << lgS) - 1);
    split_blocks(pa, grp >> lgS, pair_block_count(pa.npair, pa.lgB), it.b0, it.b1);
    if (it.b0 >= it.b1) return false;
    it.ent_off = rec.x; it.n_ent = rec.y; it.own_lo = rec.z; it.own_cnt = rec.w;
    it.nlev = rec2.x; it.a = rec2.y & 0xFFFF;
    it.d = rec2.y >> 16;
    it.lo = rec2.z;
    it.hi = rec2.w;
    return true;
}
// the patch's entry table in LDS behind `planes` tile planes of CAP + 1 slots: every thread parks the entry it owns
// in slots only it ever reads (a register file extension: no barrier, no bank conflict; compile-time offsets)
template <int CAP>
struct EntryTable {
    double *w1, *w2, *r1, *r2;
    int *pos, *u1, *u2;
    uint32_t *vis, *loc;
    __device__ __forceinline__ EntryTable(double2 *tiles, int planes)
    {
        w1 = reinterpret_cast<double *>(tiles + planes * (CAP + 1));
        w2 = w1 + CAP; r1 = w2 + CAP; r2 = r1 + CAP;
        pos = reinterpret_cast<int *>(r2 + CAP);
        u1 = pos + CAP; u2 = u1 + CAP;
        vis = reinterpret_cast<uint32_t *>(u2 + CAP);
        loc = vis + CAP;
    }
    template <typename Tables>       // PatchArgs, or any record with its e_* pointers
    __device__ __forceinline__ void park(const Tables &pa, const PatchItem &it, int tid) const
    {
        const bool ok = tid < it.n_ent;
        const int e = it.ent_off + (ok ? tid : it.n_ent - 1);
        pos[tid] = pa.e_pos[e];
        u1[tid] = pa.e_u1[e];
        u2[tid] = pa.e_u2[e];
        vis[tid] = ok ? pa.e_vis[e] : 0u;
        // an upwind outside the cone reads the zero slot (coupling x a finite 0)
        const uint32_t lc = pa.e_loc[e], l1 = lc & 0xFFFFu, l2 = lc >> 16;
        loc[tid] = (l1 == 0xFFFFu ? (uint32_t)it.n_ent : l1) | ((l2 == 0xFFFFu ? (uint32_t)it.n_ent : l2) << 16);
        w1[tid] = pa.e_w1[e]; w2[tid] = pa.e_w2[e]; r1[tid] = pa.e_r1[e]; r2[tid] = pa.e_r2[e];
    }
};

// what the pair loop of an item reads besides its entry table: the planes of the item's direction / angle
struct PairIO {
    int n;                     // sites (< 2^28 on this path: planes are addressed with 32-bit byte offsets)
    int npair, lgB, dbg;
    const void *S;             // S plane set of the item's direction
    const void *alpha;         // alpha[d] (one opacity per site, or per site and wavelength) or alpha_angle (per angle)
    void *I;
};
template <int AM>
__device__ __forceinline__ PairIO pair_io(const PatchArgs &pa, int d)
{
    PairIO io;
    io.n = (int)pa.ta.n; io.npair = pa.npair; io.lgB = pa.lgB; io.dbg = kDiag ? pa.dbg : 0;
    io.S = pa.ta.S[d];
    io.alpha = AM == VRT_ALPHA_ANGLE_SITE_LAM ? (const void *)pa.ta.alpha_angle : (const void *)pa.ta.alpha[d];
    io.I = pa.ta.I;
    return io;
}

// ---- chained launch (k_patch_chain): hand-off of final intensities between workgroups INSIDE one launch ----------------
// The reference's loop nest is `for layer ... for sweep ... for site` (irregular_ray_tracing.jl:37-80): a layer reads the
// final intensities of earlier layers.  With one launch per layer that order costs a kernel boundary per layer and a
// chip that idles while a layer's slowest workgroups finish.  The chained form keeps the SAME items (patch, angle, block
// of wavelength pairs) and the same arithmetic, but one persistent launch takes them from per-XCD queues in layer
// order, and the only ordering left is the data's own: patch q waits for the patches that store the earlier-layer
// upwind intensities it gathers (vrt_patch.cpp: dep_list), wavelength pair by wavelength pair.
//   producer: the intensities of a pair leave with write-through (sc1) stores; every storing wave drains its stores
//     (s_waitcnt vmcnt(0)), the workgroup passes a barrier, ONE lane then stores the item's progress word = epoch base +
//     pairs finished (agent-scope store) -- cdna_hip_programming.md Guideline 16, form R1;
//   consumer: every wave polls the progress words of the item's dependencies (relaxed agent-scope loads, one lane per
//     dependency) before it gathers the pair's intensities, and EVERY load of an intensity in this kernel is an sc1
//     buffer load (never served by a CU's L1).  Nothing else is handed from workgroup to workgroup: S, alpha and the
//     tables are read-only during the launch, J_dir is only written.
// A workgroup waits only for items of EARLIER layers, every queue is in layer order and is taken in order, so the
// oldest unfinished item of the launch never waits: no placement or residency assumption is needed for progress
// (blocks x, x + 8, ... taking queue x is speed only: neighbouring patches meet in one L2).  Every spin is bounded
// (ChainDev::spin_limit): on expiry the launch gives up, drains and reports through a host-visible status word.
typedef unsigned int vrt_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int vrt_u32x4 __attribute__((ext_vector_type(4)));
// cache-policy bits of the buffer instructions that move intensities between workgroups: sc0 | sc1 (system scope).
// sc1 alone (agent scope) bypasses the reading CU's L1 but is served by the reading XCD's L2, and a 128-byte line of an
// intensity plane is shared by the patches either side of a patch or layer boundary: a reader that pulled the line
// into its L2 when one half was final sees the OTHER half stale afterwards (observed: three chained launches sharing the
// chip, 1 wrong J in 90 -- MI355X_MICROARCH.md's hand-off table asks for whole lines per store for the same reason).
// System-scope accesses are not served from an XCD's L2 copy.
#ifndef VRT_CHAIN_AUX
#define VRT_CHAIN_AUX 17
#endif
constexpr int kAuxSc1 = VRT_CHAIN_AUX;
constexpr int kChainDepLds = 128;              // dependencies of an item kept in LDS (more: read from the global list)
constexpr int kChainHeadStride = 32;           // 32-bit words between the queue heads (a 128-byte line each)
constexpr int kChainAbortWord = 8 * kChainHeadStride;   // ctrl[] index of the give-up word

// What a workgroup keeps in REGISTERS about the chain: nothing.  Scalar registers are the solver's scarcest resource
// (it needs 72 of the 80 that eight waves per SIMD allow; every further one is spilled into a vector register's lanes,
// and that register is the 64th of the 64 the solver needs).  The item's chain state -- progress words of its split,
// epoch base, its own patch, its dependency list, the launch's argument block -- is parked in LDS behind the entry
// table, and the polls and the publication read it from there into vector registers that are free at that point.
struct ChainDev;
// LDS words of the chained launch behind the entry table; then 64 + kChainDepLds dependency slots (patch indices, padded
// with the item's own patch up to a multiple of 64: every lane of a poll has a word it may load)
constexpr int kCtlItem = 0, kCtlSelf = 1, kCtlNdep = 2, kCtlDepOff = 3, kCtlArgsLo = 4, kCtlArgsHi = 5, kCtlProgLo = 6, kCtlProgHi = 7,
              kCtlBase = 8, kCtlGaveUp = 9, kCtlWords = 12;
template <int CAP>
__device__ __forceinline__ uint32_t *chain_ctl_slots(const EntryTable<CAP> &tab) { return tab.loc + CAP; }
template <int CAP>
__device__ __forceinline__ int32_t *chain_dep_slots(const EntryTable<CAP> &tab) { return reinterpret_cast<int32_t *>(tab.loc + CAP + kCtlWords); }

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t *chain_progress(const uint32_t *ctl)
{
    return reinterpret_cast<uint32_t *>(((uint64_t)ctl[kCtlProgHi] << 32) | (uint64_t)ctl[kCtlProgLo]);
}
// first look at the progress words of the dependencies (lane j: dependency j; lanes beyond the list look at the
// item's own word and count as satisfied), issued ahead of its use.  `step`: pairs the dependencies must have published
__device__ __forceinline__ uint32_t chain_peek(const int32_t *s_dep, int step)
{
    const uint32_t *ctl = reinterpret_cast<const uint32_t *>(s_dep - kCtlWords);
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t v = ld_agent(chain_progress(ctl) + s_dep[lane]);
    const uint32_t target = ctl[kCtlBase] + (uint32_t)step;
    return lane < (int)ctl[kCtlNdep] ? v - target : 0u;       // >= 0 as a signed number: reached (modulo 2^32: earlier launches carry smaller epochs)
}
__device__ __forceinline__ void chain_wait_slow(const int32_t *s_dep, int step);
// returns once every dependency has published `step` pairs; `first`: what chain_peek saw
__device__ __forceinline__ void chain_wait(const int32_t *s_dep, int step, uint32_t first)
{
    const uint32_t *ctl = reinterpret_cast<const uint32_t *>(s_dep - kCtlWords);
    if (!__all((int)first >= 0 && ctl[kCtlNdep] <= 64u)) chain_wait_slow(s_dep, step);
    // (no instruction: keeps the compiler from moving the gathers that follow above the poll)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a pointer the compiler can keep in scalar registers (buffer descriptors must be wave-uniform)
template <typename P>
__device__ __forceinline__ P *uniform_ptr(P *p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (P *)(((uint64_t)hi << 32) | (uint64_t)lo);
}
// agent-scope (sc1) accesses of one plane block through a buffer descriptor: base + 32-bit byte offset
template <typename V> struct BufSc1;
template <> struct BufSc1<double2> {
    static __device__ __forceinline__ double2 load(__amdgpu_buffer_rsrc_t rs, unsigned off)
    {
        const vrt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, kAuxSc1);
        double2 d;
        __builtin_memcpy(&d, &v, 16);
        return d;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, double2 d)
    {
        vrt_u32x4 v;
        __builtin_memcpy(&v, &d, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, kAuxSc1);
    }
};
template <> struct BufSc1<float4> {
    static __device__ __forceinline__ float4 load(__amdgpu_buffer_rsrc_t rs, unsigned off)
    {
        const vrt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, kAuxSc1);
        float4 d;
        __builtin_memcpy(&d, &v, 16);
        return d;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, float4 d)
    {
        vrt_u32x4 v;
        __builtin_memcpy(&v, &d, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, kAuxSc1);
    }
};
template <> struct BufSc1<float2> {
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t rs, unsigned off)
    {
        const vrt_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, kAuxSc1);
        float2 d;
        __builtin_memcpy(&d, &v, 8);
        return d;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned off, float2 d)
    {
        vrt_u32x2 v;
        __builtin_memcpy(&v, &d, 8);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, 0, kAuxSc1);
    }
};
// descriptor of `bytes` bytes at p (raw buffer, 32-bit offsets; cdna_hip_programming.md T8)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(uniform_ptr(p)), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// an item's progress word: pairs (steps of its pair loop) whose intensities are stored and drained
__device__ __forceinline__ void chain_publish(const int32_t *s_dep, int steps_done)
{
    const uint32_t *ctl = reinterpret_cast<const uint32_t *>(s_dep - kCtlWords);
    st_agent(chain_progress(ctl) + ctl[kCtlSelf], ctl[kCtlBase] + (uint32_t)steps_done);
}

// ---- one pair at a time within 64 registers: FOUR workgroups per CU ---------------------------------------------------
// wave priority inside the level loop (s_setprio): the loop is a chain of barriers with a few instructions per wave
// between them, while the compute unit's other workgroups run their arithmetic phases -- served first, a level costs
// its own latency instead of its turn in the queue
#ifndef VRT_LEVEL_PRIO
#define VRT_LEVEL_PRIO 3
#endif

// No unit of the chip is saturated by the patch kernel; its phases (gathers, arithmetic, level loop) overlap only as
// far as three 512-thread workgroups per CU allow (72 registers).  This form fits the 64 of a fourth: the three
// alpha gathers first, the four optical depths from them (the alphas die), then the five S / I gathers in flight
// while the weights are evaluated one after the other, each folded into its share of the visit as soon as it exists.
// ---- the chained launch's second hand-off: the intensity IS the flag (small pair counts) -----------------------------
// Before the launch the intensity planes are filled with kChainSentinel in every 32-bit word (a quiet NaN as a double
// and as a float; the boundary layer and the never-visited site's zero are written behind the fill).  A patch stores
// its intensities write-through as before and publishes nothing; a gather that still holds the pattern is repeated.
// What the progress words cost per layer of the chain -- the drain of the stores, the barrier, the publishing store,
// the consumer's poll and only then its gathers: two round trips through the fabric -- becomes one.  The fill is
// 16 bytes per (site, angle, pair) per step, so this form is chosen for one or two pairs only (VRT_CHAIN_DATAFLAG).
constexpr uint32_t kChainSentinel = 0x7FF87FF8u;
__device__ __forceinline__ bool chain_is_sentinel(double2 v)
{
    return (uint32_t)__double2hiint(v.x) == kChainSentinel || (uint32_t)__double2hiint(v.y) == kChainSentinel;
}
__device__ __forceinline__ bool chain_is_sentinel(float2 v)
{
    return __float_as_uint(v.x) == kChainSentinel || __float_as_uint(v.y) == kChainSentinel;
}
// A wait is bounded by a fixed number of repeats (seconds: nothing but a fault of the launch itself makes an owner never
// store) -- no limit held in a register, no call inside the loop (the solver has neither to spare); past it the item
// is reported through the launch's status words like a give-up of chain_wait_slow.
constexpr uint32_t kChainDataSpins = 1u << 21;
__device__ __forceinline__ void chain_data_give_up_cd(const ChainDev *cd, uint32_t item);
// the abort word of a launch, from the device copy of its argument block (ChainDev: defined below; `ctrl` read through its offset)
__device__ __forceinline__ const uint32_t *chain_abort_word(const uint32_t *cd_words);
template <typename T2> __device__ __forceinline__ T2 chain_plain_nan();
template <> __device__ __forceinline__ double2 chain_plain_nan<double2>()
{
    const double q = __hiloint2double(0x7FF80000, 0);
    return make_double2(q, q);
}
template <> __device__ __forceinline__ float2 chain_plain_nan<float2>()
{
    const float q = __uint_as_float(0x7FC00000u);
    return make_float2(q, q);
}
// Every other repeat asks the reading XCD's own L2 (agent scope: sc1 alone).  A patch's upwind owners are mostly patches of
// the same queue, i.e. the same XCD (the queues cut every layer at the same eighths of the storage order): their
// write-through stores pass through this very L2, and a line a repeat has brought in is updated there -- the value is seen
// after an L2 round trip instead of a trip to memory and back.  A line whose owner ran on another XCD stays stale in this
// L2 (it still shows the fill pattern); the system-scope repeats in between see it.  A value that is not the pattern is
// final whichever way it was read: the planes are filled before the launch, and a launch starts with clean L2s.
template <typename T2>
__device__ __forceinline__ T2 load_agent_scope(__amdgpu_buffer_rsrc_t rs, unsigned off);
template <>
__device__ __forceinline__ double2 load_agent_scope<double2>(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const vrt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
template <>
__device__ __forceinline__ float2 load_agent_scope<float2>(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const vrt_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 16);
    float2 d;
    __builtin_memcpy(&d, &v, 8);
    return d;
}
#ifndef VRT_CHAIN_L2_POLL
#define VRT_CHAIN_L2_POLL 1
#endif
template <typename T2>
__device__ __forceinline__ void chain_data_wait(__amdgpu_buffer_rsrc_t rs, unsigned off1, unsigned off2, T2 &r1, T2 &r2,
                                                const int32_t *s_dep)
{
    uint32_t spins = 0;
    while (__any(chain_is_sentinel(r1) || chain_is_sentinel(r2))) {      // wave-uniform: the lanes' upwinds finish together or nearly
        __builtin_amdgcn_s_sleep(1);
        if (VRT_CHAIN_L2_POLL && !(spins & 1u)) {
            // (only the lanes still waiting: a lane that holds a value keeps it)
            const T2 a1 = load_agent_scope<T2>(rs, off1), a2 = load_agent_scope<T2>(rs, off2);
            if (chain_is_sentinel(r1)) r1 = a1;
            if (chain_is_sentinel(r2)) r2 = a2;
        } else {
            const T2 a1 = BufSc1<T2>::load(rs, off1), a2 = BufSc1<T2>::load(rs, off2);
            if (chain_is_sentinel(r1)) r1 = a1;
            if (chain_is_sentinel(r2)) r2 = a2;
        }
        ++spins;
        bool stop = spins > kChainDataSpins;
        if ((spins & 255u) == 0u) {
            // another workgroup of the launch has given up (the abort word of the launch's control block): this one would
            // otherwise spin to its own limit behind it, layer after layer
            const uint32_t *ctl = reinterpret_cast<const uint32_t *>(s_dep - kCtlWords);
            const uint32_t *cdw = reinterpret_cast<const uint32_t *>(((uint64_t)ctl[kCtlArgsHi] << 32) | (uint64_t)ctl[kCtlArgsLo]);
            stop = stop || ld_agent(chain_abort_word(cdw)) != 0u;
        }
        if (stop) {
            // (noted in the item's LDS words; chain_item reports it where the solver's registers are free again)
            const_cast<uint32_t *>(reinterpret_cast<const uint32_t *>(s_dep - kCtlWords))[kCtlGaveUp] = 1u;
            // a lane that still holds the pattern goes on with an ordinary NaN: what this patch stores is then not the
            // pattern, and the patches behind it pass through instead of waiting for it in their turn
            if (chain_is_sentinel(r1)) r1 = chain_plain_nan<T2>();
            if (chain_is_sentinel(r2)) r2 = chain_plain_nan<T2>();
            break;
        }
    }
}

// lean_pairs: the wavelength pairs [it.b0, it.b1) of one item, the patch's entry table parked in `tab`.  CHAIN: inside
// the chained launch (intensities through sc1 buffer accesses, dependencies polled per pair, progress published).
#ifndef VRT_LEAN_ATTR
#define VRT_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
template <typename T, int AM, int NT, int CHAIN>
__device__ __forceinline__ void lean_pairs(const PairIO &pa, const PatchItem &it, const EntryTable<NT> &tab,
                                           double2 *ptile)
{
    typedef typename Pair<T>::type T2;
    const int tid = threadIdx.x;
    constexpr int lgT2 = Log2Size<T2>::value;
    const unsigned n = (unsigned)pa.n;
    const int dbg = kDiag ? pa.dbg : 0;
    const T2 *Sd = reinterpret_cast<const T2 *>(pa.S);
    const T2 *Ia = reinterpret_cast<const T2 *>(pa.I) + (size_t)it.a * pa.npair * (size_t)n;
    auto at = [](const T2 *base, unsigned off) { return *reinterpret_cast<const T2 *>(reinterpret_cast<const char *>(base) + off); };
    const int32_t *s_dep = chain_dep_slots(tab);
    for (int bk = it.b0; bk < it.b1; bk++) {
        int k0, lw;
        if constexpr (CHAIN) { k0 = bk; lw = 0; }            // the chained launch: one pair per block, no siblings
        else {
            pair_block_of(bk, pa.npair, pa.lgB, k0, lw);
            if (it.sib >= (1 << lw)) break;                  // block widths only shrink
        }
        const size_t qb = (size_t)k0 * (size_t)n + (size_t)(CHAIN ? 0 : it.sib);
        const int sh = lw + lgT2;
        uint32_t seen = 0;
        if constexpr (CHAIN == 1) seen = chain_peek(s_dep, bk - it.b0 + 1);   // in flight beside the alpha gathers
        double2 c, g1, g2;
        {
            const int p = tab.pos[tid];
            int v1 = tab.u1[tid], v2 = tab.u2[tid];
            if (diag(dbg, kDiagCentreGathers)) { v1 = p; v2 = p; }          // diagnostics: gathers -> coalesced centre reads
            // ---- the four optical depths: r (alpha_c + alpha_u) / 2 = (r / 2)(alpha_c + alpha_u) ----------------
            double d1x, d2x, d1y, d2y;
            {
                const int av1 = diag(dbg, kDiagNoGatherAlpha) ? p : v1, av2 = diag(dbg, kDiagNoGatherAlpha) ? p : v2;
                double2 a_c, a_1, a_2;
                if constexpr (AM == VRT_ALPHA_SITE) {
                    const T *__restrict__ A1 = reinterpret_cast<const T *>(pa.alpha);
                    const double c0 = A1[p], c1 = A1[av1], c2 = A1[av2];
                    a_c = make_double2(c0, c0); a_1 = make_double2(c1, c1); a_2 = make_double2(c2, c2);
                } else {
                    const T2 *Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const T2 *>(pa.alpha) + qb
                                                            : reinterpret_cast<const T2 *>(pa.alpha) + (size_t)it.a * pa.npair * (size_t)n + qb;
                    a_c = to_d2(at(Al, (unsigned)p << sh)); a_1 = to_d2(at(Al, (unsigned)av1 << sh)); a_2 = to_d2(at(Al, (unsigned)av2 << sh));
                }
                const double rh1 = 0.5 * tab.r1[tid], rh2 = 0.5 * tab.r2[tid];
                d1x = rh1 * (a_c.x + a_1.x); d2x = rh2 * (a_c.x + a_2.x);
                d1y = rh1 * (a_c.y + a_1.y); d2y = rh2 * (a_c.y + a_2.y);
            }
            asm volatile("" : "+v"(d1x), "+v"(d2x), "+v"(d1y), "+v"(d2y) : : "memory");
            // ---- S and I in flight, the weights one after the other -------------------------------------------------
            // an upwind's intensity counts when it lies in an EARLIER layer (final); otherwise the gather reads the
            // never-visited site at storage position n - 1, whose intensity is 0 in every plane (:23)
            int i1 = v1 < it.lo ? v1 : (int)n - 1, i2 = v2 < it.lo ? v2 : (int)n - 1;
            if (diag(dbg, kDiagNoGatherI)) { i1 = (int)n - 1; i2 = (int)n - 1; }
            const int sv1 = diag(dbg, kDiagNoGatherS) ? p : v1, sv2 = diag(dbg, kDiagNoGatherS) ? p : v2;
            const T2 rS_c = at(Sd + qb, (unsigned)p << sh), rS_1 = at(Sd + qb, (unsigned)sv1 << sh), rS_2 = at(Sd + qb, (unsigned)sv2 << sh);
            T2 rI_1, rI_2;
            if constexpr (CHAIN) {
                // the patches that store these intensities have published this pair (Guideline 16: poll, then sc1 loads)
                if constexpr (CHAIN == 1) {
                    if (!diag(dbg, kDiagNoWait)) chain_wait(s_dep, bk - it.b0 + 1, seen);
                }
                const __amdgpu_buffer_rsrc_t rsI = plane_rsrc(Ia + qb, n << sh);
                if (diag(dbg, kDiagPlainLoadI)) {
                    rI_1 = at(Ia + qb, (unsigned)i1 << sh);
                    rI_2 = at(Ia + qb, (unsigned)i2 << sh);
                } else {
                    rI_1 = BufSc1<T2>::load(rsI, (unsigned)i1 << sh);
                    rI_2 = BufSc1<T2>::load(rsI, (unsigned)i2 << sh);
                }
                // ... or the intensities are their own flags: the planes were filled with a NaN pattern before the launch,
                // a gather that still holds it is repeated (chain_data_wait) -- behind the weights, which do not need them:
                // what a layer of the chain waits for after its intensities arrive is 14 operations, not ~190
                if constexpr (CHAIN == 2) {
                    const bool in1 = (v1 >= it.lo) & (v1 < it.hi), in2 = (v2 >= it.lo) & (v2 < it.hi);
                    if (!diag(dbg, kDiagNoWeights)) {
                        LateTerms Lx, Ly;
                        late_lambda(d1x, d2x, (double)rS_1.x, (double)rS_2.x, Lx, d1y);
                        double sink = 0.0;
                        late_lambda(d1y, d2y, (double)rS_1.y, (double)rS_2.y, Ly, sink);
                        if (!diag(dbg, kDiagNoWait)) chain_data_wait<T2>(rsI, (unsigned)i1 << sh, (unsigned)i2 << sh, rI_1, rI_2, s_dep);
                        late_apply(Lx, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.x, (double)rI_1.x, (double)rI_2.x, c.x, g1.x, g2.x);
                        late_apply(Ly, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.y, (double)rI_1.y, (double)rI_2.y, c.y, g1.y, g2.y);
                    } else {
                        if (!diag(dbg, kDiagNoWait)) chain_data_wait<T2>(rsI, (unsigned)i1 << sh, (unsigned)i2 << sh, rI_1, rI_2, s_dep);
                        c = make_double2(d1x + (double)rS_c.x + (double)rS_1.x + (double)rI_1.x, d2y + (double)rS_c.y + (double)rS_2.y + (double)rI_2.y);
                        g1 = make_double2(in1 ? d2x : 0.0, in1 ? d1y : 0.0);
                        g2 = make_double2(in2 ? d1x : 0.0, in2 ? d2y : 0.0);
                    }
                }
            } else {
                rI_1 = at(Ia + qb, (unsigned)i1 << sh);
                rI_2 = at(Ia + qb, (unsigned)i2 << sh);
            }
            const bool in1 = (v1 >= it.lo) & (v1 < it.hi), in2 = (v2 >= it.lo) & (v2 < it.hi);
            if constexpr (CHAIN == 2) {
                // (done above, around the wait)
            } else if (diag(dbg, kDiagNoWeights)) {                     // diagnostics: no weights arithmetic
                c = make_double2(d1x + (double)rS_c.x + (double)rS_1.x + (double)rI_1.x, d2y + (double)rS_c.y + (double)rS_2.y + (double)rI_2.y);
                g1 = make_double2(in1 ? d2x : 0.0, in1 ? d1y : 0.0);
                g2 = make_double2(in2 ? d1x : 0.0, in2 ? d2y : 0.0);
            } else {
                entry_lambda_seq(d1x, d2x, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.x, (double)rS_1.x, (double)rS_2.x,
                                 (double)rI_1.x, (double)rI_2.x, c.x, g1.x, g2.x, d1y);
                double sink = 0.0;
                entry_lambda_seq(d1y, d2y, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.y, (double)rS_1.y, (double)rS_2.y,
                                 (double)rI_1.y, (double)rI_2.y, c.y, g1.y, g2.y, sink);
            }
        }
        // ---- the patch's Gauss-Seidel levels on the LDS tile ------------------------------------------------------
        uint32_t vis = tab.vis[tid];
        const uint32_t loc = tab.loc[tid];
        {
            double z;
            asm volatile("v_mov_b64 %0, 0" : "=v"(z));
            if (tid < it.n_ent) ptile[tid] = make_double2(z, z);         // I = zero(S), :23
        }
        if constexpr (CHAIN == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous pair's stores have left (its loads are long consumed)
        __syncthreads();
        if constexpr (CHAIN == 1) {
            if (bk > it.b0 && tid == 0) chain_publish(s_dep, bk - it.b0);   // ... in every wave: pairs b0 .. bk-1 are published
        }
        const int nlev = diag(dbg, kDiagNoLevels) ? 0 : it.nlev;
        __builtin_amdgcn_s_setprio(VRT_LEVEL_PRIO);
        for (int t = 1; t <= nlev; t++) {
            if ((vis & 0xFFu) == (uint32_t)t) {                          // a site's visits come at increasing levels
                const double2 xv = ptile[loc & 0xFFFFu], yv = ptile[loc >> 16];
                double2 r;
                r.x = fma(g2.x, yv.x, fma(g1.x, xv.x, c.x));
                r.y = fma(g2.y, yv.y, fma(g1.y, xv.y, c.y));
                ptile[tid] = r;
                vis >>= 8;
            }
            __syncthreads();
        }
        __builtin_amdgcn_s_setprio(0);
        if (tid < it.own_cnt && !(diag(dbg, kDiagNoStores) && ptile[tid].x != 1.2345e300)) {
            if (CHAIN && !diag(dbg, kDiagPlainStoreI)) {
                const __amdgpu_buffer_rsrc_t rsI = plane_rsrc(Ia + qb, n << sh);
                BufSc1<T2>::store(rsI, (unsigned)(it.own_lo + tid) << sh, from_d2<T>(ptile[tid]));
            } else {
                T2 *I = reinterpret_cast<T2 *>(pa.I) + (size_t)it.a * pa.npair * (size_t)n + qb;
                *reinterpret_cast<T2 *>(reinterpret_cast<char *>(I) + ((unsigned)(it.own_lo + tid) << sh)) = from_d2<T>(ptile[tid]);
            }
        }
        __syncthreads();                                                 // the tile is rewritten by the next pair
    }
}

template <typename T, int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_LEAN_ATTR
k_patch_lean(PatchArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const int tid = threadIdx.x;
    if (reduce_block_index(pa) >= 0) {
        patch_reduce_role<T, NT>(pa);
        return;
    }
    PatchItem it;
    if (!patch_item(pa, pa.lgB, it)) return;
    if (diag(pa.dbg, kDiagNoPairLoop)) it.b1 = it.b0;           // diagnostics: the item's overhead alone
    const EntryTable<NT> tab(ptile, 1);
    tab.park(pa, it, tid);
    if (tid == 0) ptile[it.n_ent] = make_double2(0.0, 0.0);  // the zero slot
    exp2_table_fill();
    __syncthreads();
    lean_pairs<T, AM, NT, 0>(pair_io<AM>(pa, it.d), it, tab, ptile);
}


// ---- fp32 storage, FOUR wavelengths per lane ---------------------------------------------------------------------
// With float values a wavelength pair is an 8-byte access, and the patch kernel issues as many memory instructions
// per wavelength as with doubles: the memory path, which bounds it, sees twice the requests
// per byte.  In the layout with two (or more) pairs of a site side by side (pair blocks, vrt_device.h) two
// neighbouring pairs are ONE 16-byte access: this form solves both at once -- eight float4 gathers per entry (the
// three alphas first, then S and I under the weights, as in lean_pairs), the four evaluations of the weights one
// after the other (compiler fences), two planes of the LDS tile walked by one level loop, one float4 store.  Half
// the memory instructions and half the barriers per wavelength.  (An odd pair count leaves a last block of one pair: the host then launches the pair kernel.)
// quad_pairs: the pairs 2 sib2, 2 sib2 + 1 of every block among blocks [it.b0, it.b1) of one item (it.sib = sib2).
template <int AM, int NT, bool CHAIN>
__device__ __forceinline__ void quad_pairs(const PairIO &pa, const PatchItem &it, const EntryTable<NT> &tab,
                                           double2 *ptile)
{
    const int tid = threadIdx.x;
    const int sib2 = it.sib, b0 = it.b0, b1 = it.b1;
    const unsigned n = (unsigned)pa.n;
    const int n_ent = it.n_ent, own_lo = it.own_lo, own_cnt = it.own_cnt, nlev = it.nlev, a = it.a;
    const int lo = it.lo, hi = it.hi;
    constexpr int CAP = NT;
    double2 *tileA = ptile, *tileB = ptile + (CAP + 1);
    double *const s_r1 = tab.r1, *const s_r2 = tab.r2, *const s_w1 = tab.w1, *const s_w2 = tab.w2;
    int *const s_pos = tab.pos, *const s_u1 = tab.u1, *const s_u2 = tab.u2;
    uint32_t *const s_vis = tab.vis, *const s_loc = tab.loc;
    const float2 *Sd = reinterpret_cast<const float2 *>(pa.S);
    const float2 *Ia = reinterpret_cast<const float2 *>(pa.I) + (size_t)a * pa.npair * (size_t)n;
    auto at4 = [](const float2 *base, unsigned off) { return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off); };
    const int32_t *s_dep = chain_dep_slots(tab);
    for (int bk = b0; bk < b1; bk++) {
        int k0, lw;
        if constexpr (CHAIN) { k0 = 2 * bk; lw = 1; }        // the chained launch: blocks of two pairs, no siblings
        else {
            pair_block_of(bk, pa.npair, pa.lgB, k0, lw);
            if (2 * sib2 >= (1 << lw)) break;                // block widths only shrink (and are >= 2: even pair count)
        }
        const size_t qb = (size_t)k0 * (size_t)n + (size_t)(CHAIN ? 0 : 2 * sib2);
        const int sh = lw + 3;                               // log2 bytes per site of the block
        uint32_t seen = 0;
        if constexpr (CHAIN) seen = chain_peek(s_dep, bk - b0 + 1);
        const int p = s_pos[tid], v1 = s_u1[tid], v2 = s_u2[tid];
        // ---- the eight optical depths from the three alpha gathers (two gather phases as in lean_pairs: the
        // alphas are dead before the S / I gathers land) ------------------------------------------------------------
        double d1x, d2x, d1y, d2y, d1z, d2z, d1w, d2w;
        {
            float4 a_c, a_1, a_2;
            if constexpr (AM == VRT_ALPHA_SITE) {
                const float *__restrict__ A1 = reinterpret_cast<const float *>(pa.alpha);
                const float c0 = A1[p], c1 = A1[v1], c2 = A1[v2];
                a_c = make_float4(c0, c0, c0, c0); a_1 = make_float4(c1, c1, c1, c1); a_2 = make_float4(c2, c2, c2, c2);
            } else {
                const float2 *Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const float2 *>(pa.alpha) + qb
                                                            : reinterpret_cast<const float2 *>(pa.alpha) + (size_t)a * pa.npair * (size_t)n + qb;
                a_c = at4(Al, (unsigned)p << sh); a_1 = at4(Al, (unsigned)v1 << sh); a_2 = at4(Al, (unsigned)v2 << sh);
            }
            const double rh1 = 0.5 * s_r1[tid], rh2 = 0.5 * s_r2[tid];
            d1x = rh1 * ((double)a_c.x + (double)a_1.x); d2x = rh2 * ((double)a_c.x + (double)a_2.x);
            d1y = rh1 * ((double)a_c.y + (double)a_1.y); d2y = rh2 * ((double)a_c.y + (double)a_2.y);
            d1z = rh1 * ((double)a_c.z + (double)a_1.z); d2z = rh2 * ((double)a_c.z + (double)a_2.z);
            d1w = rh1 * ((double)a_c.w + (double)a_1.w); d2w = rh2 * ((double)a_c.w + (double)a_2.w);
        }
        asm volatile("" : "+v"(d1x), "+v"(d2x), "+v"(d1y), "+v"(d2y), "+v"(d1z), "+v"(d2z), "+v"(d1w), "+v"(d2w) : : "memory");
        // ---- S and I in flight, the weights of the four wavelengths one after the other --------------------------
        double2 cA, g1A, g2A, cB, g1B, g2B;
        {
            const int i1 = v1 < lo ? v1 : (int)n - 1, i2 = v2 < lo ? v2 : (int)n - 1;
            const bool in1 = (v1 >= lo) & (v1 < hi), in2 = (v2 >= lo) & (v2 < hi);
            const float4 S_c = at4(Sd + qb, (unsigned)p << sh), S_1 = at4(Sd + qb, (unsigned)v1 << sh), S_2 = at4(Sd + qb, (unsigned)v2 << sh);
            float4 I_1, I_2;
            if constexpr (CHAIN) {
                chain_wait(s_dep, bk - b0 + 1, seen);
                const __amdgpu_buffer_rsrc_t rsI = plane_rsrc(Ia + qb, n << sh);
                I_1 = BufSc1<float4>::load(rsI, (unsigned)i1 << sh);
                I_2 = BufSc1<float4>::load(rsI, (unsigned)i2 << sh);
            } else {
                I_1 = at4(Ia + qb, (unsigned)i1 << sh);
                I_2 = at4(Ia + qb, (unsigned)i2 << sh);
            }
            entry_lambda_seq(d1x, d2x, s_w1 + tid, s_w2 + tid, in1, in2, (double)S_c.x, (double)S_1.x, (double)S_2.x,
                             (double)I_1.x, (double)I_2.x, cA.x, g1A.x, g2A.x, d1y);
            entry_lambda_seq(d1y, d2y, s_w1 + tid, s_w2 + tid, in1, in2, (double)S_c.y, (double)S_1.y, (double)S_2.y,
                             (double)I_1.y, (double)I_2.y, cA.y, g1A.y, g2A.y, d1z);
            entry_lambda_seq(d1z, d2z, s_w1 + tid, s_w2 + tid, in1, in2, (double)S_c.z, (double)S_1.z, (double)S_2.z,
                             (double)I_1.z, (double)I_2.z, cB.x, g1B.x, g2B.x, d1w);
            double sink = 0.0;
            entry_lambda_seq(d1w, d2w, s_w1 + tid, s_w2 + tid, in1, in2, (double)S_c.w, (double)S_1.w, (double)S_2.w,
                             (double)I_1.w, (double)I_2.w, cB.y, g1B.y, g2B.y, sink);
        }
        // ---- the patch's Gauss-Seidel levels, both pairs per level -------------------------------------------------
        uint32_t vis = s_vis[tid];
        const uint32_t loc = s_loc[tid];
        {
            double z;
            asm volatile("v_mov_b64 %0, 0" : "=v"(z));
            if (tid < n_ent) {
                tileA[tid] = make_double2(z, z);             // I = zero(S), :23
                tileB[tid] = make_double2(z, z);
            }
        }
        if constexpr (CHAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (CHAIN) {
            if (bk > b0 && tid == 0) chain_publish(s_dep, bk - b0);
        }
        for (int t = 1; t <= nlev; t++) {
            if ((vis & 0xFFu) == (uint32_t)t) {
                const uint32_t l1 = loc & 0xFFFFu, l2 = loc >> 16;
                const double2 xa = tileA[l1], ya = tileA[l2], xb = tileB[l1], yb = tileB[l2];
                double2 ra, rb;
                ra.x = fma(g2A.x, ya.x, fma(g1A.x, xa.x, cA.x));
                ra.y = fma(g2A.y, ya.y, fma(g1A.y, xa.y, cA.y));
                rb.x = fma(g2B.x, yb.x, fma(g1B.x, xb.x, cB.x));
                rb.y = fma(g2B.y, yb.y, fma(g1B.y, xb.y, cB.y));
                tileA[tid] = ra;
                tileB[tid] = rb;
                vis >>= 8;
            }
            __syncthreads();
        }
        // ---- final intensities of the owned sites ---------------------------------------------------------------
        if (tid < own_cnt) {
            const double2 ra = tileA[tid], rb = tileB[tid];
            const float4 out = make_float4((float)ra.x, (float)ra.y, (float)rb.x, (float)rb.y);
            if constexpr (CHAIN) {
                const __amdgpu_buffer_rsrc_t rsI = plane_rsrc(Ia + qb, n << sh);
                BufSc1<float4>::store(rsI, (unsigned)(own_lo + tid) << sh, out);
            } else {
                float2 *I = reinterpret_cast<float2 *>(pa.I) + (size_t)a * pa.npair * (size_t)n + qb;
                *reinterpret_cast<float4 *>(reinterpret_cast<char *>(I) + ((unsigned)(own_lo + tid) << sh)) = out;
            }
        }
        __syncthreads();                                     // the tiles are rewritten by the next block
    }
}

#ifndef VRT_QUAD_ATTR            // 80 VGPRs = three 512-thread workgroups per CU (no scratch at that budget)
#define VRT_QUAD_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
template <int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_QUAD_ATTR
k_patch_quad(PatchArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const int tid = threadIdx.x;
    if (reduce_block_index(pa) >= 0) {
        patch_reduce_role<float, NT>(pa);
        return;
    }
    // workgroup (item, sib2, split): the pairs 2 sib2, 2 sib2 + 1 of every block among blocks b0 .. b1-1
    PatchItem it;
    if (!patch_item(pa, pa.lgB - 1, it)) return;
    const EntryTable<NT> tab(ptile, 2);
    tab.park(pa, it, tid);
    if (tid == 0) {
        ptile[it.n_ent] = make_double2(0.0, 0.0);            // the zero slots
        ptile[NT + 1 + it.n_ent] = make_double2(0.0, 0.0);
    }
    exp2_table_fill();
    __syncthreads();
    quad_pairs<AM, NT, false>(pair_io<AM>(pa, it.d), it, tab, ptile);
}


// ---- the chained launch --------------------------------------------------------------------------------------------
// Items (three int4 each), per XCD queue in layer order:
//   solve:   A = (first entry, first owned position, entries | owned << 10 | levels << 20, angle | dir << 6 | split << 7 | layer << 16)
//            B = (first dependency, dependencies, patch, 0)     C = (lo, hi: storage positions of the patch's layer)
//   reduce:  A = (lo, hi, 1 << 31, dir << 6 | split << 7): J_dir of storage positions [lo, hi), pairs of the split
//            B = (first dependency, dependencies, 0, steps the dependencies must have published)
//   padding: A = (0, 0, 1 << 30, 0): nothing (the queues advance through the layers in step, see ensure_patch_chain)
// The launch's arguments (ChainDev) are kernel arguments; a copy in device memory serves the give-up path, which
// reaches it through a pointer parked in LDS (nothing of the chain lives in registers across the solver).
struct ChainDev {
    TileArgs ta;
    int npair, nsplit, cap;
    const int32_t *e_pos, *e_u1, *e_u2;
    const uint32_t *e_vis, *e_loc;
    const double *e_w1, *e_w2, *e_r1, *e_r2;
    PatchReduce red;           // weights, angle lists ([0] up, [1] down), J_dir planes
    const int4 *items;
    int q_off[9];              // items of queue x: [q_off[x], q_off[x + 1])
    const int32_t *deps;       // dependency lists: patch indices
    uint32_t *progress;        // [nsplit][n_patches]
    uint32_t *ctrl;            // queue heads (kChainHeadStride words apart), give-up word
    uint32_t *host_status;     // mapped host word: non-zero once a spin has expired
    int64_t n_patches;
    uint32_t spin_limit;
    int static_items;          // block b takes item b >> 3 of queue b & 7 (no ticket)
    int dbg;                   // timing diagnostics (-DVRT_DIAG build only, WRONG results): the flags of PatchArgs::dbg, and
                               //   256 no waiting for dependencies, 512 plain (L1-cached) intensity gathers, 1024 plain intensity stores, 2048 no pair loop
};

__device__ __forceinline__ const uint32_t *chain_abort_word(const uint32_t *cd_words)
{
    return reinterpret_cast<const ChainDev *>(cd_words)->ctrl + kChainAbortWord;
}

// uniform (scalar-register) reads of the argument block
__device__ __forceinline__ int cd_int(const int *p) { return __builtin_amdgcn_readfirstlane(*p); }
template <typename P>
__device__ __forceinline__ P *cd_ptr(P *const *p) { return uniform_ptr(*p); }

__device__ __forceinline__ void chain_wait_slow(const int32_t *s_dep, int step)
{
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t *ctl = reinterpret_cast<const uint32_t *>(s_dep - kCtlWords);
    const ChainDev *cd = reinterpret_cast<const ChainDev *>(((uint64_t)ctl[kCtlArgsHi] << 32) | (uint64_t)ctl[kCtlArgsLo]);
    uint32_t *abort_word = cd->ctrl + kChainAbortWord;
    const uint32_t *progress = chain_progress(ctl);
    const uint32_t target = ctl[kCtlBase] + (uint32_t)step, limit = cd->spin_limit;
    const int ndep = (int)ctl[kCtlNdep];
    for (int c0 = 0; c0 < ndep; c0 += 64) {
        const int j = c0 + lane;
        // (the LDS slots are padded with the item's own patch; past them the global list is read where it exists)
        const int dep = j < 64 + kChainDepLds ? s_dep[j] : (j < ndep ? (cd->deps + ctl[kCtlDepOff])[j] : (int)ctl[kCtlSelf]);
        const uint32_t *w = progress + dep;
        uint32_t spins = 0;
        for (;;) {
            const uint32_t v = ld_agent(w);
            if (__all(j >= ndep || (int)(v - target) >= 0)) break;
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 127u) == 1u) {
                if (spins > limit) {                           // give up: the launch drains, the host reports it
                    st_agent(abort_word, 1u);
                    uint32_t *hs = cd->host_status;            // (the item that waited; kept short: this path shares the solver's registers)
                    __hip_atomic_store(hs + 1, ctl[kCtlItem], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(hs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (ld_agent(abort_word) != 0u) return;
            }
        }
    }
}

__device__ __forceinline__ void chain_data_give_up_cd(const ChainDev *cd, uint32_t item)
{
    st_agent(cd->ctrl + kChainAbortWord, 1u);                  // the host reports it (patch_chain_check)
    uint32_t *hs = cd->host_status;
    __hip_atomic_store(hs + 1, item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(hs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// J_dir over storage positions [lo, hi) of direction r for the pair blocks [b0, b1): the reduction of
// patch_reduce_role, every intensity read by an sc1 load
template <typename T, int NT, int LGB, int MODE>
__device__ __forceinline__ void chain_reduce(const ChainDev &cd, const ChainDev *cdp, uint32_t item, int r, int lo, int hi, int b0, int b1)
{
    typedef typename Pair<T>::type T2;
    const int tid = threadIdx.x;
    if (diag(cd.dbg, kDiagNoReduce)) return;
    const int64_t nn = cd.ta.n;
    const int npair = cd.npair;
    const size_t plane = (size_t)npair * (size_t)nn;
    T2 *Jd = reinterpret_cast<T2 *>(cd.red.Jd[r]);
    const T2 *I0 = reinterpret_cast<const T2 *>(cd.ta.I);
    const int count = cd.red.count[r];
    constexpr int lgT2 = Log2Size<T2>::value;
    for (int bk = b0; bk < b1; bk++) {
        const int k0 = bk << LGB;
        const size_t run = (size_t)(hi - lo) << LGB;
        const size_t base = (size_t)k0 * (size_t)nn + ((size_t)lo << LGB);
        for (size_t f = (size_t)tid; f < run; f += NT) {
            double ax = 0.0, ay = 0.0;
            for (int j = 0; j < count; j++) {                // the reference's angle order (lambda_iteration.jl:84,102,107)
                const int a = cd.red.angles[r][j];
                const double wa = cd.red.w[a];
                const __amdgpu_buffer_rsrc_t rs = plane_rsrc(I0 + (size_t)a * plane + base, (unsigned)(run << lgT2));
                T2 raw = BufSc1<T2>::load(rs, (unsigned)(f << lgT2));
                if constexpr (MODE == 2) {                    // the value is its own flag: repeat while it holds the fill pattern
                    uint32_t spins = 0;
                    while (chain_is_sentinel(raw) && !diag(cd.dbg, kDiagNoWait)) {
                        __builtin_amdgcn_s_sleep(2);
                        raw = BufSc1<T2>::load(rs, (unsigned)(f << lgT2));
                        if (++spins > kChainDataSpins) {
                            chain_data_give_up_cd(cdp, item);
                            break;
                        }
                    }
                }
                const double2 v = to_d2(raw);
                ax += wa * v.x;
                ay += wa * v.y;
            }
            Jd[base + f] = from_d2<T>(make_double2(ax, ay));
        }
    }
}

// ONE item per workgroup: the workgroup takes the next ticket of its queue when it starts and ends with its item --
// the hardware's workgroup dispatcher is the loop (a persistent loop around the solver would have to keep its state
// in scalar registers the solver needs: 28 of them spilled, and with them a vector register of the 64).  Tickets are
// taken in order by workgroups that are running, so the argument about progress above holds whatever order the
// blocks of the grid start in; the grid has exactly one block per item.
template <typename T, int AM, int NT, bool QUAD, int MODE>
__device__ __forceinline__ void chain_item(const ChainDev &ca, const ChainDev *cd, uint32_t base, double2 *ptile)
{
    static_assert(MODE == 1 || (MODE == 2 && !QUAD), "the data-as-flag form exists for the pair kernel");
    const int tid = threadIdx.x;
    constexpr int PLANES = QUAD ? 2 : 1;
    constexpr int LGB = QUAD ? 1 : 0;
    const EntryTable<NT> tab(ptile, PLANES);
    uint32_t *s_ctl = chain_ctl_slots(tab);
    int32_t *s_dep = chain_dep_slots(tab);
    exp2_table_fill();
    if (tid == 0) {
        // next item of this block's queue, in order (blocks x, x + 8, ... share an XCD: speed only); an exhausted
        // queue -> the next one (load balance at the end)
        uint32_t *ctrl = ca.ctrl;
        int q = (int)(blockIdx.x & 7u), idx = -1;
        // Progress-word launches: block b IS item b >> 3 of queue b & 7 -- the queues are padded to one length, the hardware
        // starts the blocks of an XCD in order, and a workgroup waits only for items of earlier layers, i.e. for blocks that
        // were started before it: the ticket's round trip through the L2 (1 of the ~5 us an item spends before its first
        // gather) buys nothing there (1 M sites x 7 wavelengths: 1.86 -> 1.75 ms).  The data-as-flag launches of one or two
        // pairs keep the tickets: an XCD's freed slot then takes the OLDEST item left, whichever queue it is in (C2: 0.431
        // against 0.437 ms).
        if (ca.static_items) {
            const int o0 = ca.q_off[q], len = ca.q_off[q + 1] - o0, t = (int)(blockIdx.x >> 3);
            if (t < len) idx = o0 + t;
        } else
        for (int tries = 0; tries < 8; tries++) {
            const int o0 = ca.q_off[q], len = ca.q_off[q + 1] - o0;
            const uint32_t t = len > 0 ? __hip_atomic_fetch_add(ctrl + q * kChainHeadStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0u;
            if (len > 0 && t < (uint32_t)len) { idx = o0 + (int)t; break; }
            q = (q + 1) & 7;
        }
        s_ctl[kCtlItem] = (uint32_t)idx;
        s_ctl[kCtlGaveUp] = 0u;
        s_ctl[kCtlArgsLo] = (uint32_t)(uint64_t)cd;
        s_ctl[kCtlArgsHi] = (uint32_t)((uint64_t)cd >> 32);
    }
    __syncthreads();
    const int idx = __builtin_amdgcn_readfirstlane((int)s_ctl[kCtlItem]);
    if (idx < 0) return;
    const int4 *items = ca.items;
    const int4 A4 = items[3 * (size_t)idx], B4 = items[3 * (size_t)idx + 1], C4 = items[3 * (size_t)idx + 2];
    const int Ax = __builtin_amdgcn_readfirstlane(A4.x), Ay = __builtin_amdgcn_readfirstlane(A4.y);
    const uint32_t Az = (uint32_t)__builtin_amdgcn_readfirstlane(A4.z), Aw = (uint32_t)__builtin_amdgcn_readfirstlane(A4.w);
    const int Bx = __builtin_amdgcn_readfirstlane(B4.x), By = __builtin_amdgcn_readfirstlane(B4.y);
    const int Bz = __builtin_amdgcn_readfirstlane(B4.z), Bw = __builtin_amdgcn_readfirstlane(B4.w);
    if (Az & 0x40000000u) return;                            // padding: every queue holds as many items of a layer as the longest
    if constexpr (MODE == 1) {
        // dependency slots: the list (its first kChainDepLds entries), padded with the item's own patch to a multiple of 64
        const int32_t *deps = ca.deps;
        const int padded = min(kChainDepLds + 64, (By + 63) / 64 * 64 + (By == 0 ? 64 : 0));
        for (int j = tid; j < padded; j += NT) s_dep[j] = j < By ? deps[Bx + j] : Bz;
    }
    const int d = (int)((Aw >> 6) & 1u), split = (int)((Aw >> 7) & 0x1FFu);
    const int npair = ca.npair;
    int b0, b1;
    {
        // blocks [b0, b1) of the split: dealt evenly as split_blocks does (Q = 1)
        const int nblock = npair >> LGB, nsplit = ca.nsplit;
        const int per = nblock / nsplit, rem = nblock - per * nsplit;
        b0 = split * per + (split < rem ? split : rem);
        b1 = b0 + per + (split < rem ? 1 : 0);
    }
    if (tid == 0) {                                          // (read behind the barrier that publishes the dependency list)
        const uint64_t prog = (uint64_t)(ca.progress + (size_t)split * (size_t)ca.n_patches);
        s_ctl[kCtlSelf] = (uint32_t)Bz;
        s_ctl[kCtlNdep] = (uint32_t)By;
        s_ctl[kCtlDepOff] = (uint32_t)Bx;
        s_ctl[kCtlProgLo] = (uint32_t)prog;
        s_ctl[kCtlProgHi] = (uint32_t)(prog >> 32);
        s_ctl[kCtlBase] = base;
    }
    if (Az >> 31) {                                          // ---- J_dir of a finished range ----------------------------
        if constexpr (MODE == 1) {
            __syncthreads();                                 // the dependency list is in LDS
            chain_wait(s_dep, Bw, chain_peek(s_dep, Bw));
        }
        chain_reduce<T, NT, LGB, MODE>(ca, cd, (uint32_t)idx, d, Ax, Ay, b0, b1);
        return;
    }
    PatchItem it;
    it.sib = 0; it.b0 = b0; it.b1 = b1;
    it.ent_off = Ax; it.own_lo = Ay;
    it.n_ent = (int)(Az & 0x3FFu); it.own_cnt = (int)((Az >> 10) & 0x3FFu); it.nlev = (int)((Az >> 20) & 0xFFu);
    it.a = (int)(Aw & 63u); it.d = d;
    it.lo = __builtin_amdgcn_readfirstlane(C4.x);
    it.hi = __builtin_amdgcn_readfirstlane(C4.y);
    PairIO pa;
    pa.n = (int)ca.ta.n;                                     // n < 2^28
    pa.npair = npair;
    pa.lgB = LGB;
    pa.dbg = kDiag ? ca.dbg : 0;
    pa.S = ca.ta.S[d];
    pa.I = ca.ta.I;
    pa.alpha = AM == VRT_ALPHA_ANGLE_SITE_LAM ? (const void *)ca.ta.alpha_angle : (const void *)ca.ta.alpha[d];
    tab.park(ca, it, tid);
    if (tid == 0) {
        ptile[it.n_ent] = make_double2(0.0, 0.0);            // the zero slot(s)
        if (QUAD) ptile[NT + 1 + it.n_ent] = make_double2(0.0, 0.0);
    }
    __syncthreads();                                         // the dependency list is in LDS
    if (diag(ca.dbg, kDiagNoPairLoop)) it.b1 = it.b0;             // diagnostics: the item's overhead alone
    if constexpr (QUAD) quad_pairs<AM, NT, true>(pa, it, tab, ptile);
    else lean_pairs<T, AM, NT, MODE>(pa, it, tab, ptile);
    if constexpr (MODE == 2) {
        __syncthreads();
        if (tid == 0 && s_ctl[kCtlGaveUp]) chain_data_give_up_cd(cd, (uint32_t)idx);
    }
    if constexpr (MODE == 1) {
        // the last pair: every storing wave drains, then ONE lane publishes the finished item
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) chain_publish(s_dep, b1 - b0);
    }
}

template <typename T, int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_LEAN_ATTR
k_patch_chain(ChainDev ca, const ChainDev *cd, uint32_t base)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    chain_item<T, AM, NT, false, 1>(ca, cd, base, ptile);
}
// the intensities are their own flags (chain_data_wait): the planes were filled with kChainSentinel before the launch.
// (Six waves per SIMD: the repeat loop holds the gathered pair and its offsets beside everything the solver has live at
// that point -- 12 B of scratch at 64 registers -- and this form runs where the chip is far from full anyway.)
template <typename T, int AM, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_patch_chain_df(ChainDev ca, const ChainDev *cd, uint32_t base)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    chain_item<T, AM, NT, false, 2>(ca, cd, base, ptile);
}
template <int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_QUAD_ATTR
k_patch_chain_quad(ChainDev ca, const ChainDev *cd, uint32_t base)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    chain_item<float, AM, NT, true, 1>(ca, cd, base, ptile);
}

// the instantiated launch shapes (entries per thread, pairs at a time, threads)
#define VRT_PATCH_SHAPES(X) \
    X(1, 1, 256) X(1, 1, 512) X(1, 1, 1024) X(2, 1, 256) X(2, 1, 512) X(1, 2, 256) X(1, 2, 512) X(1, 2, 1024) X(2, 2, 512)

template <typename T, int AM>
static int launch_shape(int K, int Q, int NT, dim3 grid, size_t lds, hipStream_t st, const PatchArgs &pa)
{
    if constexpr (sizeof(T) == 4) {
        if (pa.quad) {
            switch (NT) {
            case 256: hipLaunchKernelGGL((k_patch_quad<AM, 256>), grid, dim3(256), lds, st, pa); return VRT_OK;
            case 512: hipLaunchKernelGGL((k_patch_quad<AM, 512>), grid, dim3(512), lds, st, pa); return VRT_OK;
            case 1024: hipLaunchKernelGGL((k_patch_quad<AM, 1024>), grid, dim3(1024), lds, st, pa); return VRT_OK;
            default: break;
            }
        }
    }
    if (pa.lean && K == 1 && Q == 1) {
        switch (NT) {
        case 256: hipLaunchKernelGGL((k_patch_lean<T, AM, 256>), grid, dim3(256), lds, st, pa); return VRT_OK;
        case 512: hipLaunchKernelGGL((k_patch_lean<T, AM, 512>), grid, dim3(512), lds, st, pa); return VRT_OK;
        case 1024: hipLaunchKernelGGL((k_patch_lean<T, AM, 1024>), grid, dim3(1024), lds, st, pa); return VRT_OK;
        default: break;
        }
    }
#define VRT_PATCH_CASE(k, q, nt) \
    if (K == k && Q == q && NT == nt) { hipLaunchKernelGGL((k_patch_solve<T, AM, k, q, nt>), grid, dim3(nt), lds, st, pa); return VRT_OK; }
    VRT_PATCH_SHAPES(VRT_PATCH_CASE)
#undef VRT_PATCH_CASE
    return fail(VRT_EINVAL, "no patch kernel for this (entries per thread, pairs, threads) shape");
}

template <typename T>
static int launch_mode(int am, int K, int Q, int NT, dim3 grid, size_t lds, hipStream_t st, const PatchArgs &pa)
{
    switch (am) {
    case VRT_ALPHA_SITE: return launch_shape<T, VRT_ALPHA_SITE>(K, Q, NT, grid, lds, st, pa);
    case VRT_ALPHA_SITE_LAM: return launch_shape<T, VRT_ALPHA_SITE_LAM>(K, Q, NT, grid, lds, st, pa);
    default: return launch_shape<T, VRT_ALPHA_ANGLE_SITE_LAM>(K, Q, NT, grid, lds, st, pa);
    }
}

bool patch_shape_exists(int K, int Q, int NT)
{
#define VRT_PATCH_CASE(k, q, nt) if (K == k && Q == q && NT == nt) return true;
    VRT_PATCH_SHAPES(VRT_PATCH_CASE)
#undef VRT_PATCH_CASE
    return false;
}

// work lists of the launches: per (stream group, layer) the patches of the group's angles, sorted by
// (first owned position, angle) and cut into 8 runs of equal count; XCD x (blocks x, x + 8, ...)
// walks run x, so that neighbouring patches -- and the angles of one patch, which read the same S
// lines -- meet in one L2
int ensure_patch_work(vrt_plan *p, int G, const std::vector<int32_t> &group_angles, const std::vector<int> &group_off)
{
    if (p->d_patch_work && p->patch_work_groups == G) return VRT_OK;
    if (p->d_patch_work) { (void)hipFree(p->d_patch_work); p->d_patch_work = nullptr; }
    const int maxL = p->tile_max_layers;
    std::vector<int4> work;           // two per slot (PatchArgs::wrec)
    p->patch_work_off.assign((size_t)G * (size_t)(maxL + 2) + 1, 0);
    std::vector<std::pair<int64_t, int32_t>> items;
    for (int gi = 0; gi < G; gi++)
        for (int layer = 0; layer <= maxL + 1; layer++) {
            p->patch_work_off[(size_t)gi * (size_t)(maxL + 2) + (size_t)layer] = (int64_t)(work.size() / 2);
            if (layer < 2 || layer > maxL) continue;
            items.clear();
            for (int j = group_off[(size_t)gi]; j < group_off[(size_t)gi + 1]; j++) {
                const int a = group_angles[(size_t)j];
                const int32_t *first = p->h_patch_first.data() + (size_t)a * (size_t)(maxL + 2);
                for (int32_t q = first[layer]; q < first[layer + 1]; q++)
                    items.push_back({(int64_t)p->h_patch_rec[(size_t)q].z * 64 + (j - group_off[(size_t)gi]), q});
            }
            std::sort(items.begin(), items.end());
            const size_t m = items.size();
            if (m == 0) continue;
            const size_t slots = (m + 7) / 8;
            const size_t base = work.size();
            work.resize(base + slots * 8 * 2, make_int4(0, 0, 0, 0));      // (entries = 0: padding)
            for (int x = 0; x < 8; x++) {
                const size_t b0 = m * (size_t)x / 8, b1 = m * (size_t)(x + 1) / 8;
                for (size_t t = b0; t < b1; t++) {
                    const int32_t q = items[t].second;
                    const int2 r2 = p->h_patch_rec2[(size_t)q];
                    const int d = p->dir_of_active[(size_t)r2.y] > 0 ? 0 : 1;
                    const Direction &dir = d == 0 ? p->g->up : p->g->down;
                    const size_t slot = base + 2 * ((t - b0) * 8 + (size_t)x);
                    work[slot] = p->h_patch_rec[(size_t)q];
                    work[slot + 1] = make_int4(r2.x, r2.y | (d << 16), (int)(dir.reduced[(size_t)layer - 1] - 1), (int)(dir.reduced[(size_t)layer] - 1));
                }
            }
        }
    p->patch_work_off.back() = (int64_t)(work.size() / 2);
    VRT_HIP_TRY(hipMalloc((void **)&p->d_patch_work, sizeof(int4) * std::max<size_t>(work.size(), 1)));
    VRT_HIP_TRY(hipMemcpy(p->d_patch_work, work.data(), sizeof(int4) * work.size(), hipMemcpyHostToDevice));
    p->patch_work_groups = G;
    return VRT_OK;
}

// one layer of one stream group
// fills the block counts of a reduction request for NT-thread blocks
static void size_reduce(PatchReduce &red, int npair, int lgB, int NT)
{
    red.nred = 0;
    // elements per thread: one while the whole reduction is a fraction of a round of workgroups (a launch of a small
    // layer lasts as long as its slowest block), four otherwise
    int64_t total = 0;
    for (int r = 0; r < 2; r++)
        if (red.count[r] > 0 && red.Jd[r]) total += (int64_t)std::max(0, red.hi[r] - red.lo[r]) * npair;
    red.ppb = total <= (int64_t)NT * 1024 ? 1 : 4;
    const int64_t per = (int64_t)NT * red.ppb;               // pair elements per block
    const int nblock = pair_block_count(npair, lgB);
    for (int r = 0; r < 2; r++) {
        const int64_t len = red.hi[r] - red.lo[r];
        red.nblk[r] = 0;
        if (len > 0 && red.count[r] > 0 && red.Jd[r])
            for (int k = 0; k < nblock; k++) {
                int k0, lw;
                pair_block_of(k, npair, lgB, k0, lw);
                red.nblk[r] += (int)(((len << lw) + per - 1) / per);
            }
        red.nred += red.nblk[r];
    }
    red.nred = (red.nred + 7) / 8 * 8;
}

int launch_patch_layer(vrt_plan *p, const TileArgs &ta, int npair, int layer, int group, int Q, hipStream_t st,
                       bool f32, const PatchReduce *reduce)
{
    // workgroups per launch when VRT_PATCH_TARGET fixes them (0: the split is chosen below)
    const int target_wgs = p->tune.patch_target;
    const int maxL = p->tile_max_layers;
    const size_t wo = (size_t)group * (size_t)(maxL + 2) + (size_t)layer;
    const int64_t w0 = (layer >= 2 && layer <= maxL) ? p->patch_work_off[wo] : 0, w1 = (layer >= 2 && layer <= maxL) ? p->patch_work_off[wo + 1] : 0;
    PatchArgs pa;
    if (reduce) {
        pa.red = *reduce;
        size_reduce(pa.red, npair, native_lg(p, f32), p->patch_NT);
    }
    if (w1 <= w0 && pa.red.nred == 0) return VRT_OK;
    pa.ta = ta;
    pa.npair = npair;
    pa.layer = layer;
    pa.lgB = native_lg(p, f32);
    // fp32 storage: two neighbouring pairs of a block per workgroup as 16-byte accesses (k_patch_quad) when the pair
    // count is even (every block then holds >= 2 pairs and every plane starts 16-byte aligned; else the pair kernel)
    pa.quad = (f32 && pa.lgB >= 1 && p->patch_K == 1 && p->tune.patch_quad != 0 && (npair & 1) == 0) ? 1 : 0;
    if (pa.quad) Q = 1;
    // (a lone pair per workgroup is a latency chain: the plain kernel's single gather phase is 3 % shorter there)
    pa.lean = (p->tune.patch_lean != 0 && p->patch_K == 1 && Q == 1 && !pa.quad && npair >= 2) ? 1 : 0;
    const int lgS = pa.lgB - pa.quad;                          // log2 of the sibling workgroups per block
    pa.Q = Q;
    pa.nsplit = 1;
    pa.ngrp = 1 << lgS;
    if (w1 > w0) {
        const int64_t items = (w1 - w0) << lgS;                // work-list slots (a few of them padding) x siblings
        const int nblock = pair_block_count(npair, pa.lgB);
        const int steps_all = (nblock + Q - 1) / Q;            // Q blocks at a time
        // Workgroups per item (each walks its share of the item's `steps_all` pair steps with one read of the entry table).
        // Only BALANCED splits are candidates -- s = ceil(steps / c) for c = 1, 2, ...: a launch ends with its longest workgroup,
        // so among the splits with the same longest share the one with the fewest workgroups is the cheapest (8 steps over 7
        // workgroups last as long as over 4 and read the table 7 times).  Among those: the one whose launch comes closest to
        // T = min(900, 460 + 4100 / steps) workgroups -- the longer an item (a pair step is ~ 9 us), the closer a launch should
        // stay to ONE round on its stream's half of the chip (512 slots: a second, partly filled round costs a whole item's
        // time); short items tolerate, and for the launch's tail want, more and smaller workgroups -- but never less than that
        // one round.  Measured on C4's grid, 108 items per launch (profiles/r5/split_table.txt): 8 / 10 / 13 / 18 / 26 / 35 / 50
        // steps are fastest at 8 / 10 / 7 / 6 / 6 / 5 / 5 workgroups per item, C3 (10 steps, 180 items) at 5: what this picks.
        // VRT_PATCH_TARGET > 0: the fixed workgroup count per launch of rounds 3-5 (768); VRT_PATCH_SPLIT: s itself.
        int nsplit;
        if (p->tune.patch_split > 0) nsplit = std::min(steps_all, p->tune.patch_split);
        else if (target_wgs > 0) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(steps_all, (target_wgs + items - 1) / items));
        else {
            const double T = std::min(900.0, 460.0 + 4100.0 / (double)steps_all);
            const int s_min = (int)std::min<int64_t>(steps_all, std::max<int64_t>(1, (512 + items - 1) / items));
            double dmin = 1e300;
            for (int c = steps_all; c >= 1; c--) {
                const int sc = (steps_all + c - 1) / c;
                if (sc >= s_min) dmin = std::min(dmin, std::fabs(std::log((double)items * sc / T)));
            }
            nsplit = s_min;
            for (int c = steps_all; c >= 1; c--) {          // (s grows along the loop: the largest within 3 % of the closest)
                const int sc = (steps_all + c - 1) / c;
                if (sc >= s_min && std::fabs(std::log((double)items * sc / T)) <= dmin + 0.03) nsplit = sc;
            }
        }
        pa.nsplit = nsplit;
        pa.ngrp = nsplit << lgS;
    }
    pa.stride = p->patch_cap + 1;
    pa.cap = p->patch_cap;
    pa.dbg = kDiag ? p->tune.debug_flags : 0;
    pa.wrec = p->d_patch_work + 2 * w0;
    pa.e_pos = p->e_pos; pa.e_u1 = p->e_u1; pa.e_u2 = p->e_u2;
    pa.e_vis = p->e_vis; pa.e_loc = p->e_loc;
    pa.e_w1 = p->e_w1; pa.e_w2 = p->e_w2; pa.e_r1 = p->e_r1; pa.e_r2 = p->e_r2;
    const dim3 grid((unsigned)(pa.red.nred + (w1 - w0) * pa.ngrp));
    const size_t lds = (size_t)(pa.quad ? 2 : Q) * (size_t)pa.stride * sizeof(double2) + (size_t)pa.cap * (4 * sizeof(double) + 5 * sizeof(int32_t));
    const int rc = f32 ? launch_mode<float>(ta.alpha_mode, p->patch_K, Q, p->patch_NT, grid, lds, st, pa)
                       : launch_mode<double>(ta.alpha_mode, p->patch_K, Q, p->patch_NT, grid, lds, st, pa);
    return rc;
}

// ---- the chained launch: host side ------------------------------------------------------------------------------------
// (1 << lgS) sibling workgroups per pair block exist only in the pair-block layouts (VRT_PAIR_BLOCK > 1 with doubles,
// an odd pair count with floats): those, the other kernel shapes and thread counts keep the per-layer launches.
bool patch_chain_possible(const vrt_plan *p, int npair, bool f32)
{
    if (!p->patch_ok || p->tune.patch_chain == 0 || p->patch_K != 1 || p->patch_NT != 512) return false;
    const int lgB = native_lg(p, f32);
    const bool quad = f32 && lgB >= 1 && p->tune.patch_quad != 0 && (npair & 1) == 0;
    if (lgB - (quad ? 1 : 0) != 0) return false;
    if (!f32 && p->tune.patch_lean == 0) return false;             // VRT_PATCH_LEAN=0 asks for the 72-register kernel
    if (p->tile_max_layers >= 65535 || p->A > 63) return false;    // item encoding
    if (p->tune.patch_chain == 1) return true;
    // auto: the chained launch where a layer alone cannot fill the chip -- its items cost ~5 us more than those of a
    // per-layer launch (ticket, item record, the final drain before the progress word), which pays while the per-layer
    // launches are bound by their 286-launch chain (C2 0.85 -> 0.54 ms, 1 M sites x 7 wavelengths 2.10 -> 1.73 ms)
    // and not when the pair loops saturate the chip (C4 7.5 -> 8.4 ms).  Measured crossover:
    // patches of a layer (both directions) x wavelength-pair blocks ~ 1400 (C4's grid, 217 patches per layer: 6 pairs chained 2.30 ms
    // against 2.36 on per-layer launches, 7 pairs 2.59 against 2.53: profiles/r5/chain_crossover.txt).  fp32 storage: on request only.
    if (f32) return false;
    const double per_layer = (double)p->n_patches / (double)std::max(1, p->tile_max_layers);
    return per_layer * (double)pair_block_count(npair, lgB) <= 1400.0;
}

// the chained launch's hand-off by the data itself (chain_data_wait): fp64 pair kernel only; auto: one or two wavelength
// pairs -- the fill of the planes costs 16 bytes per (site, angle, pair) and step, the progress words two fabric round
// trips per LAYER (C2: 0.54 -> see DESIGN.md section 5)
bool patch_chain_dataflag(const vrt_plan *p, int npair, bool f32)
{
    if (f32 || p->tune.chain_dataflag == 0 || native_lg(p, f32) != 0) return false;
    return p->tune.chain_dataflag == 1 || npair <= 2;
}

// Items per patch: enough that ONE layer (both directions) offers about as many items as the chip holds workgroups
// (1024: a workgroup takes its item in queue order and, if the item's layer is not ready, waits for it while holding
// its place -- items of a later layer started early are slots spent waiting), and at most VRT_CHAIN_PAIRS blocks each
static int chain_nsplit(const vrt_plan *p, int nblock)
{
    const int per = std::max(1, p->tune.chain_pairs);
    const double per_layer = (double)p->n_patches / (double)std::max(1, p->tile_max_layers);
    const int fill = (int)std::ceil(1024.0 / std::max(1.0, per_layer));
    return std::max(1, std::min({std::max((nblock + per - 1) / per, fill), nblock, 511}));
}

// Items of the chained launch for `npair` pairs in blocks of 2^lgB, `nsplit` items per patch: eight queues (XCD x
// takes queue x first), each in layer order: the solve items of a layer's patches -- sorted by (first owned position,
// angle) and cut into 8 runs as the per-layer work lists are, the splits of a patch next to each other -- followed
// by the J_dir items of the layer before (lagged: their dependencies have then long been claimed).
static int ensure_patch_chain(vrt_plan *p, int npair, int lgB, int nsplit, bool with_reduce)
{
    if (p->d_chain_items && p->chain_npair == npair && p->chain_lgB == lgB && p->chain_nsplit == nsplit &&
        p->chain_reduce == (with_reduce ? 1 : 0))
        return VRT_OK;
    {
        // the set in use goes to the plan's small cache, a cached set of this combination comes back (no rebuild, no free)
        auto stash = [&]() {
            if (!p->d_chain_items) return;
            vrt_plan::ChainSet cs;
            cs.npair = p->chain_npair; cs.lgB = p->chain_lgB; cs.nsplit = p->chain_nsplit; cs.reduce = p->chain_reduce;
            cs.items = p->d_chain_items; cs.deps = p->d_chain_deps; cs.n_items = p->chain_items;
            for (int x = 0; x <= 8; x++) cs.q_off[x] = p->chain_q_off[x];
            p->d_chain_items = nullptr; p->d_chain_deps = nullptr;
            p->chain_npair = -1;
            p->chain_cache.push_back(cs);
            if (p->chain_cache.size() > 3) {                  // (the oldest goes: hipFree waits for the device)
                (void)hipFree(p->chain_cache.front().items);
                (void)hipFree(p->chain_cache.front().deps);
                p->chain_cache.erase(p->chain_cache.begin());
            }
        };
        for (size_t c = 0; c < p->chain_cache.size(); c++) {
            const vrt_plan::ChainSet cs = p->chain_cache[c];
            if (cs.npair == npair && cs.lgB == lgB && cs.nsplit == nsplit && cs.reduce == (with_reduce ? 1 : 0)) {
                p->chain_cache.erase(p->chain_cache.begin() + (long)c);
                stash();
                p->d_chain_items = cs.items; p->d_chain_deps = cs.deps; p->chain_items = cs.n_items;
                for (int x = 0; x <= 8; x++) p->chain_q_off[x] = cs.q_off[x];
                p->chain_npair = npair; p->chain_lgB = lgB; p->chain_nsplit = nsplit; p->chain_reduce = with_reduce ? 1 : 0;
                const size_t words = (size_t)nsplit * (size_t)std::max<int64_t>(p->n_patches, 1);
                if (words <= p->chain_progress_cap) return VRT_OK;
                break;                                       // (cannot happen: the progress words only grow; rebuild below)
            }
        }
        stash();
    }
    const vrt_grid *g = p->g;
    const int maxL = p->tile_max_layers, A = p->A;
    const int64_t n = g->n, n_patches = p->n_patches;
    const int nblock = pair_block_count(npair, lgB);
    if ((int64_t)nsplit * n_patches >= ((int64_t)1 << 31)) return fail(VRT_EINVAL, "too many (split, patch) progress words");
    PatchArgs sp;                     // split_blocks reads nsplit and Q only
    sp.nsplit = nsplit;
    sp.Q = 1;
    std::vector<int> steps((size_t)nsplit);
    for (int s = 0; s < nsplit; s++) {
        int b0, b1;
        split_blocks(sp, s, nblock, b0, b1);
        steps[(size_t)s] = b1 - b0;
    }
    std::vector<int4> q[8];
    std::vector<int32_t> rdeps;       // dependency lists of the J_dir items (patch ids), appended behind the patches' own
    const size_t dep_base = p->h_patch_deps.size();
    std::vector<std::pair<int64_t, int32_t>> items;
    std::vector<int> dir_angles[2];
    for (int a = 0; a < A; a++) dir_angles[p->dir_of_active[(size_t)a] > 0 ? 0 : 1].push_back(a);
    // first owned position of each XCD run of (layer, direction): J_dir items follow the patches of their positions
    std::vector<int32_t> run_start((size_t)(maxL + 2) * 2 * 8, 0);
    auto first_of = [&](int a, int layer) { return p->h_patch_first[(size_t)a * (size_t)(maxL + 2) + (size_t)layer]; };
    // Every queue holds the same number of items per (layer, direction) segment (padding items end at once), so the
    // queues advance through the layers in step however the blocks of the grid are dealt to them: the blocks
    // x, x + 8, ... take queue x in order, a workgroup only waits for items of EARLIER layers, and with static
    // block -> XCD dealing (what the hardware does) the slots an XCD frees go to its own queue's next items.
    auto level_queues = [&]() {
        size_t longest = 0;
        for (int x = 0; x < 8; x++) longest = std::max(longest, q[x].size());
        for (int x = 0; x < 8; x++)
            while (q[x].size() < longest) {
                q[x].push_back(make_int4(0, 0, 0x40000000, 0));
                q[x].push_back(make_int4(0, 0, 0, 0));
                q[x].push_back(make_int4(0, 0, 0, 0));
            }
    };
    auto emit_reduce = [&](int layer) {
        if (!with_reduce) return;
        for (int d = 0; d < 2; d++) {
            const Direction &dir = d == 0 ? g->up : g->down;
            const int Ld = (int)dir.reduced.size() - 1;
            if (dir_angles[d].empty() || layer < 1 || layer > Ld) continue;
            const int64_t lo = dir.reduced[(size_t)layer - 1] - 1;
            const int64_t hi = layer == Ld ? n : dir.reduced[(size_t)layer] - 1;      // + the never-visited last site (I = 0)
            const int32_t *rs = run_start.data() + ((size_t)layer * 2 + (size_t)d) * 8;
            constexpr int64_t R = 512;
            for (int64_t clo = lo; clo < hi; clo += R) {
                const int64_t chi = std::min(hi, clo + R);
                int x = 0;
                if (layer >= 2) { while (x < 7 && rs[x + 1] <= clo) x++; }
                else x = (int)std::min<int64_t>(7, 8 * (clo - lo) / std::max<int64_t>(1, hi - lo));
                const size_t d0 = rdeps.size();
                if (layer >= 2)
                    for (int a : dir_angles[d])
                        for (int32_t pq = first_of(a, layer); pq < first_of(a, layer + 1); pq++) {
                            const int4 &rec = p->h_patch_rec[(size_t)pq];
                            if (rec.z < chi && rec.z + rec.w > clo) rdeps.push_back(pq);
                        }
                for (int s = 0; s < nsplit; s++) {
                    q[x].push_back(make_int4((int)clo, (int)chi, (int)0x80000000u, (d << 6) | (s << 7)));
                    q[x].push_back(make_int4((int)(dep_base + d0), (int)(rdeps.size() - d0), 0, steps[(size_t)s]));
                    q[x].push_back(make_int4(0, 0, 0, 0));
                }
            }
        }
    };
    for (int layer = 2; layer <= maxL; layer++) {
        for (int d = 0; d < 2; d++) {
            const Direction &dir = d == 0 ? g->up : g->down;
            if (layer > (int)dir.reduced.size() - 1) continue;
            items.clear();
            for (size_t j = 0; j < dir_angles[d].size(); j++) {
                const int a = dir_angles[d][j];
                for (int32_t pq = first_of(a, layer); pq < first_of(a, layer + 1); pq++)
                    items.push_back({(int64_t)p->h_patch_rec[(size_t)pq].z * 64 + (int64_t)j, pq});
            }
            std::sort(items.begin(), items.end());
            const size_t m = items.size();
            int32_t *rs = run_start.data() + ((size_t)layer * 2 + (size_t)d) * 8;
            for (int x = 0; x < 8; x++) {
                const size_t t0 = m * (size_t)x / 8, t1 = m * (size_t)(x + 1) / 8;
                rs[x] = t0 < m ? p->h_patch_rec[(size_t)items[t0].second].z : INT32_MAX;
                for (size_t t = t0; t < t1; t++) {
                    const int32_t pq = items[t].second;
                    const int4 &rec = p->h_patch_rec[(size_t)pq];
                    const int2 &rec2 = p->h_patch_rec2[(size_t)pq];
                    const int64_t o0 = p->h_patch_dep_off[(size_t)pq], o1 = p->h_patch_dep_off[(size_t)pq + 1];
                    for (int s = 0; s < nsplit; s++) {
                        q[x].push_back(make_int4(rec.x, rec.z, rec.y | (rec.w << 10) | (rec2.x << 20),
                                                 rec2.y | (d << 6) | (s << 7) | (layer << 16)));
                        q[x].push_back(make_int4((int)o0, (int)(o1 - o0), pq, 0));
                        q[x].push_back(make_int4((int)(dir.reduced[(size_t)layer - 1] - 1), (int)(dir.reduced[(size_t)layer] - 1), 0, 0));
                    }
                }
            }
            level_queues();
        }
        emit_reduce(layer - 1);
        level_queues();
    }
    emit_reduce(maxL);
    if (maxL < 1) emit_reduce(1);                         // (a grid of one layer: its J_dir items, once)
    level_queues();
    std::vector<int4> all;
    for (int x = 0; x < 8; x++) {
        p->chain_q_off[x] = (int)(all.size() / 3);
        all.insert(all.end(), q[x].begin(), q[x].end());
        std::vector<int4>().swap(q[x]);
    }
    p->chain_q_off[8] = (int)(all.size() / 3);
    if (all.size() / 3 >= (size_t)INT32_MAX) return fail(VRT_EINVAL, "too many items for the chained launch");
    std::vector<int32_t> deps(p->h_patch_deps);
    deps.insert(deps.end(), rdeps.begin(), rdeps.end());
    if (p->d_chain_items) { (void)hipFree(p->d_chain_items); p->d_chain_items = nullptr; }     // (stashed above: NULL here)
    if (p->d_chain_deps) { (void)hipFree(p->d_chain_deps); p->d_chain_deps = nullptr; }
    VRT_HIP_TRY(hipMalloc((void **)&p->d_chain_items, sizeof(int4) * std::max<size_t>(all.size(), 1)));
    VRT_HIP_TRY(hipMalloc((void **)&p->d_chain_deps, sizeof(int32_t) * std::max<size_t>(deps.size(), 1)));
    VRT_HIP_TRY(hipMemcpy(p->d_chain_items, all.data(), sizeof(int4) * all.size(), hipMemcpyHostToDevice));
    VRT_HIP_TRY(hipMemcpy(p->d_chain_deps, deps.data(), sizeof(int32_t) * deps.size(), hipMemcpyHostToDevice));
    const size_t words = (size_t)nsplit * (size_t)std::max<int64_t>(n_patches, 1);
    if (words > p->chain_progress_cap) {
        if (p->d_chain_progress) { (void)hipFree(p->d_chain_progress); p->d_chain_progress = nullptr; }
        p->chain_progress_cap = 0;
        VRT_HIP_TRY(hipMalloc((void **)&p->d_chain_progress, sizeof(uint32_t) * words));
        p->chain_progress_cap = words;
        p->chain_progress_fresh = true;
    }
    // (epochs keep counting across item sets: a word of an earlier set compares as "behind" whatever it meant there;
    // freshly allocated words are zeroed on the launch stream, ahead of the first launch that polls them)
    if (!p->d_chain_ctrl) VRT_HIP_TRY(hipMalloc((void **)&p->d_chain_ctrl, sizeof(uint32_t) * (kChainAbortWord + 4)));
    if (!p->h_chain_status) {
        VRT_HIP_TRY(hipHostMalloc((void **)&p->h_chain_status, 64, hipHostMallocMapped));
        *p->h_chain_status = 0;
        VRT_HIP_TRY(hipHostGetDevicePointer((void **)&p->d_chain_status, p->h_chain_status, 0));
    }
    p->chain_npair = npair;
    p->chain_lgB = lgB;
    p->chain_nsplit = nsplit;
    p->chain_reduce = with_reduce ? 1 : 0;
    p->chain_items = (int64_t)(all.size() / 3);
    return VRT_OK;
}

template <typename T, int AM>
static int launch_chain_mode(bool quad, bool dataflag, int64_t items, size_t lds, hipStream_t st, const ChainDev &h, const ChainDev *cd, uint32_t base)
{
    constexpr int NT = 512;
    if constexpr (sizeof(T) == 8) {
        if (dataflag) {
            hipLaunchKernelGGL((k_patch_chain_df<T, AM, NT>), dim3((unsigned)items), dim3(NT), lds, st, h, cd, base);
            return VRT_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (quad) {
            hipLaunchKernelGGL((k_patch_chain_quad<AM, NT>), dim3((unsigned)items), dim3(NT), lds, st, h, cd, base);
            return VRT_OK;
        }
    }
    hipLaunchKernelGGL((k_patch_chain<T, AM, NT>), dim3((unsigned)items), dim3(NT), lds, st, h, cd, base);
    return VRT_OK;
}

// ONE launch for every layer of every active angle (and J_dir of both directions when `reduce` is given: its weights,
// angle lists [0] = up, [1] = down and J_dir planes)
int chain_ctrl_words() { return kChainAbortWord + 4; }

int launch_patch_chain(vrt_plan *p, const TileArgs &ta, int npair, hipStream_t st, bool f32, const PatchReduce *reduce, bool dataflag,
                       bool ctrl_zeroed)
{
    // a give-up of an EARLIER chained launch of this plan (its results were wrong) is reported here at the latest
    if (int rc0 = patch_chain_check(p)) return rc0;
    const int lgB = native_lg(p, f32);
    const bool quad = f32 && lgB >= 1 && p->tune.patch_quad != 0 && (npair & 1) == 0;
    const int nblock = pair_block_count(npair, lgB);
    const int nsplit = chain_nsplit(p, nblock);
    int rc;
    if ((rc = ensure_patch_chain(p, npair, lgB, nsplit, reduce != nullptr))) return rc;
    if (p->chain_items == 0) return VRT_OK;
    static_assert(std::is_trivially_copyable<ChainDev>::value, "ChainDev is uploaded byte for byte");
    ChainDev h;
    std::memset(&h, 0, sizeof(h));
    h.ta = ta;
    h.npair = npair;
    h.nsplit = nsplit;
    h.cap = p->patch_cap;
    h.e_pos = p->e_pos; h.e_u1 = p->e_u1; h.e_u2 = p->e_u2;
    h.e_vis = p->e_vis; h.e_loc = p->e_loc;
    h.e_w1 = p->e_w1; h.e_w2 = p->e_w2; h.e_r1 = p->e_r1; h.e_r2 = p->e_r2;
    if (reduce) h.red = *reduce;
    h.items = p->d_chain_items;
    for (int x = 0; x <= 8; x++) h.q_off[x] = p->chain_q_off[x];
    h.deps = p->d_chain_deps;
    h.progress = p->d_chain_progress;
    h.ctrl = p->d_chain_ctrl;
    h.host_status = p->d_chain_status;
    h.n_patches = p->n_patches;
    h.spin_limit = (uint32_t)std::max(1, p->tune.chain_spin) << 10;
    {
        // (static only while every queue holds the same number of items: ensure_patch_chain pads them so)
        bool even = true;
        for (int x = 1; x < 8; x++) even = even && (p->chain_q_off[x + 1] - p->chain_q_off[x]) == (p->chain_q_off[1] - p->chain_q_off[0]);
        h.static_items = (!dataflag && even && p->tune.chain_static != 0) ? 1 : 0;
    }
    h.dbg = kDiag ? p->tune.debug_flags : 0;
    // the argument block travels only when it has changed (stream-ordered: behind the launches that read the old one)
    if (!p->d_chain_dev) VRT_HIP_TRY(hipMalloc((void **)&p->d_chain_dev, sizeof(ChainDev)));
    if (p->h_chain_dev.size() != sizeof(ChainDev) || std::memcmp(p->h_chain_dev.data(), &h, sizeof(h)) != 0) {
        if (!p->h_chain_dev_pinned) VRT_HIP_TRY(hipHostMalloc((void **)&p->h_chain_dev_pinned, sizeof(ChainDev), hipHostMallocDefault));
        // the pinned staging copy may still be in flight for an earlier launch: wait for that copy only
        if (p->chain_dev_ev_valid) VRT_HIP_TRY(hipEventSynchronize(p->chain_dev_ev));
        std::memcpy(p->h_chain_dev_pinned, &h, sizeof(h));
        VRT_HIP_TRY(hipMemcpyAsync(p->d_chain_dev, p->h_chain_dev_pinned, sizeof(h), hipMemcpyHostToDevice, st));
        if (!p->chain_dev_ev) VRT_HIP_TRY(hipEventCreateWithFlags(&p->chain_dev_ev, hipEventDisableTiming));
        VRT_HIP_TRY(hipEventRecord(p->chain_dev_ev, st));
        p->chain_dev_ev_valid = true;
        p->h_chain_dev.assign(reinterpret_cast<const char *>(&h), reinterpret_cast<const char *>(&h) + sizeof(h));
    }
    // epochs count the launches of this item set; stale words compare as "behind" while epochs differ by < 2^22
    p->chain_epoch++;
    if ((p->chain_epoch & 0x3FFFFFu) == 0u || p->chain_progress_fresh) {
        // stream-ordered (the plan's streams do not synchronise with the null stream)
        VRT_HIP_TRY(hipMemsetAsync(p->d_chain_progress, 0, sizeof(uint32_t) * p->chain_progress_cap, st));
        if ((p->chain_epoch & 0x3FFFFFu) == 0u) p->chain_epoch = 1;
        p->chain_progress_fresh = false;
    }
    const uint32_t base = p->chain_epoch << 8;
    if (!ctrl_zeroed) VRT_HIP_TRY(hipMemsetAsync(p->d_chain_ctrl, 0, sizeof(uint32_t) * (kChainAbortWord + 4), st));
    const size_t lds = (size_t)(quad ? 2 : 1) * (size_t)(p->patch_cap + 1) * sizeof(double2) + (size_t)p->patch_cap * (4 * sizeof(double) + 5 * sizeof(int32_t)) +
                       sizeof(uint32_t) * (kCtlWords + 64 + kChainDepLds);
    const ChainDev *cd = reinterpret_cast<const ChainDev *>(p->d_chain_dev);
    switch (ta.alpha_mode) {
    case VRT_ALPHA_SITE:
        rc = f32 ? launch_chain_mode<float, VRT_ALPHA_SITE>(quad, dataflag, p->chain_items, lds, st, h, cd, base)
                 : launch_chain_mode<double, VRT_ALPHA_SITE>(quad, dataflag, p->chain_items, lds, st, h, cd, base);
        break;
    case VRT_ALPHA_SITE_LAM:
        rc = f32 ? launch_chain_mode<float, VRT_ALPHA_SITE_LAM>(quad, dataflag, p->chain_items, lds, st, h, cd, base)
                 : launch_chain_mode<double, VRT_ALPHA_SITE_LAM>(quad, dataflag, p->chain_items, lds, st, h, cd, base);
        break;
    default:
        rc = f32 ? launch_chain_mode<float, VRT_ALPHA_ANGLE_SITE_LAM>(quad, dataflag, p->chain_items, lds, st, h, cd, base)
                 : launch_chain_mode<double, VRT_ALPHA_ANGLE_SITE_LAM>(quad, dataflag, p->chain_items, lds, st, h, cd, base);
        break;
    }
    if (rc) return rc;
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// VRT_OK, or the give-up of a chained launch that has finished since the last check (introspection; no synchronisation)
int patch_chain_check(vrt_plan *p)
{
    if (p->h_chain_status && *p->h_chain_status) {
        volatile uint32_t *hs = p->h_chain_status;
        char msg[384];
        std::snprintf(msg, sizeof(msg),
                      "a chained patch launch gave up waiting for a dependency of item %u (the results of that execute are "
                      "invalid; VRT_PATCH_CHAIN=0 selects the per-layer launches)", hs[1]);
        *p->h_chain_status = 0;
        return fail(VRT_ENODEVICE, msg);
    }
    return VRT_OK;
}

}  // namespace vrt
