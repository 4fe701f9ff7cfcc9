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
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vrt_device.h"
#include "vrt_internal.h"

namespace vrt {

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
    int duo;                  // fp64 storage: k_patch_duo (two pairs per step, one level loop)
    int lean;                 // k_patch_lean (64 registers: four workgroups per CU)
    int dbg;                  // timing diagnostics (-DVRT_DIAG build only, WRONG results): 1 no levels, 2 gathers ->
                              //   coalesced centre reads, 4 no weights arithmetic, 8 no stores, 16 / 32 / 64 no upwind gathers
                              //   of I / alpha / S, 128 no J reduction
    // J reduction riding along (lagged by one layer): the first nred blocks of the launch do not solve a patch
    // but form J_dir = Σ_a w_a I_a (reference's angle order inside the direction) over storage positions
    // [red_lo, red_hi) of up to two directions -- layers the stream's previous launch has made final
    PatchReduce red;
    const int32_t *work;      // this launch's work list: patch index per (slot, XCD), -1 = padding
    const int4 *rec;          // per patch: first entry, entries, first owned storage position, owned sites
    const int2 *rec2;         // per patch: in-layer levels, active angle
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

// exp(-x) for 5e-4 <= x <= 50 to ~2e-13 relative (the parity contract is 1e-10): Cody-Waite reduction as
// exp_neg (vrt_device.h), Taylor to r^10/10! (remainder 0.3466^11/11! = 2e-13)
__device__ __forceinline__ double exp_neg10(double x)
{
    const double t = -x;
    const double kf = rint(t * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, t);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.0 / 3628800.0;
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

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
// being interleaved -- the pipelined kernel holds the next pair's 32 landed registers while it computes.  The
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
__device__ __forceinline__ void split_blocks(const PatchArgs &pa, int split, int nblock, int &b0, int &b1)
{
    const int nstep = (nblock + pa.Q - 1) / pa.Q;
    const int base = nstep / pa.nsplit, rem = nstep - base * pa.nsplit;
    const int s0 = split * base + min(split, rem), s1 = s0 + base + (split < rem ? 1 : 0);
    b0 = s0 * pa.Q;
    b1 = min(nblock, s1 * pa.Q);
}

// ---- reduction role of a patch launch: J_dir of a finished layer (NT x ppb pair elements per block) --------------
template <typename T, int NT>
__device__ __forceinline__ void patch_reduce_role(const PatchArgs &pa)
{
    typedef typename Pair<T>::type T2;
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    int b = blockIdx.x, r = 0;
    if (b >= pa.red.nblk[0]) { b -= pa.red.nblk[0]; r = 1; }
    if (b >= pa.red.nblk[r]) return;                         // padding to a multiple of 8
    if (kDiag && (pa.dbg & 128)) return;
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
#ifndef VRT_PIPE_ATTR            // the pipelined kernel: 80 VGPRs = three 512-thread workgroups per CU (no scratch at that budget)
#define VRT_PIPE_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
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
    if ((int)blockIdx.x < pa.red.nred) {
        patch_reduce_role<T, NT>(pa);
        return;
    }
    const int bid = (int)blockIdx.x - pa.red.nred;
    const int x = bid & 7, rr = bid >> 3;
    const int grp = rr % pa.ngrp, sj = rr / pa.ngrp;
    const int item = pa.work[sj * 8 + x];
    if (item < 0) return;
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
    const int4 rec = pa.rec[item];
    const int2 rec2 = pa.rec2[item];
    const int ent_off = rec.x, n_ent = rec.y, own_lo = rec.z, own_cnt = rec.w;
    const int dbg = kDiag ? pa.dbg : 0;
    const int nlev = (dbg & 1) ? 0 : rec2.x, a = rec2.y;
    const int d = ta.angle_dir[a];
    const int lo = ta.lay[d][pa.layer - 1], hi = ta.lay[d][pa.layer];
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
            if (dbg & 2) { v1 = p; v2 = p; }
            // the intensity of an upwind counts when it lies in an EARLIER layer (final); an upwind in this
            // layer enters through the tile, one in a later layer reads 0 (:23): those gather the never-visited
            // site perm[n] at storage position n - 1, whose intensity is 0 in every plane
            int i1 = v1 < lo ? v1 : (int)n - 1, i2 = v2 < lo ? v2 : (int)n - 1;
            if (dbg & 16) { i1 = (int)n - 1; i2 = (int)n - 1; }
            const int av1 = (dbg & 32) ? p : v1, av2 = (dbg & 32) ? p : v2;       // traffic split (diagnostic build)
            const int sv1 = (dbg & 64) ? p : v1, sv2 = (dbg & 64) ? p : v2;
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
                if (dbg & 4) {
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
                    if (qok[qi] && !((dbg & 8) && ptile[qi * stride + i].x != 1.2345e300)) {
                        T2 *I = reinterpret_cast<T2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n + qbase[qi];
                        const unsigned off = (unsigned)(own_lo + i) << qsh[qi];
                        *reinterpret_cast<T2 *>(reinterpret_cast<char *>(I) + off) = from_d2<T>(ptile[qi * stride + i]);
                    }
                }
            }
        }
    }
}


// ---- what the one-entry-per-thread kernels (k_patch_pipe, k_patch_quad, k_patch_duo) share ----------------------------
// the work of this workgroup: patch, sibling number inside a pair block, blocks [b0, b1), the patch's record
struct PatchItem {
    int sib, b0, b1;
    int ent_off, n_ent, own_lo, own_cnt, nlev, a, d, lo, hi;
};
// lgS: log2 of the sibling workgroups per pair block (lgB, or lgB - 1 when a workgroup takes two pairs of a block);
// false: a padding slot, or nothing of the item for this workgroup
__device__ __forceinline__ bool patch_item(const PatchArgs &pa, int lgS, PatchItem &it)
{
    const int bid = (int)blockIdx.x - pa.red.nred;
    const int x = bid & 7, rr = bid >> 3;
    const int grp = rr % pa.ngrp, sj = rr / pa.ngrp;
    const int item = pa.work[sj * 8 + x];
    if (item < 0) return false;
    it.sib = grp & ((1 << lgS) - 1);
    split_blocks(pa, grp >> lgS, pair_block_count(pa.npair, pa.lgB), it.b0, it.b1);
    if (it.b0 >= it.b1) return false;
    const int4 rec = pa.rec[item];
    const int2 rec2 = pa.rec2[item];
    it.ent_off = rec.x; it.n_ent = rec.y; it.own_lo = rec.z; it.own_cnt = rec.w;
    it.nlev = rec2.x; it.a = rec2.y;
    it.d = pa.ta.angle_dir[it.a];
    it.lo = pa.ta.lay[it.d][pa.layer - 1];
    it.hi = pa.ta.lay[it.d][pa.layer];
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
    __device__ __forceinline__ void park(const PatchArgs &pa, const PatchItem &it, int tid) const
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

// ---- the default shape (one entry per thread, one pair at a time), software-pipelined ------------------------------
// A workgroup's pair costs ~15 000 cycles end to end: ~6 000 waiting for the eight gathers of an entry, ~1 500 of
// arithmetic, ~7 000 in the level loop (a barrier and an LDS round trip per level, ~18 levels on an inclined
// direction).  Here the gathers of the NEXT pair are issued before the level loop of the current one and land
// while it runs: their destination registers (32) are live across the loop instead of the arithmetic's
// temporaries, so the kernel keeps its 3 workgroups per CU.
template <typename T, int AM> struct PatchRaw {
    typedef typename Pair<T>::type T2;
    T2 S_c, S_1, S_2, I_1, I_2, a_c, a_1, a_2;           // AM == VRT_ALPHA_SITE: .x of the alphas only
};

template <typename T, int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_PIPE_ATTR
k_patch_pipe(PatchArgs pa)
{
    typedef typename Pair<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < pa.red.nred) {
        patch_reduce_role<T, NT>(pa);
        return;
    }
    PatchItem it;
    if (!patch_item(pa, pa.lgB, it)) return;
    const int sib = it.sib, b0 = it.b0, b1 = it.b1;
    constexpr int lgT2 = Log2Size<T2>::value;
    const int64_t n = ta.n;
    size_t qb;                                               // element base of the pair being loaded
    int sh;
    {
        int k0, lw;
        pair_block_of(b0, pa.npair, pa.lgB, k0, lw);
        if (sib >= (1 << lw)) return;
        qb = (size_t)k0 * (size_t)n + (size_t)sib;
        sh = lw + lgT2;
    }
    const int n_ent = it.n_ent, own_lo = it.own_lo, own_cnt = it.own_cnt, nlev = it.nlev, a = it.a, d = it.d;
    const int lo = it.lo, hi = it.hi;
    constexpr int CAP = NT;
    const EntryTable<CAP> tab(ptile, 1);
    tab.park(pa, it, tid);
    double *const s_w1 = tab.w1, *const s_w2 = tab.w2, *const s_r1 = tab.r1, *const s_r2 = tab.r2;
    int *const s_pos = tab.pos, *const s_u1 = tab.u1, *const s_u2 = tab.u2;
    uint32_t *const s_vis = tab.vis, *const s_loc = tab.loc;
    if (tid == 0) ptile[n_ent] = make_double2(0.0, 0.0);     // the zero slot

    const T2 *Sd = reinterpret_cast<const T2 *>(ta.S[d]);
    const T2 *Ia = reinterpret_cast<const T2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n;
    PatchRaw<T, AM> raw;
    // the eight gathers of this thread's entry for the pair at element base qb (own slots of LDS: no barrier)
    auto issue = [&]() {
        const int p = s_pos[tid], v1 = s_u1[tid], v2 = s_u2[tid];
        // an upwind's intensity counts when it lies in an EARLIER layer (final); otherwise the gather reads the
        // never-visited site at storage position n - 1, whose intensity is 0 in every plane (:23)
        const int i1 = v1 < lo ? v1 : (int)n - 1, i2 = v2 < lo ? v2 : (int)n - 1;
        const unsigned op = (unsigned)p << sh, o1 = (unsigned)v1 << sh, o2 = (unsigned)v2 << sh;
        auto at = [](const T2 *base, unsigned off) { return *reinterpret_cast<const T2 *>(reinterpret_cast<const char *>(base) + off); };
        if constexpr (AM == VRT_ALPHA_SITE) {
            const T *__restrict__ Al = reinterpret_cast<const T *>(ta.alpha[d]);
            raw.a_c.x = Al[p]; raw.a_1.x = Al[v1]; raw.a_2.x = Al[v2];
        } else {
            const T2 *__restrict__ Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const T2 *>(ta.alpha[d]) + qb
                                                                  : reinterpret_cast<const T2 *>(ta.alpha_angle) + (size_t)a * pa.npair * (size_t)n + qb;
            raw.a_c = at(Al, op); raw.a_1 = at(Al, o1); raw.a_2 = at(Al, o2);
        }
        raw.S_c = at(Sd + qb, op); raw.S_1 = at(Sd + qb, o1); raw.S_2 = at(Sd + qb, o2);
        raw.I_1 = at(Ia + qb, (unsigned)i1 << sh); raw.I_2 = at(Ia + qb, (unsigned)i2 << sh);
    };
    issue();
    for (int bk = b0;; bk++) {
        // ---- integration coefficients of the entry for the pair that has landed ---------------------------------
        double2 c, g1, g2;
        {
            const int v1 = s_u1[tid], v2 = s_u2[tid];
            const bool in1 = (v1 >= lo) & (v1 < hi), in2 = (v2 >= lo) & (v2 < hi);
            const double rh1 = 0.5 * s_r1[tid], rh2 = 0.5 * s_r2[tid];  // exact: r (α_c + α_u) / 2 = (r / 2)(α_c + α_u)
            double2 a_c, a_1, a_2;
            if constexpr (AM == VRT_ALPHA_SITE) {
                a_c = make_double2((double)raw.a_c.x, (double)raw.a_c.x);
                a_1 = make_double2((double)raw.a_1.x, (double)raw.a_1.x);
                a_2 = make_double2((double)raw.a_2.x, (double)raw.a_2.x);
            } else {
                a_c = to_d2(raw.a_c); a_1 = to_d2(raw.a_1); a_2 = to_d2(raw.a_2);
            }
            const double2 S_c = to_d2(raw.S_c), S_1 = to_d2(raw.S_1), S_2 = to_d2(raw.S_2);
            const double2 I_1 = to_d2(raw.I_1), I_2 = to_d2(raw.I_2);
            // the four optical depths first (frees the six alpha registers), then one upwind of one wavelength at a time
            const double d1x = rh1 * (a_c.x + a_1.x), d2x = rh2 * (a_c.x + a_2.x);
            double d1y = rh1 * (a_c.y + a_1.y), d2y = rh2 * (a_c.y + a_2.y);
            entry_lambda_seq(d1x, d2x, s_w1 + tid, s_w2 + tid, in1, in2, S_c.x, S_1.x, S_2.x, I_1.x, I_2.x, c.x, g1.x, g2.x, d1y);
            double sink = 0.0;
            entry_lambda_seq(d1y, d2y, s_w1 + tid, s_w2 + tid, in1, in2, S_c.y, S_1.y, S_2.y, I_1.y, I_2.y, c.y, g1.y, g2.y, sink);
        }
        // ---- the next pair's gathers go out now and land during the level loop ---------------------------------
        // (compiler fence tied to the coefficients: issued before the arithmetic has consumed the landed pair, the
        // gathers would need a second set of destination registers)
        asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(g1.x), "+v"(g1.y), "+v"(g2.x), "+v"(g2.y) : : "memory");
        const size_t qb_cur = qb;
        const int sh_cur = sh;
        bool more = false;
        if (bk + 1 < b1) {
            int k0, lw;
            pair_block_of(bk + 1, pa.npair, pa.lgB, k0, lw);
            if (sib < (1 << lw)) {
                more = true;
                qb = (size_t)k0 * (size_t)n + (size_t)sib;
                sh = lw + lgT2;
                issue();
            }
        }
        asm volatile("" ::: "memory");
        // ---- the patch's Gauss-Seidel levels on the LDS tile ------------------------------------------------------
        uint32_t vis = s_vis[tid];
        const uint32_t loc = s_loc[tid];
        {
            double z;                                                    // made here: a hoisted zero would hold four
            asm volatile("v_mov_b64 %0, 0" : "=v"(z));                   // registers across the whole loop
            if (tid < n_ent) ptile[tid] = make_double2(z, z);            // I = zero(S), :23
        }
        __syncthreads();
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
        // ---- final intensities of the owned sites --------------------------------------------------------------
        if (tid < own_cnt) {
            T2 *I = reinterpret_cast<T2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n + qb_cur;
            const unsigned off = (unsigned)(own_lo + tid) << sh_cur;
            *reinterpret_cast<T2 *>(reinterpret_cast<char *>(I) + off) = from_d2<T>(ptile[tid]);
        }
        if (!more) break;
        __syncthreads();                                                 // the tile is rewritten by the next pair
    }
}


// ---- fp32 storage, FOUR wavelengths per lane ---------------------------------------------------------------------
// With float values a wavelength pair is an 8-byte access, and the patch kernel issues as many memory instructions
// per wavelength as with doubles: the memory path, which bounds it, sees twice the requests
// per byte.  In the layout with two (or more) pairs of a site side by side (pair blocks, vrt_device.h) two
// neighbouring pairs are ONE 16-byte access: this kernel solves both at once -- eight float4 gathers per entry (the
// three alphas first, then S and I under the weights, as in k_patch_lean), the four evaluations of the weights one
// after the other (compiler fences), two planes of the LDS tile walked by one level loop, one float4 store.  Half
// the memory instructions and half the barriers per wavelength.  (An odd pair count leaves a last block of one pair: the host then launches the pair kernel.)
template <int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_PIPE_ATTR
k_patch_quad(PatchArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < pa.red.nred) {
        patch_reduce_role<float, NT>(pa);
        return;
    }
    // workgroup (item, sib2, split): the pairs 2 sib2, 2 sib2 + 1 of every block among blocks b0 .. b1-1
    PatchItem it;
    if (!patch_item(pa, pa.lgB - 1, it)) return;
    const int sib2 = it.sib, b0 = it.b0, b1 = it.b1;
    const int64_t n = ta.n;
    const int n_ent = it.n_ent, own_lo = it.own_lo, own_cnt = it.own_cnt, nlev = it.nlev, a = it.a, d = it.d;
    const int lo = it.lo, hi = it.hi;
    constexpr int CAP = NT;
    double2 *tileA = ptile, *tileB = ptile + (CAP + 1);
    const EntryTable<CAP> tab(ptile, 2);
    tab.park(pa, it, tid);
    double *const s_w1 = tab.w1, *const s_w2 = tab.w2, *const s_r1 = tab.r1, *const s_r2 = tab.r2;
    int *const s_pos = tab.pos, *const s_u1 = tab.u1, *const s_u2 = tab.u2;
    uint32_t *const s_vis = tab.vis, *const s_loc = tab.loc;
    if (tid == 0) {
        tileA[n_ent] = make_double2(0.0, 0.0);               // the zero slots
        tileB[n_ent] = make_double2(0.0, 0.0);
    }
    const float2 *Sd = reinterpret_cast<const float2 *>(ta.S[d]);
    const float2 *Ia = reinterpret_cast<const float2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n;
    auto at4 = [](const float2 *base, unsigned off) { return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off); };
    for (int bk = b0; bk < b1; bk++) {
        int k0, lw;
        pair_block_of(bk, pa.npair, pa.lgB, k0, lw);
        if (2 * sib2 >= (1 << lw)) break;                    // block widths only shrink (and are >= 2: even pair count)
        const size_t qb = (size_t)k0 * (size_t)n + (size_t)(2 * sib2);
        const int sh = lw + 3;                               // log2 bytes per site of the block
        const int p = s_pos[tid], v1 = s_u1[tid], v2 = s_u2[tid];
        // ---- the eight optical depths from the three alpha gathers (two gather phases as in k_patch_lean: the
        // alphas are dead before the S / I gathers land) ------------------------------------------------------------
        double d1x, d2x, d1y, d2y, d1z, d2z, d1w, d2w;
        {
            float4 a_c, a_1, a_2;
            if constexpr (AM == VRT_ALPHA_SITE) {
                const float *__restrict__ A1 = reinterpret_cast<const float *>(ta.alpha[d]);
                const float c0 = A1[p], c1 = A1[v1], c2 = A1[v2];
                a_c = make_float4(c0, c0, c0, c0); a_1 = make_float4(c1, c1, c1, c1); a_2 = make_float4(c2, c2, c2, c2);
            } else {
                const float2 *Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const float2 *>(ta.alpha[d]) + qb
                                                            : reinterpret_cast<const float2 *>(ta.alpha_angle) + (size_t)a * pa.npair * (size_t)n + qb;
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
            const float4 I_1 = at4(Ia + qb, (unsigned)i1 << sh), I_2 = at4(Ia + qb, (unsigned)i2 << sh);
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
        __syncthreads();
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
            float2 *I = reinterpret_cast<float2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n + qb;
            char *dst = reinterpret_cast<char *>(I) + ((unsigned)(own_lo + tid) << sh);
            const double2 ra = tileA[tid], rb = tileB[tid];
            *reinterpret_cast<float4 *>(dst) = make_float4((float)ra.x, (float)ra.y, (float)rb.x, (float)rb.y);
        }
        __syncthreads();                                     // the tiles are rewritten by the next block
    }
}


// ---- fp64 storage, TWO wavelength pairs per workgroup step ----------------------------------------------------------
// The level loop (a barrier, an LDS round trip and a handful of scalar / vector instructions per level and wave,
// ~18 levels on an inclined direction) costs about as many instruction issues as the weights of a pair.  Here two
// pairs share it: the gathers and the arithmetic of pair A, then those of pair B (so that only one set of eight
// landed gathers is live at a time: 80 registers hold), then ONE level loop over both tile planes.
template <int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_PIPE_ATTR
k_patch_duo(PatchArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < pa.red.nred) {
        patch_reduce_role<double, NT>(pa);
        return;
    }
    PatchItem it;
    if (!patch_item(pa, pa.lgB, it)) return;
    const int sib = it.sib, b0 = it.b0, b1 = it.b1;
    const int64_t n = ta.n;
    const int n_ent = it.n_ent, own_lo = it.own_lo, own_cnt = it.own_cnt, nlev = it.nlev, a = it.a, d = it.d;
    const int lo = it.lo, hi = it.hi;
    constexpr int CAP = NT;
    double2 *tileA = ptile, *tileB = ptile + (CAP + 1);
    const EntryTable<CAP> tab(ptile, 2);
    tab.park(pa, it, tid);
    double *const s_w1 = tab.w1, *const s_w2 = tab.w2, *const s_r1 = tab.r1, *const s_r2 = tab.r2;
    int *const s_pos = tab.pos, *const s_u1 = tab.u1, *const s_u2 = tab.u2;
    uint32_t *const s_vis = tab.vis, *const s_loc = tab.loc;
    if (tid == 0) {
        tileA[n_ent] = make_double2(0.0, 0.0);               // the zero slots
        tileB[n_ent] = make_double2(0.0, 0.0);
    }
    const double2 *Sd = reinterpret_cast<const double2 *>(ta.S[d]);
    const double2 *Ia = reinterpret_cast<const double2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n;
    // gathers + weights of the entry for the pair at element base qb -> (c, g1, g2)
    auto coefficients = [&](size_t qb, int sh, double2 &c, double2 &g1, double2 &g2) {
        const int p = s_pos[tid], v1 = s_u1[tid], v2 = s_u2[tid];
        const int i1 = v1 < lo ? v1 : (int)n - 1, i2 = v2 < lo ? v2 : (int)n - 1;
        const unsigned op = (unsigned)p << sh, o1 = (unsigned)v1 << sh, o2 = (unsigned)v2 << sh;
        auto at = [](const double2 *base, unsigned off) { return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + off); };
        double2 a_c, a_1, a_2;
        if constexpr (AM == VRT_ALPHA_SITE) {
            const double *__restrict__ A1 = ta.alpha[d];
            const double c0 = A1[p], c1 = A1[v1], c2 = A1[v2];
            a_c = make_double2(c0, c0); a_1 = make_double2(c1, c1); a_2 = make_double2(c2, c2);
        } else {
            const double2 *Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const double2 *>(ta.alpha[d]) + qb
                                                         : reinterpret_cast<const double2 *>(ta.alpha_angle) + (size_t)a * pa.npair * (size_t)n + qb;
            a_c = at(Al, op); a_1 = at(Al, o1); a_2 = at(Al, o2);
        }
        const double2 S_c = at(Sd + qb, op), S_1 = at(Sd + qb, o1), S_2 = at(Sd + qb, o2);
        const double2 I_1 = at(Ia + qb, (unsigned)i1 << sh), I_2 = at(Ia + qb, (unsigned)i2 << sh);
        const bool in1 = (v1 >= lo) & (v1 < hi), in2 = (v2 >= lo) & (v2 < hi);
        const double rh1 = 0.5 * s_r1[tid], rh2 = 0.5 * s_r2[tid];
        const double d1x = rh1 * (a_c.x + a_1.x), d2x = rh2 * (a_c.x + a_2.x);
        double d1y = rh1 * (a_c.y + a_1.y), d2y = rh2 * (a_c.y + a_2.y);
        entry_lambda_seq(d1x, d2x, s_w1 + tid, s_w2 + tid, in1, in2, S_c.x, S_1.x, S_2.x, I_1.x, I_2.x, c.x, g1.x, g2.x, d1y);
        double sink = 0.0;
        entry_lambda_seq(d1y, d2y, s_w1 + tid, s_w2 + tid, in1, in2, S_c.y, S_1.y, S_2.y, I_1.y, I_2.y, c.y, g1.y, g2.y, sink);
    };
    for (int bk = b0; bk < b1; bk += 2) {
        int k0, lw;
        pair_block_of(bk, pa.npair, pa.lgB, k0, lw);
        if (sib >= (1 << lw)) break;                         // block widths only shrink
        const size_t qbA = (size_t)k0 * (size_t)n + (size_t)sib;
        const int shA = lw + 4;
        bool haveB = bk + 1 < b1;
        size_t qbB = qbA;
        int shB = shA;
        if (haveB) {
            pair_block_of(bk + 1, pa.npair, pa.lgB, k0, lw);
            haveB = sib < (1 << lw);
            if (haveB) { qbB = (size_t)k0 * (size_t)n + (size_t)sib; shB = lw + 4; }
        }
        double2 cA, g1A, g2A, cB, g1B, g2B;
        coefficients(qbA, shA, cA, g1A, g2A);
        asm volatile("" : "+v"(cA.x), "+v"(cA.y), "+v"(g1A.x), "+v"(g1A.y), "+v"(g2A.x), "+v"(g2A.y) : : "memory");
        if (haveB) coefficients(qbB, shB, cB, g1B, g2B);
        else { cB = make_double2(0.0, 0.0); g1B = cB; g2B = cB; }
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
        __syncthreads();
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
        if (tid < own_cnt) {
            double2 *I = reinterpret_cast<double2 *>(ta.I) + (size_t)a * pa.npair * (size_t)n;
            *reinterpret_cast<double2 *>(reinterpret_cast<char *>(I + qbA) + ((unsigned)(own_lo + tid) << shA)) = tileA[tid];
            if (haveB) *reinterpret_cast<double2 *>(reinterpret_cast<char *>(I + qbB) + ((unsigned)(own_lo + tid) << shB)) = tileB[tid];
        }
        __syncthreads();                                     // the tiles are rewritten by the next step
    }
}


// ---- one pair at a time within 64 registers: FOUR workgroups per CU ---------------------------------------------------
// No unit of the chip is saturated by the patch kernel; its phases (gathers, arithmetic, level loop) overlap only as
// far as three 512-thread workgroups per CU allow (72 registers).  This form fits the 64 of a fourth: the three
// alpha gathers first, the four optical depths from them (the alphas die), then the five S / I gathers in flight
// while the weights are evaluated one after the other, each folded into its share of the visit as soon as it exists.
#ifndef VRT_LEAN_ATTR
#define VRT_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
template <typename T, int AM, int NT>
__global__ void __launch_bounds__(NT) VRT_LEAN_ATTR
k_patch_lean(PatchArgs pa)
{
    typedef typename Pair<T>::type T2;
    extern __shared__ __attribute__((aligned(16))) double2 ptile[];
    const TileArgs &ta = pa.ta;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < pa.red.nred) {
        patch_reduce_role<T, NT>(pa);
        return;
    }
    PatchItem it;
    if (!patch_item(pa, pa.lgB, it)) return;
    constexpr int lgT2 = Log2Size<T2>::value;
    const int64_t n = ta.n;
    constexpr int CAP = NT;
    const EntryTable<CAP> tab(ptile, 1);
    tab.park(pa, it, tid);
    if (tid == 0) ptile[it.n_ent] = make_double2(0.0, 0.0);  // the zero slot
    const T2 *Sd = reinterpret_cast<const T2 *>(ta.S[it.d]);
    const T2 *Ia = reinterpret_cast<const T2 *>(ta.I) + (size_t)it.a * pa.npair * (size_t)n;
    auto at = [](const T2 *base, unsigned off) { return *reinterpret_cast<const T2 *>(reinterpret_cast<const char *>(base) + off); };
    for (int bk = it.b0; bk < it.b1; bk++) {
        int k0, lw;
        pair_block_of(bk, pa.npair, pa.lgB, k0, lw);
        if (it.sib >= (1 << lw)) break;                      // block widths only shrink
        const size_t qb = (size_t)k0 * (size_t)n + (size_t)it.sib;
        const int sh = lw + lgT2;
        double2 c, g1, g2;
        {
            const int p = tab.pos[tid], v1 = tab.u1[tid], v2 = tab.u2[tid];
            // ---- the four optical depths: r (alpha_c + alpha_u) / 2 = (r / 2)(alpha_c + alpha_u) ----------------
            double d1x, d2x, d1y, d2y;
            {
                double2 a_c, a_1, a_2;
                if constexpr (AM == VRT_ALPHA_SITE) {
                    const T *__restrict__ A1 = reinterpret_cast<const T *>(ta.alpha[it.d]);
                    const double c0 = A1[p], c1 = A1[v1], c2 = A1[v2];
                    a_c = make_double2(c0, c0); a_1 = make_double2(c1, c1); a_2 = make_double2(c2, c2);
                } else {
                    const T2 *Al = AM == VRT_ALPHA_SITE_LAM ? reinterpret_cast<const T2 *>(ta.alpha[it.d]) + qb
                                                            : reinterpret_cast<const T2 *>(ta.alpha_angle) + (size_t)it.a * pa.npair * (size_t)n + qb;
                    a_c = to_d2(at(Al, (unsigned)p << sh)); a_1 = to_d2(at(Al, (unsigned)v1 << sh)); a_2 = to_d2(at(Al, (unsigned)v2 << sh));
                }
                const double rh1 = 0.5 * tab.r1[tid], rh2 = 0.5 * tab.r2[tid];
                d1x = rh1 * (a_c.x + a_1.x); d2x = rh2 * (a_c.x + a_2.x);
                d1y = rh1 * (a_c.y + a_1.y); d2y = rh2 * (a_c.y + a_2.y);
            }
            asm volatile("" : "+v"(d1x), "+v"(d2x), "+v"(d1y), "+v"(d2y) : : "memory");
            // ---- S and I in flight, the weights one after the other -------------------------------------------------
            // an upwind's intensity counts when it lies in an EARLIER layer (final); otherwise the gather reads the
            // never-visited site at storage position n - 1, whose intensity is 0 in every plane (:23)
            const int i1 = v1 < it.lo ? v1 : (int)n - 1, i2 = v2 < it.lo ? v2 : (int)n - 1;
            const T2 rS_c = at(Sd + qb, (unsigned)p << sh), rS_1 = at(Sd + qb, (unsigned)v1 << sh), rS_2 = at(Sd + qb, (unsigned)v2 << sh);
            const T2 rI_1 = at(Ia + qb, (unsigned)i1 << sh), rI_2 = at(Ia + qb, (unsigned)i2 << sh);
            const bool in1 = (v1 >= it.lo) & (v1 < it.hi), in2 = (v2 >= it.lo) & (v2 < it.hi);
            entry_lambda_seq(d1x, d2x, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.x, (double)rS_1.x, (double)rS_2.x,
                             (double)rI_1.x, (double)rI_2.x, c.x, g1.x, g2.x, d1y);
            double sink = 0.0;
            entry_lambda_seq(d1y, d2y, tab.w1 + tid, tab.w2 + tid, in1, in2, (double)rS_c.y, (double)rS_1.y, (double)rS_2.y,
                             (double)rI_1.y, (double)rI_2.y, c.y, g1.y, g2.y, sink);
        }
        // ---- the patch's Gauss-Seidel levels on the LDS tile ------------------------------------------------------
        uint32_t vis = tab.vis[tid];
        const uint32_t loc = tab.loc[tid];
        {
            double z;
            asm volatile("v_mov_b64 %0, 0" : "=v"(z));
            if (tid < it.n_ent) ptile[tid] = make_double2(z, z);         // I = zero(S), :23
        }
        __syncthreads();
        for (int t = 1; t <= it.nlev; t++) {
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
        if (tid < it.own_cnt) {
            T2 *I = reinterpret_cast<T2 *>(ta.I) + (size_t)it.a * pa.npair * (size_t)n + qb;
            *reinterpret_cast<T2 *>(reinterpret_cast<char *>(I) + ((unsigned)(it.own_lo + tid) << sh)) = from_d2<T>(ptile[tid]);
        }
        __syncthreads();                                                 // the tile is rewritten by the next pair
    }
}

// the instantiated launch shapes (entries per thread, pairs at a time, threads)
#define VRT_PATCH_SHAPES(X) \
    X(1, 1, 256) X(1, 1, 512) X(1, 1, 1024) X(2, 1, 256) X(2, 1, 512) X(1, 2, 256) X(1, 2, 512) X(1, 2, 1024) X(2, 2, 512)

template <typename T, int AM>
static int launch_shape(int K, int Q, int NT, dim3 grid, size_t lds, hipStream_t st, const PatchArgs &pa, bool pipe)
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
    if constexpr (sizeof(T) == 8) {
        if (pa.duo && K == 1 && Q == 2 && !(kDiag && pa.dbg)) {
            switch (NT) {
            case 256: hipLaunchKernelGGL((k_patch_duo<AM, 256>), grid, dim3(256), lds, st, pa); return VRT_OK;
            case 512: hipLaunchKernelGGL((k_patch_duo<AM, 512>), grid, dim3(512), lds, st, pa); return VRT_OK;
            case 1024: hipLaunchKernelGGL((k_patch_duo<AM, 1024>), grid, dim3(1024), lds, st, pa); return VRT_OK;
            default: break;
            }
        }
    }
    if (pa.lean && K == 1 && Q == 1 && !(kDiag && pa.dbg)) {
        switch (NT) {
        case 256: hipLaunchKernelGGL((k_patch_lean<T, AM, 256>), grid, dim3(256), lds, st, pa); return VRT_OK;
        case 512: hipLaunchKernelGGL((k_patch_lean<T, AM, 512>), grid, dim3(512), lds, st, pa); return VRT_OK;
        case 1024: hipLaunchKernelGGL((k_patch_lean<T, AM, 1024>), grid, dim3(1024), lds, st, pa); return VRT_OK;
        default: break;
        }
    }
    if (pipe && K == 1 && Q == 1 && !(kDiag && pa.dbg)) {
        switch (NT) {
        case 256: hipLaunchKernelGGL((k_patch_pipe<T, AM, 256>), grid, dim3(256), lds, st, pa); return VRT_OK;
        case 512: hipLaunchKernelGGL((k_patch_pipe<T, AM, 512>), grid, dim3(512), lds, st, pa); return VRT_OK;
        case 1024: hipLaunchKernelGGL((k_patch_pipe<T, AM, 1024>), grid, dim3(1024), lds, st, pa); return VRT_OK;
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
static int launch_mode(int am, int K, int Q, int NT, dim3 grid, size_t lds, hipStream_t st, const PatchArgs &pa, bool pipe)
{
    switch (am) {
    case VRT_ALPHA_SITE: return launch_shape<T, VRT_ALPHA_SITE>(K, Q, NT, grid, lds, st, pa, pipe);
    case VRT_ALPHA_SITE_LAM: return launch_shape<T, VRT_ALPHA_SITE_LAM>(K, Q, NT, grid, lds, st, pa, pipe);
    default: return launch_shape<T, VRT_ALPHA_ANGLE_SITE_LAM>(K, Q, NT, grid, lds, st, pa, pipe);
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
    std::vector<int32_t> work;
    p->patch_work_off.assign((size_t)G * (size_t)(maxL + 2) + 1, 0);
    std::vector<std::pair<int64_t, int32_t>> items;
    for (int gi = 0; gi < G; gi++)
        for (int layer = 0; layer <= maxL + 1; layer++) {
            p->patch_work_off[(size_t)gi * (size_t)(maxL + 2) + (size_t)layer] = (int64_t)work.size();
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
            work.resize(base + slots * 8, -1);
            for (int x = 0; x < 8; x++) {
                const size_t b0 = m * (size_t)x / 8, b1 = m * (size_t)(x + 1) / 8;
                for (size_t t = b0; t < b1; t++) work[base + (t - b0) * 8 + (size_t)x] = items[t].second;
            }
        }
    p->patch_work_off.back() = (int64_t)work.size();
    VRT_HIP_TRY(hipMalloc((void **)&p->d_patch_work, sizeof(int32_t) * std::max<size_t>(work.size(), 1)));
    VRT_HIP_TRY(hipMemcpy(p->d_patch_work, work.data(), sizeof(int32_t) * work.size(), hipMemcpyHostToDevice));
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
    // workgroups per launch: enough to fill the chip a few times over with both direction streams running
    const int target_wgs = std::max(1, p->tune.patch_target);
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
    pa.duo = (!f32 && p->tune.patch_duo != 0 && p->patch_K == 1 && Q == 1 && npair >= 2) ? 1 : 0;
    if (pa.duo) Q = 2;
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
        const int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(steps_all, (target_wgs + items - 1) / items));
        pa.nsplit = nsplit;
        pa.ngrp = nsplit << lgS;
    }
    pa.stride = p->patch_cap + 1;
    pa.cap = p->patch_cap;
    pa.dbg = kDiag ? p->tune.debug_flags : 0;
    pa.work = p->d_patch_work + w0;
    pa.rec = p->d_patch_rec;
    pa.rec2 = p->d_patch_rec2;
    pa.e_pos = p->e_pos; pa.e_u1 = p->e_u1; pa.e_u2 = p->e_u2;
    pa.e_vis = p->e_vis; pa.e_loc = p->e_loc;
    pa.e_w1 = p->e_w1; pa.e_w2 = p->e_w2; pa.e_r1 = p->e_r1; pa.e_r2 = p->e_r2;
    const dim3 grid((unsigned)(pa.red.nred + (w1 - w0) * pa.ngrp));
    const size_t lds = (size_t)(pa.quad ? 2 : Q) * (size_t)pa.stride * sizeof(double2) + (size_t)pa.cap * (4 * sizeof(double) + 5 * sizeof(int32_t));
    const bool pipe = p->tune.patch_pipe == 1 || (p->tune.patch_pipe == 2 && f32);
    const int rc = f32 ? launch_mode<float>(ta.alpha_mode, p->patch_K, Q, p->patch_NT, grid, lds, st, pa, pipe)
                       : launch_mode<double>(ta.alpha_mode, p->patch_K, Q, p->patch_NT, grid, lds, st, pa, pipe);
    return rc;
}

}  // namespace vrt
