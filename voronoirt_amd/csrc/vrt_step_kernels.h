// Kernel templates of the LAYER-STEP path ("steps": a chip-wide coefficient launch + a level launch per BFS
// layer), included by vrt_layers.hip.
#pragma once

#include "vrt_device.h"

namespace vrt {

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

}  // namespace vrt
