// Device helpers and kernel-argument records shared by the HIP translation units of the layer
// solvers (vrt_layers.hip with vrt_tile_kernels.h / vrt_step_kernels.h / vrt_layout_kernels.h, vrt_patch.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "vrt_internal.h"

namespace vrt {

// Timing diagnostics that switch pieces of the memory traffic off (and give WRONG results) exist
// only in the -DVRT_DIAG build (voronoirt_amd/libvrt_hip_diag.so, used by tools/flags_sweep.sh):
// in the product library every such branch folds away at compile time and no environment
// variable can change a result.
#ifdef VRT_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

constexpr uint32_t kNoSlot = 0xFFFFu;   // t_loc entry of an upwind outside the site's own layer (layers <= 8192 sites)

// ---- layout changes -----------------------------------------------------------------------------
// Storage-order arrays are wavelength-major.  lb = 1: plain planes [λ][pos] (persistent tile kernel).
// lb = 2 B >= 2: wavelength PAIRS (one 16-byte access = two wavelengths), B pairs of a site side by side:
// pair q of storage position p lives at pair element  k0 n + (p << lw) + (q - k0),  where [k0, k0 + 2^lw) is
// the BLOCK of pair q: blocks of B pairs, then -- when B does not divide the pair count -- one block per set bit
// of the remainder, widest first (26 pairs, B = 8: 8, 8, 8, 2), so that no plane is padded and a site's record
// never holds bytes nobody reads.  B = 1 ([λ/2][pos][2]) is what the layer-step kernels read; B = 8 makes a
// gathered 128-byte line ONE site's eight pairs, which the eight sibling workgroups of a patch (one pair of the
// block each) use whole (vrt_patch.hip).  Planes hold ceil(nλ / 2) pairs.
__host__ __device__ __forceinline__ void pair_block_at(int q, int npair, int lgB, int &k0, int &lw)
{
    const int full = npair >> lgB << lgB;
    if (q < full) { k0 = q >> lgB << lgB; lw = lgB; return; }
    int k = full;
    const int r = npair - full;
    for (int bit = lgB - 1; bit >= 0; bit--)
        if (r & (1 << bit)) {
            if (q < k + (1 << bit)) { k0 = k; lw = bit; return; }
            k += 1 << bit;
        }
    k0 = k; lw = 0;                                   // q >= npair: not a pair of the plane
}
// block number k -> its first pair and log2 width (k0 = npair when there is no such block)
__host__ __device__ __forceinline__ void pair_block_of(int k, int npair, int lgB, int &k0, int &lw)
{
    const int nfull = npair >> lgB;
    if (k < nfull) { k0 = k << lgB; lw = lgB; return; }
    int j = k - nfull;
    k0 = nfull << lgB;
    const int r = npair - k0;
    for (int bit = lgB - 1; bit >= 0; bit--)
        if (r & (1 << bit)) {
            if (j == 0) { lw = bit; return; }
            j--;
            k0 += 1 << bit;
        }
    lw = 0;
}
__host__ __device__ __forceinline__ int pair_block_count(int npair, int lgB)
{
    return (npair >> lgB) + __builtin_popcount((unsigned)(npair - (npair >> lgB << lgB)));
}
__host__ __device__ __forceinline__ size_t pair_index(int q, int64_t p, int64_t n, int lgB, int npair)
{
    int k0, lw;
    pair_block_at(q, npair, lgB, k0, lw);
    return (size_t)k0 * (size_t)n + ((size_t)p << lw) + (size_t)(q - k0);
}
__host__ __device__ __forceinline__ int log2_pairs(int lb)     // lb = 2 B wavelengths per block -> log2 B
{
    int lg = 0;
    while ((2 << lg) < lb) lg++;
    return lg;
}
// scalar element (wavelength l, storage position p); npair = ceil(nλ / 2) (unused for lb = 1)
__device__ __forceinline__ size_t sw_index(int l, int64_t p, int64_t n, int lb, int npair)
{
    return lb == 1 ? (size_t)l * (size_t)n + (size_t)p
                   : (pair_index(l >> 1, p, n, log2_pairs(lb), npair) << 1) + (size_t)(l & 1);
}

// storage types: T = double, or float for the fp32 VALUE path (BASELINE config C5: S, α, I, J held as
// float, all arithmetic fp64); a wavelength pair is one 16-byte (double2) or 8-byte (float2) access
template <typename T> struct Pair;
template <> struct Pair<double> { typedef double2 type; };
template <> struct Pair<float> { typedef float2 type; };
__device__ __forceinline__ double2 to_d2(double2 v) { return v; }
__device__ __forceinline__ double2 to_d2(float2 v) { return make_double2((double)v.x, (double)v.y); }
template <typename T> __device__ __forceinline__ typename Pair<T>::type from_d2(double2 v);
template <> __device__ __forceinline__ double2 from_d2<double>(double2 v) { return v; }
template <> __device__ __forceinline__ float2 from_d2<float>(double2 v) { return make_float2((float)v.x, (float)v.y); }


// ---- the solver ---------------------------------------------------------------------------------
// linear_weights (functions.jl:484-500) with the arithmetic trimmed for the ALU-bound phase 1:
// one Newton-refined reciprocal shared by the thick and the exponential branch, the Taylor
// branch's /3 and /6 as multiplications, and exp(-x) for the only range it is needed in
// (5e-4 <= x <= 50: no overflow, underflow, NaN or subnormal handling).  Each piece is accurate
// to ~1 ulp; results differ from the oracle's libm at the 1e-16 level (contract: 1e-10).
__device__ __forceinline__ double exp_neg(double x)       // exp(-x), 5e-4 <= x <= 50
{
    const double t = -x;
    const double kf = rint(t * 1.4426950408889634074);    // k = round(t / ln 2), |k| <= 73
    double r = fma(-kf, 6.93147180369123816490e-01, t);   // Cody-Waite: ln2 = hi + lo
    r = fma(-kf, 1.90821492927058770002e-10, r);           // |r| <= 0.3466
    double p = 1.0 / 6227020800.0;                         // Taylor to r^13/13!: remainder < 4e-18
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
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

// exp(-x) for 0 <= x (the patch kernels call it for 5e-4 <= x <= 50), table-driven: -x = N ln2/32 + r with |r| <= ln2/64, exp(-x) = 2^(N >> 5) T[N & 31] p(r),
// T[j] = 2^(j/32) from a 32-entry LDS table (one 8-byte read per evaluation, conflict-free: the 32 entries cover the
// 64 banks once) and p the degree-5 Taylor polynomial (remainder r^6/720 < 2.3e-15).  Against the degree-10 polynomial
// on |r| <= ln2/2 it replaces: four fused multiply-adds fewer per evaluation, and five of its eleven non-inline fp64
// constants -- ten scalar registers of a kernel that is short of exactly those.  Relative error ~3e-15 (contract 1e-10).
static __device__ const double kExp2_32[32] = {
    1.0, 1.0218971486541166, 1.0442737824274138, 1.0671404006768237, 1.0905077326652577, 1.1143867425958924,
    1.1387886347566916, 1.1637248587775775, 1.189207115002721, 1.215247359980469, 1.241857812073484, 1.2690509571917332,
    1.2968395546510096, 1.3252366431597413, 1.3542555469368927, 1.383909881963832, 1.4142135623730951, 1.4451808069770467,
    1.4768261459394993, 1.5091644275934228, 1.5422108254079407, 1.5759808451078865, 1.6104903319492543, 1.645755478153965,
    1.681792830507429, 1.718619298122478, 1.7562521603732995, 1.7947090750031072, 1.8340080864093424, 1.8741676341103,
    1.9152065613971474, 1.9571441241754002};
__device__ __forceinline__ double *exp2_table()
{
    __shared__ double t[32];
    return t;
}
// every kernel that evaluates it: fill the table, then a barrier before the first evaluation
__device__ __forceinline__ void exp2_table_fill()
{
    if (threadIdx.x < 32) exp2_table()[threadIdx.x] = kExp2_32[threadIdx.x];
}
__device__ __forceinline__ double exp_neg_tab(double x)          // exp(-x), 0 <= x <= 700
{
    const double t = -x;
    const double nf = rint(t * 46.16624130844683);            // N = round(t 32 / ln 2), |N| <= 2309
    double r = fma(-nf, 0.021660849335603416, t);             // Cody-Waite: ln2/32 = hi (29 bits) + lo
    r = fma(-nf, 5.689487495325457e-11, r);                   // |r| <= 0.01084
    const int N = (int)nf;
    double p = 1.0 / 120.0;
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p * exp2_table()[N & 31], N >> 5);
}

__device__ __forceinline__ void lin_weights(double dtau, double &a, double &b, double &e)
{
    // reciprocal of dtau (only consumed when dtau >= 5e-4): hardware estimate + 2 Newton steps
    double rc = __builtin_amdgcn_rcp(dtau);
    rc = fma(fma(-dtau, rc, 1.0), rc, rc);
    rc = fma(fma(-dtau, rc, 1.0), rc, rc);
    const double ee = exp_neg(fmin(fmax(dtau, 5e-4), 50.0));
    if (dtau < 5e-4) {
        e = 1.0 - dtau + 0.5 * (dtau * dtau);
        a = dtau * (0.5 - dtau * (1.0 / 3.0));
        b = dtau * (0.5 - dtau * (1.0 / 6.0));
    } else if (dtau > 50.0) {
        e = 0.0;
        a = rc;
        b = 1.0 - a;
    } else {
        e = ee;
        a = (1.0 - e) * rc - e;
        b = 1.0 - a - e;
    }
}

// 32-bit byte offsets from a wave-uniform base: lets the compiler use the saddr + voffset form of
// global_load (one VGPR per address instead of a 64-bit pair) -- the phase-1 batches are
// register-bound.  Planes are n * 8 bytes < 4 GiB.
__device__ __forceinline__ double ldd(const double *base, unsigned idx)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (size_t)(idx << 3));
}
__device__ __forceinline__ int ldi(const int32_t *base, unsigned idx)
{
    return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(base) + (size_t)(idx << 2));
}
__device__ __forceinline__ uint32_t ldu(const uint32_t *base, unsigned idx)
{
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + (size_t)(idx << 2));
}

struct TileArgs {
    int64_t n;
    int nlam;
    int A;
    int alpha_mode;
    int max_layers;                 // stride of nlev
    int tile_stride;                // doubles per LDS array (>= largest layer)
    const int32_t *task_map;        // block -> angle | wavelength << 8 (XCD-aware, see build_task_map)
    const int32_t *angle_dir;       // [A] 0 = up, 1 = down
    const int32_t *lay[2];          // per direction: 0-based [lo, hi) boundaries, lay[d][L+1]
    int nlayers[2];                 // number of BFS layers per direction
    const int32_t *nlev;            // [A][max_layers + 1] in-layer level counts (index = layer)
    const int32_t *t_u1, *t_u2;     // [A][n] sweep positions of the upwinds
    const double *t_w1, *t_w2, *t_r1, *t_r2;
    const uint32_t *t_vis;
    const uint32_t *t_loc;          // [A][n] packed in-layer tile slots of the two upwinds
    const int32_t *t_self;          // [A][n] sorted thread order of k_step_levels (build_sorted_slots)
    const uint32_t *t_vis_s, *t_loc_s;
    const uint32_t *t_gpos;         // [A][n] compact in-layer coupling list positions (k_gpos)
    const double *S[2];             // per direction [nlam][n]
    const double *alpha[2];         // SITE: [n]; SITE_LAM: [nlam][n] per direction
    const double *alpha_angle;      // ANGLE: [A][nlam][n]
    double *I;                      // [A][nlam][n]
    long long *dbg;                 // diagnostics (VRT_TILE_DEBUG=1): per task phase cycles, else NULL
};

struct DirWeights {
    double w[kMaxAngles];
    int32_t idx[kMaxAngles];
    int count;
};

struct StepArgs {
    TileArgs ta;              // S, alpha, I in pair layout; ta.nlam = the caller's wavelength count
    int npair;                // ceil(nlam / 2)
    int layer;                // 1-based BFS layer being solved
    int cg_stride;            // slots (double2 each) per (angle, wavelength pair) in the coefficient buffers
    double2 *cg_c;            // constant terms, cg_stride per (angle, pair)
    double2 *cg_g;            // in-layer couplings, compact list (t_gpos), 2 cg_stride per (angle, pair)
    const int32_t *angle_list;      // the angles this launch works on (one stream's share)
    int n_list;
    int pairs_per_thread;     // wavelength pairs one k_step_coeffs thread loops over
    int chunks;               // 256-slot chunks per layer (k_step_coeffs grid.x / 1)
    int xcd_map;              // 0: plain grid; 1, 2: contiguous chunk ranges per XCD (2: angle fastest)
    const int32_t *level_map; // level kernels: block -> task (index into angle_list x wavelengths), -1 = padding; NULL: identity
    const int32_t *t_rank_s;  // single-wavelength level kernel: storage position -> sorted index
    const uint32_t *t_loc_ss; //   and the upwind tile slots in sorted terms
    int debug_skip_levels;    // diagnostics only (VRT_DEBUG_SKIP_LEVELS=1): wrong results
    int debug_flags;          // diagnostics only (VRT_DEBUG_FLAGS bit mask): wrong results, see execute_tiles
};

__device__ __forceinline__ double2 ld2(const double2 *base, unsigned idx)
{
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + ((size_t)idx << 4));
}
__device__ __forceinline__ double2 ld2(const float2 *base, unsigned idx)     // fp32 storage, fp64 arithmetic
{
    return to_d2(*reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + ((size_t)idx << 3)));
}

// one upwind's share of the visit: t = ((e I_u + a S_u) + b S_c) w  (I_u dropped unless the upwind
// lies in an earlier layer), g = e w if the upwind lies in the site's own layer
__device__ __forceinline__ void upwind_term(double r, double w, double a_c, double a_u, double S_c,
                                            double S_u, double I_u, bool early, bool inl, double &t,
                                            double &g, bool cheap = false)
{
    double ca, cb, ce;
    if (cheap) { ca = a_c; cb = a_u; ce = r; }                     // diagnostics: no linear_weights
    else
    lin_weights(r * (a_c + a_u) / 2.0, ca, cb, ce);                // trapezoidal, functions.jl:393
    t = early ? ((ce * I_u + ca * S_u) + cb * S_c) * w : (ca * S_u + cb * S_c) * w;
    g = inl ? ce * w : 0.0;
}

// J reduction riding along the fused patch launches (vrt_patch.hip), lagged by one layer
struct PatchReduce {
    int nred = 0;                    // blocks in the reduction role (a multiple of 8: the XCD dealing of the rest stays aligned)
    int nblk[2] = {0, 0};            // of which for range 0 / 1
    int lo[2] = {0, 0}, hi[2] = {0, 0};   // storage position ranges
    int count[2] = {0, 0};           // angles of the range's direction
    int ppb = 4;                     // wavelength pairs per block
    void *Jd[2] = {nullptr, nullptr};     // J_dir planes [npair][n] (pairs of T)
    double w[kMaxAngles];            // quadrature weight per ACTIVE angle
    int32_t angles[2][kMaxAngles];   // active angles of the range's direction, in the reference's order
};


}  // namespace vrt
