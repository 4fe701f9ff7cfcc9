// Layout changes between the caller's (nλ, n) site-major arrays and the storage order of the layer paths
// (LDS-tiled transposes), the boundary intensities, and the J reduction / combination.  Kernel templates,
// included by the translation unit that launches them (vrt_layers.hip).
#pragma once

#include "vrt_device.h"

namespace vrt {

// out[l][p] = in[order[p]][l]   (caller's (nλ, n) site-major rows -> wavelength-major sweep order)
// gridDim.z = 2: the second array pair (in2 -> out2: S and α of a direction in ONE launch) in the blocks with z = 1
template <typename T>
__global__ void __launch_bounds__(256)
k_to_sweep_order(int64_t n, int nlam, int64_t ld, int lb, const int32_t *__restrict__ order,
                 const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ in2 = nullptr, T *__restrict__ out2 = nullptr)
{
    if (blockIdx.z) {
        in = in2;
        out = out2;
    }
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
        // consecutive lanes write the consecutive pairs of a site's block, then the next site: whole lines
        // (the tile's 32 pairs start on a block boundary: blocks hold at most 32 pairs)
        const int lgB = log2_pairs(lb), npair = (nlam + 1) >> 1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = j * 256 + (int)threadIdx.x;
            const int pb = e & ((1 << lgB) - 1), px = (e >> lgB) & 63, bt = e >> (lgB + 6);
            const int c = 2 * ((bt << lgB) + pb), l = l0 + c;
            if (l < nlam && p0 + px < n) {
                typename Pair<T>::type v2;
                v2.x = tile[px][c];
                v2.y = l + 1 < nlam ? tile[px][c + 1] : (T)0;
                reinterpret_cast<typename Pair<T>::type *>(out)[pair_index(l >> 1, p0 + px, n, lgB, npair)] = v2;
            }
        }
    }
}

// The same for a handful of wavelengths (nλ <= 16, pair layout): the 64-wavelength tile above would run with most of its
// lanes idle (1 M sites x 7 λ: 100 µs; this form: one thread per (storage position, wavelength pair), a site's pairs in
// neighbouring lanes).  lgP = log2 of the lanes per site (>= pairs)
template <typename T>
__global__ void __launch_bounds__(256)
k_to_sweep_order_narrow(int64_t n, int nlam, int64_t ld, int lgB, int lgP, const int32_t *__restrict__ order,
                        const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ in2, T *__restrict__ out2)
{
    if (blockIdx.y) {
        in = in2;
        out = out2;
    }
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pos = e >> lgP;
    const int pr = (int)(e & ((1 << lgP) - 1)), npair = (nlam + 1) >> 1;
    if (pos >= n || pr >= npair) return;
    const T *row = in + (size_t)order[pos] * ld;
    typename Pair<T>::type v2;
    v2.x = row[2 * pr];
    v2.y = 2 * pr + 1 < nlam ? row[2 * pr + 1] : (T)0;
    reinterpret_cast<typename Pair<T>::type *>(out)[pair_index(pr, pos, n, lgB, npair)] = v2;
}

// Everything a chained step of one or two wavelength pairs needs before its launch, in ONE launch (fp64 planes, pairs not
// blocked): the narrow layout changes of S (and of a caller-layout alpha) of both directions, every intensity plane --
// the boundary layer's I_0 (irregular_ray_tracing.jl:33: ordered like perm[1:n1]; the boundary layer is storage positions
// [0, n1)), the never-visited site's zero (voronoi_utils.jl:266), the NaN pattern of the data-as-flag hand-off everywhere
// else -- and the zeroed control words of the chained launch.  As six launches of 5-9 us each these were a tenth of a
// C2 step (246 420 sites: 43 of 457 us); the fill is bandwidth, the rest is latency, and side by side they take the fill's time.
struct ChainPrep {
    int64_t n, ld;
    int nlam, npair, lgP;
    int njob;                              // layout changes: up to S and alpha of two directions
    unsigned tblocks;                      // blocks per layout change
    const double *tin[4];
    double *tout[4];
    const int32_t *torder[4];
    int A;
    unsigned fblocks;                      // blocks per angle of the intensity part: npair x ceil(n / 1024)
    double *I;
    int64_t n1[2];
    const int32_t *store[2], *rank[2];     // storage position -> site, site -> sweep position, per direction
    const double *I0[2];
    unsigned char down[kMaxAngles];
    uint32_t fill;
    uint32_t *ctrl;                        // may be NULL (the chained launch then zeroes its words itself)
    int nctrl;
};
__global__ void __launch_bounds__(256)
k_chain_prepare(ChainPrep cp)
{
    unsigned b = blockIdx.x;
    const unsigned tall = (unsigned)cp.njob * cp.tblocks;
    if (b < tall) {                        // ---- layout change (k_to_sweep_order_narrow) ----
        const int job = (int)(b / cp.tblocks);
        const int64_t e = (int64_t)(b - (unsigned)job * cp.tblocks) * 256 + threadIdx.x;
        const int64_t pos = e >> cp.lgP;
        const int pr = (int)(e & ((1 << cp.lgP) - 1));
        if (pos >= cp.n || pr >= cp.npair) return;
        const double *row = cp.tin[job] + (size_t)cp.torder[job][pos] * cp.ld;
        double2 v2;
        v2.x = row[2 * pr];
        v2.y = 2 * pr + 1 < cp.nlam ? row[2 * pr + 1] : 0.0;
        reinterpret_cast<double2 *>(cp.tout[job])[pair_index(pr, pos, cp.n, 0, cp.npair)] = v2;
        return;
    }
    b -= tall;
    const unsigned fall = (unsigned)cp.A * cp.fblocks;
    if (b < fall) {                        // ---- intensity planes ----
        const int a = (int)(b / cp.fblocks);
        const unsigned fb = b - (unsigned)a * cp.fblocks, pblocks = cp.fblocks / (unsigned)cp.npair;   // (block-uniform divisions)
        const int q = (int)(fb / pblocks);
        const int d = cp.down[a];
        const int64_t plane = (int64_t)cp.npair * cp.n;
#pragma unroll
        for (int k = 0; k < 4; k++) {          // 1024 positions per block: a quarter of the workgroups to dispatch
        const int64_t pos = (int64_t)(fb - (unsigned)q * pblocks) * 1024 + k * 256 + threadIdx.x;
        if (pos >= cp.n) return;
        const int64_t e = (int64_t)q * cp.n + pos;
        double2 v;
        if (pos < cp.n1[d]) {
            const double *I0 = cp.I0[d];
            const int64_t j = cp.rank[d][cp.store[d][pos]];
            v.x = I0 ? I0[(size_t)j * cp.nlam + 2 * q] : 0.0;
            v.y = (I0 && 2 * q + 1 < cp.nlam) ? I0[(size_t)j * cp.nlam + 2 * q + 1] : 0.0;
        } else if (pos == cp.n - 1) {
            v = make_double2(0.0, 0.0);
        } else {
            const double f = __hiloint2double((int)cp.fill, (int)cp.fill);
            v = make_double2(f, f);
        }
        // (streaming stores: 47 MB at C2's size that the chained launch reads with system-scope gathers anyway)
        double *dst = cp.I + 2 * ((size_t)a * (size_t)plane + (size_t)e);
        __builtin_nontemporal_store(v.x, dst);
        __builtin_nontemporal_store(v.y, dst + 1);
        }
        return;
    }
    b -= fall;
    const int w = (int)b * 256 + (int)threadIdx.x;
    if (cp.ctrl && w < cp.nctrl) cp.ctrl[w] = 0u;
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
    const int npair = (nlam + 1) >> 1;
    const int nl_pad = lb == 1 ? nlam : 2 * npair;
    T *Ia = I + (size_t)a * (size_t)nl_pad * (size_t)n;
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        // I_0 is ordered like perm[1:n1] (irregular_ray_tracing.jl:33); storage has its own order inside a layer
        if (l < nlam && p0 + tx < n1) {
            const int32_t pos = srank[order[p0 + tx]];
            Ia[sw_index(l, pos, n, lb, npair)] = tile[tx][c];
            if (l == nlam - 1 && nl_pad > nlam) Ia[sw_index(nlam, pos, n, lb, npair)] = (T)0;   // padding wavelength
        }
        // the never-visited site perm[n] (storage position n-1) keeps I = 0 (voronoi_utils.jl:266)
        // -- also on a single-layer grid, where no layer kernel ever runs
        if (blockIdx.x == 0 && tx == 0 && l < nl_pad) Ia[sw_index(l, n - 1, n, lb, npair)] = (T)0;
    }
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
    if (lb == 1) {
        for (int c = ty; c < 64; c += 4) {
            const int l = l0 + c;
            if (l < nlam && p < n) {
                double v = 0.0;
                if (Ju) v = (double)Ju[(size_t)l * n + p];
                if (Jdn) v = v + (double)Jdn[(size_t)l * n + pd];
                tile[tx][c] = (T)v;
            }
        }
    } else {                                   // pair accesses, a site's block of pairs by consecutive lanes
        typedef typename Pair<T>::type T2;
        __shared__ int32_t pds[64];
        if (threadIdx.x < 64) pds[threadIdx.x] = pd;
        __syncthreads();
        const int lgB = log2_pairs(lb), npair = (nlam + 1) >> 1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = j * 256 + (int)threadIdx.x;
            const int pb = e & ((1 << lgB) - 1), px = (e >> lgB) & 63, bt = e >> (lgB + 6);
            const int c = 2 * ((bt << lgB) + pb), l = l0 + c;
            if (l < nlam && p0 + px < n) {
                double2 v = make_double2(0.0, 0.0);
                if (Ju) v = to_d2(reinterpret_cast<const T2 *>(Ju)[pair_index(l >> 1, p0 + px, n, lgB, npair)]);
                if (Jdn) {
                    const double2 u = to_d2(reinterpret_cast<const T2 *>(Jdn)[pair_index(l >> 1, pds[px], n, lgB, npair)]);
                    v.x = v.x + u.x; v.y = v.y + u.y;
                }
                tile[px][c] = (T)v.x;
                tile[px][c + 1] = (T)v.y;
            }
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = p0 + r;
        if (q < n && l0 + tx < nlam) J[(size_t)order_up[q] * ldJ + l0 + tx] = tile[r][tx];
    }
}

// k_combine_J for a handful of wavelengths (nλ <= 16, pair layout): one thread per (up position, wavelength pair)
template <typename T>
__global__ void __launch_bounds__(256)
k_combine_J_narrow(int64_t n, int nlam, int64_t ldJ, int lgB, int lgP, const int32_t *__restrict__ order_up,
                   const int32_t *__restrict__ rank_down, const T *__restrict__ Ju, const T *__restrict__ Jdn, T *__restrict__ J)
{
    typedef typename Pair<T>::type T2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pos = e >> lgP;
    const int pr = (int)(e & ((1 << lgP) - 1)), npair = (nlam + 1) >> 1;
    if (pos >= n || pr >= npair) return;
    const int32_t site = order_up[pos];
    double2 v = make_double2(0.0, 0.0);
    if (Ju) v = to_d2(reinterpret_cast<const T2 *>(Ju)[pair_index(pr, pos, n, lgB, npair)]);
    if (Jdn) {
        const double2 u = to_d2(reinterpret_cast<const T2 *>(Jdn)[pair_index(pr, rank_down[site], n, lgB, npair)]);
        v.x = v.x + u.x; v.y = v.y + u.y;
    }
    T *row = J + (size_t)site * ldJ;
    row[2 * pr] = (T)v.x;
    if (2 * pr + 1 < nlam) row[2 * pr + 1] = (T)v.y;
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
        if (l < nlam && p0 + tx < n) tile[tx][c] = in ? in[sw_index(l, p0 + tx, n, lb, (nlam + 1) >> 1)] : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = p0 + r;
        if (q < n && l0 + tx < nlam) out[(size_t)order[q] * ld + l0 + tx] = tile[r][tx];
    }
}

}  // namespace vrt
