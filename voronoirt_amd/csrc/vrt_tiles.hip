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

#include "vrt_internal.h"

namespace vrt {

// ---- table in sweep order -----------------------------------------------------------------------
// t_u1/t_u2: sweep positions of the upwind sites; everything else copied from the site-order table.
__global__ void __launch_bounds__(256)
k_permute_table(int64_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ rank,
                const int32_t *__restrict__ up1, const int32_t *__restrict__ up2,
                const double *__restrict__ w1, const double *__restrict__ w2,
                const double *__restrict__ r1, const double *__restrict__ r2,
                const uint32_t *__restrict__ vis, int32_t *__restrict__ t_u1,
                int32_t *__restrict__ t_u2, double *__restrict__ t_w1, double *__restrict__ t_w2,
                double *__restrict__ t_r1, double *__restrict__ t_r2, uint32_t *__restrict__ t_vis)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t s = order[p];
    const int32_t a = up1[s], b = up2[s];
    t_u1[p] = a >= 0 ? rank[a] : -1;
    t_u2[p] = b >= 0 ? rank[b] : -1;
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
                       dir.d_order, dir.d_rank, p->d_up1 + o, p->d_up2 + o, p->d_w1 + o, p->d_w2 + o,
                       p->d_r1 + o, p->d_r2 + o, d_vis_site, p->t_u1 + o, p->t_u2 + o, p->t_w1 + o,
                       p->t_w2 + o, p->t_r1 + o, p->t_r2 + o, p->t_vis + o);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// ---- layout changes -----------------------------------------------------------------------------
// out[l][p] = in[order[p]][l]   (caller's (nλ, n) site-major rows -> wavelength-major sweep order)
__global__ void __launch_bounds__(256)
k_to_sweep_order(int64_t n, int nlam, int64_t ld, const int32_t *__restrict__ order,
                 const double *__restrict__ in, double *__restrict__ out)
{
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        const int64_t p = p0 + r;
        if (p < n && l0 + tx < nlam) tile[r][tx] = in[(size_t)order[p] * ld + l0 + tx];
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        if (l < nlam && p0 + tx < n) out[(size_t)l * n + p0 + tx] = tile[tx][c];
    }
}

// out[p] = in[order[p]]   (per-site vector, e.g. wavelength-independent α)
__global__ void __launch_bounds__(256)
k_gather_vec(int64_t n, const int32_t *__restrict__ order, const double *__restrict__ in,
             double *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[order[p]];
}

// I[a][l][p] = I0[p][l] for p < n1 (boundary layer, already in sweep order), blockIdx.z = angle slot
__global__ void __launch_bounds__(256)
k_boundary_sweep_order(int64_t n, int nlam, int64_t n1, const int32_t *__restrict__ angles,
                       const double *__restrict__ I0, double *__restrict__ I)
{
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    const int a = angles[blockIdx.z];
    for (int r = ty; r < 64; r += 4) {
        const int64_t p = p0 + r;
        if (p < n1 && l0 + tx < nlam) tile[r][tx] = I0 ? I0[(size_t)p * nlam + l0 + tx] : 0.0;
    }
    __syncthreads();
    double *Ia = I + (size_t)a * (size_t)nlam * (size_t)n;
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        if (l < nlam && p0 + tx < n1) Ia[(size_t)l * n + p0 + tx] = tile[tx][c];
    }
}

// ---- the solver ---------------------------------------------------------------------------------
__device__ __forceinline__ void lin_weights(double dtau, double &a, double &b, double &e)
{
    if (dtau < 5e-4) {                       // functions.jl:484-500
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

struct TileArgs {
    int64_t n;
    int nlam;
    int A;
    int alpha_mode;
    int max_layers;                 // stride of nlev
    const int32_t *angle_sorted;    // heaviest angle first (dispatch order = task order)
    const int32_t *angle_dir;       // [A] 0 = up, 1 = down
    const int32_t *lay[2];          // per direction: 0-based [lo, hi) boundaries, lay[d][L+1]
    int nlayers[2];                 // number of BFS layers per direction
    const int32_t *nlev;            // [A][max_layers + 1] in-layer level counts (index = layer)
    const int32_t *t_u1, *t_u2;     // [A][n] sweep positions of the upwinds
    const double *t_w1, *t_w2, *t_r1, *t_r2;
    const uint32_t *t_vis;
    const double *S[2];             // per direction [nlam][n]
    const double *alpha[2];         // SITE: [n]; SITE_LAM: [nlam][n] per direction
    const double *alpha_angle;      // ANGLE: [A][nlam][n]
    double *I;                      // [A][nlam][n]
};

template <int K>
__global__ void __launch_bounds__(1024)
k_sweep_tiles(TileArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];   // I of the current layer
    const int T = 1024;
    const int tid = threadIdx.x;
    const int task = blockIdx.x;
    const int a = ta.angle_sorted[task / ta.nlam];
    const int l = task % ta.nlam;
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

    for (int layer = 2; layer <= L; layer++) {          // irregular_ray_tracing.jl:37
        const int lo = lay[layer - 1], hi = lay[layer];  // hi of the last layer = n-1: perm[n] is never visited
        const int cnt = hi - lo;
        double c[K], g1[K], g2[K];
        int loc1[K], loc2[K];
        uint32_t vis[K];
        // ---- phase 1: coefficients of every site of the layer (global reads, registers) ----
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = tid + k * T;
            c[k] = 0.0; g1[k] = 0.0; g2[k] = 0.0; loc1[k] = 0; loc2[k] = 0; vis[k] = 0;
            if (slot < cnt) {
                const int p = lo + slot;
                const int u1 = tu1[p], u2 = tu2[p];
                const double w1 = tw1[p], w2 = tw2[p], r1 = tr1[p], r2 = tr2[p];
                vis[k] = tvis[p];
                const double S_c = S[p], a_c = Al[p];
                const double S_1 = S[u1], a_1 = Al[u1], S_2 = S[u2], a_2 = Al[u2];
                double ca, cb, ce;
                lin_weights(r1 * (a_c + a_1) / 2.0, ca, cb, ce);       // trapezoidal, functions.jl:393
                double t1;
                if (u1 < lo) {                                           // earlier layer: final value
                    t1 = ((ce * I[u1] + ca * S_1) + cb * S_c) * w1;
                } else {
                    t1 = (ca * S_1 + cb * S_c) * w1;
                    if (u1 < hi) { g1[k] = ce * w1; loc1[k] = u1 - lo; } // in-layer: coupled through the tile
                }                                                        // later layer / perm[n]: reads 0
                lin_weights(r2 * (a_c + a_2) / 2.0, ca, cb, ce);
                double t2;
                if (u2 < lo) {
                    t2 = ((ce * I[u2] + ca * S_2) + cb * S_c) * w2;
                } else {
                    t2 = (ca * S_2 + cb * S_c) * w2;
                    if (u2 < hi) { g2[k] = ce * w2; loc2[k] = u2 - lo; }
                }
                c[k] = t1 + t2;
                tile[slot] = 0.0;                                        // I = zero(S), :23
            }
        }
        __syncthreads();
        // ---- phase 2: the layer's Gauss-Seidel levels on the LDS tile ------------------------
        const int nl = nlev[layer];
        for (int t = 1; t <= nl; t++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t v = vis[k];
                const bool hit = ((v & 0xFFu) == (uint32_t)t) | (((v >> 8) & 0xFFu) == (uint32_t)t) |
                                 (((v >> 16) & 0xFFu) == (uint32_t)t) | ((v >> 24) == (uint32_t)t);
                if (hit) tile[tid + k * T] = c[k] + g1[k] * tile[loc1[k]] + g2[k] * tile[loc2[k]];
            }
            __syncthreads();
        }
        // ---- phase 3: the layer is final -> global, visible to this workgroup's next layers ---
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int slot = tid + k * T;
            if (slot < cnt) I[lo + slot] = tile[slot];
        }
        __syncthreads();
    }
    if (tid == 0) I[n - 1] = 0.0;   // the never-visited site perm[n] keeps I = 0 (voronoi_utils.jl:266)
}

// J_d[l][p] = Σ_{angles of direction d} w_a I_a[l][p], reference's angle order within the direction
struct DirWeights {
    double w[kMaxAngles];
    int32_t idx[kMaxAngles];
    int count;
};

__global__ void __launch_bounds__(256)
k_reduce_dir(int64_t total, int64_t stride_angle, DirWeights dw, const double *__restrict__ I,
             double *__restrict__ Jd)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    double acc = 0.0;
    for (int j = 0; j < dw.count; j++) acc += dw.w[j] * I[(size_t)dw.idx[j] * stride_angle + t];
    Jd[t] = acc;
}

// J[site][l] = J_up[l][rank_up[site]] + J_down[l][rank_down[site]], walking sites in up order so
// the J_up reads are coalesced and the J_down reads are piecewise contiguous on stratified grids.
__global__ void __launch_bounds__(256)
k_combine_J(int64_t n, int nlam, int64_t ldJ, const int32_t *__restrict__ order_up,
            const int32_t *__restrict__ rank_down, const double *__restrict__ Ju,
            const double *__restrict__ Jdn, double *__restrict__ J)
{
    __shared__ double tile[64][65];
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
            if (Ju) v = Ju[(size_t)l * n + p];
            if (Jdn) v = v + Jdn[(size_t)l * n + pd];
            tile[tx][c] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t q = p0 + r;
        if (q < n && l0 + tx < nlam) J[(size_t)order_up[q] * ldJ + l0 + tx] = tile[r][tx];
    }
}

// out[order[p]][l] = in[l][p]  (sweep order, wavelength-major -> caller's site-major rows)
__global__ void __launch_bounds__(256)
k_from_sweep_order(int64_t n, int nlam, int64_t ld, const int32_t *__restrict__ order,
                   const double *__restrict__ in, double *__restrict__ out)
{
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int l0 = blockIdx.y * 64;
    for (int c = ty; c < 64; c += 4) {
        const int l = l0 + c;
        if (l < nlam && p0 + tx < n) tile[tx][c] = in ? in[(size_t)l * n + p0 + tx] : 0.0;
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

int execute_tiles(vrt_plan *p, int64_t nlam, int64_t ld, const double *dS, const double *dalpha,
                  int alpha_mode, const double *dI0_up, const double *dI0_down,
                  const double *weights_user, double *dJ, double *dI_out, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int A = p->A;
    const size_t plane = (size_t)nlam * (size_t)n;
    int rc;
    if ((rc = ensure_dev(p->d_I, p->I_cap, (size_t)std::max(1, A) * plane))) return rc;
    const bool use_dir[2] = {p->n_up > 0, p->n_down > 0};
    for (int d = 0; d < 2; d++)
        if (use_dir[d] && (rc = ensure_dev(p->ws_S[d], p->ws_S_cap[d], plane))) return rc;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    TileArgs ta;
    ta.n = n;
    ta.nlam = (int)nlam;
    ta.A = A;
    ta.alpha_mode = alpha_mode;
    ta.max_layers = p->tile_max_layers;
    ta.angle_sorted = p->d_angle_sorted;
    ta.angle_dir = p->d_angle_dir;
    ta.nlev = p->d_nlev;
    ta.t_u1 = p->t_u1; ta.t_u2 = p->t_u2;
    ta.t_w1 = p->t_w1; ta.t_w2 = p->t_w2; ta.t_r1 = p->t_r1; ta.t_r2 = p->t_r2;
    ta.t_vis = p->t_vis;
    ta.alpha_angle = nullptr;
    ta.I = p->d_I;
    for (int d = 0; d < 2; d++) {
        const Direction &dir = d == 0 ? g->up : g->down;
        ta.lay[d] = dir.d_lay;
        ta.nlayers[d] = (int)dir.reduced.size() - 1;
        ta.S[d] = nullptr;
        ta.alpha[d] = nullptr;
        if (!use_dir[d]) continue;
        hipLaunchKernelGGL(k_to_sweep_order, tgrid, dim3(256), 0, st, n, (int)nlam, ld, dir.d_order, dS,
                           p->ws_S[d]);
        ta.S[d] = p->ws_S[d];
        if (alpha_mode == VRT_ALPHA_SITE) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], (size_t)n))) return rc;
            hipLaunchKernelGGL(k_gather_vec, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                               dir.d_order, dalpha, p->ws_A[d]);
            ta.alpha[d] = p->ws_A[d];
        } else if (alpha_mode == VRT_ALPHA_SITE_LAM) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], plane))) return rc;
            hipLaunchKernelGGL(k_to_sweep_order, tgrid, dim3(256), 0, st, n, (int)nlam, ld, dir.d_order,
                               dalpha, p->ws_A[d]);
            ta.alpha[d] = p->ws_A[d];
        }
        const int cnt = d == 0 ? p->n_up : p->n_down;
        if (dir.n1 > 0) {
            const dim3 bgrid((unsigned)((dir.n1 + 63) / 64), (unsigned)((nlam + 63) / 64), (unsigned)cnt);
            hipLaunchKernelGGL(k_boundary_sweep_order, bgrid, dim3(256), 0, st, n, (int)nlam, dir.n1,
                               d == 0 ? p->d_angles_up : p->d_angles_down, d == 0 ? dI0_up : dI0_down,
                               p->d_I);
        }
    }
    if (alpha_mode == VRT_ALPHA_ANGLE_SITE_LAM) {
        if ((rc = ensure_dev(p->ws_AA, p->ws_AA_cap, (size_t)A * plane))) return rc;
        for (int a = 0; a < A; a++) {
            const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
            hipLaunchKernelGGL(k_to_sweep_order, tgrid, dim3(256), 0, st, n, (int)nlam, ld, dir.d_order,
                               dalpha + (size_t)a * (size_t)n * (size_t)ld, p->ws_AA + (size_t)a * plane);
        }
        ta.alpha_angle = p->ws_AA;
    }
    VRT_HIP_TRY(hipGetLastError());

    VRT_HIP_TRY(hipEventRecord(p->ev0, st));
    if (A > 0) {
        const size_t lds = (size_t)std::max<int64_t>(p->tile_max_layer_size, 1) * sizeof(double);
        const dim3 grid((unsigned)((size_t)A * (size_t)nlam));
        switch (p->tile_K) {
        case 1: hipLaunchKernelGGL(k_sweep_tiles<1>, grid, dim3(1024), lds, st, ta); break;
        case 2: hipLaunchKernelGGL(k_sweep_tiles<2>, grid, dim3(1024), lds, st, ta); break;
        case 4: hipLaunchKernelGGL(k_sweep_tiles<4>, grid, dim3(1024), lds, st, ta); break;
        default: hipLaunchKernelGGL(k_sweep_tiles<8>, grid, dim3(1024), lds, st, ta); break;
        }
        VRT_HIP_TRY(hipGetLastError());
    }
    VRT_HIP_TRY(hipEventRecord(p->ev1, st));
    p->ev_valid = true;
    p->last_launches = 1;

    if (dJ) {
        double *Jd[2] = {nullptr, nullptr};
        for (int d = 0; d < 2; d++) {
            if (!use_dir[d]) continue;
            if ((rc = ensure_dev(p->ws_J[d], p->ws_J_cap[d], plane))) return rc;
            DirWeights dw;
            dw.count = 0;
            for (int a = 0; a < A; a++)
                if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) {
                    dw.w[dw.count] = weights_user[p->user_of_active[(size_t)a]];
                    dw.idx[dw.count] = a;
                    dw.count++;
                }
            hipLaunchKernelGGL(k_reduce_dir, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st,
                               (int64_t)plane, (int64_t)plane, dw, p->d_I, p->ws_J[d]);
            Jd[d] = p->ws_J[d];
        }
        hipLaunchKernelGGL(k_combine_J, tgrid, dim3(256), 0, st, n, (int)nlam, ld, g->up.d_order,
                           g->down.d_rank, Jd[0], Jd[1], dJ);
        VRT_HIP_TRY(hipGetLastError());
    }
    if (dI_out) {
        std::vector<int> active_of_user((size_t)p->n_angles_user, -1);
        for (int a = 0; a < A; a++) active_of_user[(size_t)p->user_of_active[(size_t)a]] = a;
        for (int64_t u = 0; u < p->n_angles_user; u++) {
            const int a = active_of_user[(size_t)u];
            const Direction &dir = (a >= 0 && p->dir_of_active[(size_t)a] < 0) ? g->down : g->up;
            hipLaunchKernelGGL(k_from_sweep_order, tgrid, dim3(256), 0, st, n, (int)nlam, ld, dir.d_order,
                               a >= 0 ? p->d_I + (size_t)a * plane : nullptr,
                               dI_out + (size_t)u * (size_t)n * (size_t)ld);
        }
        VRT_HIP_TRY(hipGetLastError());
    }
    return VRT_OK;
}

}  // namespace vrt
