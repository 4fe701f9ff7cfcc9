// Plan-time tables of the layer paths in STORAGE order (layers contiguous, strips of rows inside: vrt_grid.cpp): the upwind
// table of an angle permuted from site order, the sorted thread assignment of the layer-step level kernels and
// the compact list of in-layer couplings.  One-time work per (plan, angle), launched by vrt_plan_create.
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

}  // namespace vrt
