// Host side of one execute on the layer paths: "tiles" (one persistent workgroup per (angle, wavelength)),
// "steps" (two launches per BFS layer) and "patches" (one fused launch per BFS layer, vrt_patch.hip).  They share
// the storage-order layouts (layers contiguous, wavelength pairs side by side), the layout-change kernels and the
// J reduction; the kernels live in vrt_layout_kernels.h, vrt_tile_kernels.h, vrt_step_kernels.h and vrt_patch.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vrt_device.h"
#include "vrt_internal.h"

#include "vrt_layout_kernels.h"
#include "vrt_step_kernels.h"
#include "vrt_tile_kernels.h"

namespace vrt {

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
template <typename T>
static int alpha_to_native_t(vrt_plan *p, int64_t nlam, int64_t ld, const T *dalpha, T *out, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int64_t nl_pad = (nlam + 1) / 2 * 2;
    const size_t plane = (size_t)nl_pad * (size_t)n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    for (int a = 0; a < p->A; a++) {
        const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
        hipLaunchKernelGGL(k_to_sweep_order<T>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2 << native_lg(p, sizeof(T) == 4), dir.d_store,
                           dalpha + (size_t)p->user_of_active[(size_t)a] * (size_t)n * (size_t)ld,
                           out + (size_t)a * plane, (const T *)nullptr, (T *)nullptr);
    }
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int alpha_to_native(vrt_plan *p, int64_t nlam, int64_t ld, const void *dalpha, void *out, hipStream_t st, bool f32)
{
    return f32 ? alpha_to_native_t<float>(p, nlam, ld, (const float *)dalpha, (float *)out, st)
               : alpha_to_native_t<double>(p, nlam, ld, (const double *)dalpha, (double *)out, st);
}

// ---- sweep-order ("native") S and J: per sweep direction a plane set [pairs][n][2] in that direction's storage order ------
// (the layout the layer paths read S from and reduce J into; what vrt_plan_execute_native_dev takes and returns in place)
int native_planes_ok(const vrt_plan *p, bool f32)
{
    if (f32) {          // float planes: the patch path only, in its pair blocks (vrt_plan_native_pair_block_f32)
        if (!p->patch_ok) return fail(VRT_EINVAL, "sweep-order float S and J need the patch path (at most 4 visits per site, 255 levels per layer, 512 entries per cone)");
        return VRT_OK;
    }
    if (!(p->patch_ok || p->tile_ok)) return fail(VRT_EINVAL, "sweep-order S and J need a layer path (at most 4 visits per site and 255 levels per layer)");
    if (native_lg(p, false) != 0) return fail(VRT_EINVAL, "sweep-order S and J need one wavelength pair per block (VRT_PAIR_BLOCK=1)");
    return VRT_OK;
}

// the float forms of the three layout helpers below (pair blocks of the plan's float layout)
int planes_to_native_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *din, float *out_up, float *out_down, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int lb = 2 << native_lg(p, true);
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    for (int d = 0; d < 2; d++) {
        float *out = d == 0 ? out_up : out_down;
        if (!out) continue;
        const Direction &dir = d == 0 ? g->up : g->down;
        hipLaunchKernelGGL(k_to_sweep_order<float>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, lb, dir.d_store, din, out,
                           (const float *)nullptr, (float *)nullptr);
    }
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}
int plane_from_native_f32(vrt_plan *p, int dir_index, int64_t nlam, int64_t ld, const float *din, float *dout, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    const Direction &dir = dir_index == 0 ? g->up : g->down;
    hipLaunchKernelGGL(k_from_sweep_order<float>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2 << native_lg(p, true), dir.d_store, din, dout);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}
int J_from_native_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dJ_up, const float *dJ_down, float *dJ, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    hipLaunchKernelGGL(k_combine_J<float>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2 << native_lg(p, true), g->up.d_store, g->down.d_srank, dJ_up, dJ_down, dJ);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// caller's (n, ld) rows -> the planes of both directions (either output may be NULL)
int planes_to_native(vrt_plan *p, int64_t nlam, int64_t ld, const double *din, double *out_up, double *out_down, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    for (int d = 0; d < 2; d++) {
        double *out = d == 0 ? out_up : out_down;
        if (!out) continue;
        const Direction &dir = d == 0 ? g->up : g->down;
        hipLaunchKernelGGL(k_to_sweep_order<double>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2, dir.d_store, din, out,
                           (const double *)nullptr, (double *)nullptr);
    }
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// the planes of ONE direction -> the caller's (n, ld) rows
int plane_from_native(vrt_plan *p, int dir_index, int64_t nlam, int64_t ld, const double *din, double *dout, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    const Direction &dir = dir_index == 0 ? g->up : g->down;
    hipLaunchKernelGGL(k_from_sweep_order<double>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2, dir.d_store, din, dout);
    VRT_HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// J[site][l] = J_up + J_down from the two directions' planes (either may be NULL: that direction had no angle)
int J_from_native(vrt_plan *p, int64_t nlam, int64_t ld, const double *dJ_up, const double *dJ_down, double *dJ, hipStream_t st)
{
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    hipLaunchKernelGGL(k_combine_J<double>, tgrid, dim3(256), 0, st, n, (int)nlam, ld, 2, g->up.d_store, g->down.d_srank, dJ_up, dJ_down, dJ);
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
    const int lb = patches ? 2 << native_lg(p, kF32) : steps ? 2 : 1;
    const int64_t nl_pad = lb == 1 ? nlam : (nlam + 1) / 2 * 2;
    const size_t plane = (size_t)nl_pad * (size_t)n;
    // workspaces are kept as double buffers; a plane of T needs this many doubles
    auto dcount = [](size_t elems) { return (elems * sizeof(T) + 7) / 8; };
    int rc;
    if ((rc = ensure_dev(p->d_I, p->I_cap, dcount((size_t)std::max(1, A) * plane)))) return rc;
    T *wI = reinterpret_cast<T *>(p->d_I);
    // chained launch with the intensities as their own flags (vrt_patch.hip: chain_data_wait): every plane is filled with
    // the NaN pattern first; the boundary kernel below then writes the boundary layer and the never-visited site's zero
    const bool chain_df = patches && A > 0 && patch_chain_possible(p, (int)(nl_pad / 2), kF32) && patch_chain_dataflag(p, (int)(nl_pad / 2), kF32);
    // ... in ONE launch with the step's other preparations where those are a few microseconds each (k_chain_prepare)
    const bool prep = chain_df && !kF32 && lb == 2 && nlam <= 16 && alpha_mode != VRT_ALPHA_ANGLE_SITE_LAM;
    if (chain_df && !prep) VRT_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p->d_I, (int)0x7FF87FF8u, (size_t)A * plane * sizeof(T) / 4, st));
    ChainPrep cp{};
    bool prep_ctrl = false;
    const bool use_dir[2] = {p->n_up > 0, p->n_down > 0};
    // sweep-order S and J handed over by the caller (vrt_plan_execute_native_dev): the layer paths read / write them in place
    const bool nat = p->nat_mode;
    if (nat && (!steps || (kF32 ? !patches : lb != 2)))
        return fail(VRT_EINVAL, "sweep-order S and J need a layer path with one wavelength pair per block (float planes: the patch path)");
    for (int d = 0; d < 2; d++)
        if (use_dir[d] && !nat && (rc = ensure_dev(p->ws_S[d], p->ws_S_cap[d], dcount(plane)))) return rc;
    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)((nlam + 63) / 64));
    // a handful of wavelengths in the pair layout: the narrow forms of the layout changes (vrt_layout_kernels.h)
    const bool narrow = lb != 1 && nlam <= 16;
    int narrow_lgP = 0;
    while ((1 << narrow_lgP) < (int)((nlam + 1) / 2)) narrow_lgP++;
    const unsigned narrow_blocks = (unsigned)((((int64_t)n << narrow_lgP) + 255) / 256);
    TileArgs ta{};                                  // (zeroed, padding included: the chained launch compares argument blocks byte for byte)
    ta.n = n;
    ta.nlam = (int)nlam;
    ta.A = A;
    ta.alpha_mode = alpha_mode;
    ta.max_layers = p->tile_max_layers;
    ta.tile_stride = (int)((std::max<int64_t>(p->tile_max_layer_size, 1) + 2) & ~(int64_t)1);   // + the zero slot
    if (!patches && (rc = ensure_step_tables(p))) return rc;        // (their pointers are read just below)
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
    // a direction whose angles all advance on ONE internal stream gets its layout changes there too: the two
    // directions' transposes then run side by side instead of one after the other (C4: 0.2 ms of 8.7)
    hipStream_t dir_st[2] = {st, st};
    if (steps && A > 0) {
        const int G = std::max(1, std::min({p->tune.step_streams, 4, A}));
        if ((rc = ensure_step_streams(p, G))) return rc;
        bool forked = false;
        // (not for a chained step of a few wavelengths: its layout changes are a few microseconds each, less than the
        // event round trip that brings the second stream back)
        const bool tiny = narrow && patches && patch_chain_possible(p, (int)(nl_pad / 2), kF32);
        for (int d = 0; d < 2 && !tiny; d++) {
            if (!use_dir[d]) continue;
            for (int gi = 1; gi < G; gi++) {
                int have = 0;
                for (int j = p->step_group_off[(size_t)gi]; j < p->step_group_off[(size_t)gi + 1]; j++)
                    have += (p->dir_of_active[(size_t)p->h_step_angles[(size_t)j]] > 0) == (d == 0);
                if (have != (d == 0 ? p->n_up : p->n_down)) continue;
                if (!forked) VRT_HIP_TRY(hipEventRecord(p->step_fork, st));
                forked = true;
                VRT_HIP_TRY(hipStreamWaitEvent(p->step_stream[gi], p->step_fork, 0));
                dir_st[d] = p->step_stream[gi];
            }
        }
    }
    for (int d = 0; d < 2; d++) {
        const Direction &dir = d == 0 ? g->up : g->down;
        ta.lay[d] = dir.d_lay;
        ta.nlayers[d] = (int)dir.reduced.size() - 1;
        ta.S[d] = nullptr;
        ta.alpha[d] = nullptr;
        if (!use_dir[d]) continue;
        hipStream_t st = dir_st[d];                 // (shadows the caller's stream inside this loop)
        const bool with_alpha = alpha_mode == VRT_ALPHA_SITE_LAM;           // S and α of the direction in ONE launch
        if (with_alpha) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], dcount(plane)))) return rc;
            ta.alpha[d] = p->ws_A[d];
        }
        const T *in2 = with_alpha ? dalpha : nullptr;
        T *out2 = with_alpha ? reinterpret_cast<T *>(p->ws_A[d]) : nullptr;
        // (sweep-order S of the caller: read in place, no layout change; a caller-layout alpha still has its own)
        const T *inS = nat ? (with_alpha ? in2 : nullptr) : dS;
        T *outS = nat ? out2 : reinterpret_cast<T *>(p->ws_S[d]);
        const T *inB = nat ? nullptr : in2;
        T *outB = nat ? nullptr : out2;
        if (prep) {
            if constexpr (!kF32) {
                if (inS) { cp.tin[cp.njob] = inS; cp.tout[cp.njob] = reinterpret_cast<double *>(outS); cp.torder[cp.njob] = dir.d_store; cp.njob++; }
                if (inB) { cp.tin[cp.njob] = inB; cp.tout[cp.njob] = reinterpret_cast<double *>(outB); cp.torder[cp.njob] = dir.d_store; cp.njob++; }
                cp.n1[d] = dir.n1; cp.store[d] = dir.d_store; cp.rank[d] = dir.d_rank; cp.I0[d] = d == 0 ? dI0_up : dI0_down;
            }
        } else if (!inS) {
            // nothing to lay out
        } else if (narrow)
            hipLaunchKernelGGL(k_to_sweep_order_narrow<T>, dim3(narrow_blocks, inB ? 2 : 1), dim3(256), 0, st, n, (int)nlam, ld,
                               log2_pairs(lb), narrow_lgP, dir.d_store, inS, outS, inB, outB);
        else
            hipLaunchKernelGGL(k_to_sweep_order<T>, dim3(tgrid.x, tgrid.y, inB ? 2 : 1), dim3(256), 0, st, n, (int)nlam, ld, lb,
                               dir.d_store, inS, outS, inB, outB);
        ta.S[d] = nat ? reinterpret_cast<const double *>(p->nat_S[d]) : p->ws_S[d];
        // alpha per (site, wavelength) already in sweep order: the direction's plane set, read in place
        if (alpha_mode == VRT_ALPHA_SITE_LAM_NATIVE)
            ta.alpha[d] = reinterpret_cast<const double *>(dalpha) + (size_t)d * dcount(plane);
        if (alpha_mode == VRT_ALPHA_SITE) {
            if ((rc = ensure_dev(p->ws_A[d], p->ws_A_cap[d], dcount((size_t)n)))) return rc;
            hipLaunchKernelGGL(k_gather_vec<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                               dir.d_store, dalpha, reinterpret_cast<T *>(p->ws_A[d]));
            ta.alpha[d] = p->ws_A[d];
        }
        const int cnt = d == 0 ? p->n_up : p->n_down;
        if (dir.n1 > 0 && !prep) {
            const dim3 bgrid((unsigned)((dir.n1 + 63) / 64), (unsigned)((nlam + 63) / 64), (unsigned)cnt);
            hipLaunchKernelGGL(k_boundary_sweep_order<T>, bgrid, dim3(256), 0, st, n, (int)nlam, lb, dir.n1,
                               d == 0 ? p->d_angles_up : p->d_angles_down, dir.d_order, dir.d_srank,
                               d == 0 ? dI0_up : dI0_down, wI);
        }
    }
    if (prep) {
        if constexpr (!kF32) {
            cp.n = n; cp.ld = ld; cp.nlam = (int)nlam; cp.npair = (int)(nl_pad / 2); cp.lgP = narrow_lgP;
            cp.tblocks = narrow_blocks;
            cp.A = A;
            cp.fblocks = (unsigned)cp.npair * (unsigned)((n + 1023) / 1024);
            cp.I = p->d_I;
            for (int a = 0; a < A; a++) cp.down[a] = p->dir_of_active[(size_t)a] > 0 ? 0 : 1;
            cp.fill = 0x7FF87FF8u;
            cp.ctrl = p->d_chain_ctrl;                 // (exists from the plan's first chained launch on)
            cp.nctrl = chain_ctrl_words();
            prep_ctrl = cp.ctrl != nullptr;
            const unsigned blocks = (unsigned)cp.njob * cp.tblocks + (unsigned)A * cp.fblocks + (unsigned)((cp.nctrl + 255) / 256);
            hipLaunchKernelGGL(k_chain_prepare, dim3(blocks), dim3(256), 0, st, cp);
        }
    }
    if (alpha_mode == VRT_ALPHA_SITE_LAM_NATIVE) ta.alpha_mode = VRT_ALPHA_SITE_LAM;
    if (alpha_mode == VRT_ALPHA_ANGLE_NATIVE) {
        // already in storage-pair order per active angle (vrt_plan_alpha_to_native_dev or the
        // opacity prologue wrote it): no transposed copy, the kernels read the caller's buffer
        ta.alpha_mode = VRT_ALPHA_ANGLE_SITE_LAM;
        ta.alpha_angle = reinterpret_cast<const double *>(dalpha);
    } else if (alpha_mode == VRT_ALPHA_ANGLE_SITE_LAM) {
        if ((rc = ensure_dev(p->ws_AA, p->ws_AA_cap, dcount((size_t)A * plane)))) return rc;
        for (int a = 0; a < A; a++) {
            const Direction &dir = p->dir_of_active[(size_t)a] > 0 ? g->up : g->down;
            hipLaunchKernelGGL(k_to_sweep_order<T>, tgrid, dim3(256), 0, dir_st[p->dir_of_active[(size_t)a] > 0 ? 0 : 1],
                               n, (int)nlam, ld, lb, dir.d_store, dalpha + (size_t)a * (size_t)n * (size_t)ld,
                               reinterpret_cast<T *>(p->ws_AA) + (size_t)a * plane, (const T *)nullptr, (T *)nullptr);
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
        PatchReduce red_tmpl{};
        int owner_of_dir[2] = {-1, -1};
        int64_t reduced_upto[2] = {0, 0};
        if (patches && dJ) {
            for (int a = 0; a < A; a++) red_tmpl.w[a] = weights_user[p->user_of_active[(size_t)a]];
            for (int d = 0; d < 2; d++) {
                if (!use_dir[d]) continue;
                if (!nat && (rc = ensure_dev(p->ws_J[d], p->ws_J_cap[d], dcount(plane)))) return rc;
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
                red.Jd[r] = nat ? reinterpret_cast<double *>(p->nat_J[d]) : p->ws_J[d];
                red.count[r] = 0;
                for (int a = 0; a < A; a++)
                    if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) red.angles[r][red.count[r]++] = a;
                reduced_upto[d] = upto;
                r++;
            }
            return r > 0;
        };
        sa.level_map = nullptr;
        // ---- patches, chained: every layer of every angle inside ONE persistent launch (vrt_patch.hip: k_patch_chain) ----
        const bool chain = patches && patch_chain_possible(p, npair, kF32);
        if (chain) {
            // the layout changes of a direction may have run on an internal stream: the launch follows both
            for (int d = 0; d < 2; d++)
                for (int gi = 1; gi < G; gi++)
                    if (use_dir[d] && dir_st[d] == p->step_stream[gi]) {
                        VRT_HIP_TRY(hipEventRecord(p->step_join[gi], p->step_stream[gi]));
                        VRT_HIP_TRY(hipStreamWaitEvent(st, p->step_join[gi], 0));
                    }
            VRT_HIP_TRY(hipEventRecord(p->ev0, st));
            PatchReduce red = red_tmpl;
            if (dJ)
                for (int d = 0; d < 2; d++) {
                    red.count[d] = 0;
                    red.Jd[d] = use_dir[d] ? (nat ? reinterpret_cast<double *>(p->nat_J[d]) : p->ws_J[d]) : nullptr;
                    for (int a = 0; a < A; a++)
                        if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) red.angles[d][red.count[d]++] = a;
                    fused_dir[d] = use_dir[d];
                }
            if ((rc = launch_patch_chain(p, sa.ta, npair, st, kF32, dJ ? &red : nullptr, chain_df, prep_ctrl))) return rc;
            launches = 1;
            VRT_HIP_TRY(hipEventRecord(p->ev1, st));
        } else {
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
        }
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
            if (!use_dir[d]) {
                // (a direction without angles contributes nothing: its sweep-order J is handed back as zeros)
                if (nat && p->nat_J[d]) VRT_HIP_TRY(hipMemsetAsync(p->nat_J[d], 0, plane * sizeof(T), st));
                continue;
            }
            if (!nat && (rc = ensure_dev(p->ws_J[d], p->ws_J_cap[d], dcount(plane)))) return rc;
            DirWeights dw;
            dw.count = 0;
            for (int a = 0; a < A; a++)
                if ((p->dir_of_active[(size_t)a] > 0) == (d == 0)) {
                    dw.w[dw.count] = weights_user[p->user_of_active[(size_t)a]];
                    dw.idx[dw.count] = a;
                    dw.count++;
                }
            Jd[d] = reinterpret_cast<T *>(nat ? p->nat_J[d] : (void *)p->ws_J[d]);
            if (fused_dir[d]) continue;                  // formed layer by layer inside the sweep's launches
            hipLaunchKernelGGL(k_reduce_dir<T>, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st,
                               (int64_t)plane, (int64_t)plane, dw, wI, Jd[d]);
        }
        if (nat) {
            // the caller keeps J per direction in sweep order: no combination, no layout change
        } else if (narrow)
            hipLaunchKernelGGL(k_combine_J_narrow<T>, dim3(narrow_blocks), dim3(256), 0, st, n, (int)nlam, ld, log2_pairs(lb), narrow_lgP,
                               g->up.d_store, g->down.d_srank, Jd[0], Jd[1], dJ);
        else
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
