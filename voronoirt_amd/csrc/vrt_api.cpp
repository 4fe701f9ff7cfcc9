// extern "C" entry points of libvrt_hip.so (declared in include/voronoirt.h).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "vrt_internal.h"

namespace vrt {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

void tuning_from_env(Tuning &t)
{
    static const char *names[] = {"VRT_PATH", "VRT_STEP_K", "VRT_STEP_SINGLE", "VRT_STEP_PAIRS", "VRT_STEP_XCD",
                                  "VRT_STEP_STREAMS", "VRT_STEP_LEVEL_MAP", "VRT_STEP_GROUP_DIR", "VRT_TILE_WIDE",
                                  "VRT_TILE_PRE", "VRT_GRAPH", "VRT_PATCH_K", "VRT_PATCH_NT", "VRT_PATCH_OWN",
                                  "VRT_PATCH_Q", "VRT_PATCH_TARGET", "VRT_PATCH_SPLIT", "VRT_PAIR_BLOCK", "VRT_PATCH_QUAD", "VRT_PATCH_LEAN", "VRT_PATCH_CHAIN", "VRT_CHAIN_PAIRS", "VRT_CHAIN_SPIN", "VRT_CHAIN_DATAFLAG", "VRT_CHAIN_STATIC", "VRT_LAMBDA_NATIVE", "VRT_DEBUG_FLAGS", "VRT_DEBUG_SKIP_LEVELS",
                                  "VRT_TILE_DEBUG"};
    for (const char *nm : names) {
        const char *e = std::getenv(nm);
        if (e && *e && tuning_set(t, nm, e, /*created=*/false) != VRT_OK) {
            // a bad value keeps the default -- said once per variable, so that a typo such as VRT_PATH=patch does not
            // quietly benchmark another path than the one meant
            static std::mutex mu;
            static std::vector<std::string> said;
            std::lock_guard<std::mutex> lock(mu);
            if (std::find(said.begin(), said.end(), std::string(nm)) == said.end()) {
                said.push_back(nm);
                std::fprintf(stderr, "[libvrt_hip] ignoring %s=%s: %s\n", nm, e, g_err.c_str());
            }
        }
    }
}

int tuning_set(Tuning &t, const char *name, const char *value, bool created)
{
    if (!name || !value) return fail(VRT_EINVAL, "NULL option name or value");
    const std::string nm(name), v(value);
    if (nm == "VRT_PATH") {
        static const char *paths[] = {"auto", "levels", "tiles", "steps", "patches"};
        for (int i = 0; i < 5; i++)
            if (v == paths[i]) { t.path = i; return VRT_OK; }
        if (v.empty()) { t.path = 0; return VRT_OK; }
        return fail(VRT_EINVAL, "VRT_PATH must be auto, levels, tiles, steps or patches");
    }
    char *end = nullptr;
    const long x = std::strtol(value, &end, 10);
    if (end == value || *end != 0) return fail(VRT_EINVAL, "option " + nm + " needs an integer value");
    struct { const char *name; int *field; long lo, hi; bool creation_only; } tab[] = {
        {"VRT_STEP_K", &t.step_K, 0, 18, false}, {"VRT_STEP_SINGLE", &t.step_single, 0, 1, false},
        {"VRT_STEP_PAIRS", &t.step_pairs, 0, 64, false}, {"VRT_STEP_XCD", &t.step_xcd, 0, 2, false},
        {"VRT_STEP_STREAMS", &t.step_streams, 1, 4, false}, {"VRT_STEP_LEVEL_MAP", &t.step_level_map, 0, 1, false},
        {"VRT_STEP_GROUP_DIR", &t.step_group_dir, 0, 1, false}, {"VRT_TILE_WIDE", &t.tile_wide, 0, 1, false},
        {"VRT_TILE_PRE", &t.tile_pre, 0, 1, false}, {"VRT_GRAPH", &t.graph, 0, 1, false},
        {"VRT_PATCH_K", &t.patch_K, 1, 8, true}, {"VRT_PATCH_NT", &t.patch_NT, 64, 1024, true},
        {"VRT_PATCH_OWN", &t.patch_own, 0, 65535, true}, {"VRT_PATCH_Q", &t.patch_Q, 1, 4, false},
        {"VRT_PATCH_TARGET", &t.patch_target, 0, 1 << 20, false}, {"VRT_PATCH_SPLIT", &t.patch_split, 0, 4096, false}, {"VRT_PAIR_BLOCK", &t.pair_block, 1, 16, true},
        {"VRT_PATCH_QUAD", &t.patch_quad, 0, 1, true}, {"VRT_PATCH_LEAN", &t.patch_lean, 0, 1, false},
        {"VRT_PATCH_CHAIN", &t.patch_chain, 0, 2, false}, {"VRT_CHAIN_PAIRS", &t.chain_pairs, 1, 256, false},
        {"VRT_CHAIN_SPIN", &t.chain_spin, 1, 1 << 20, false}, {"VRT_CHAIN_DATAFLAG", &t.chain_dataflag, 0, 2, false},
        {"VRT_LAMBDA_NATIVE", &t.lambda_native, 0, 1, false}, {"VRT_CHAIN_STATIC", &t.chain_static, 0, 1, false}, {"VRT_DEBUG_FLAGS", &t.debug_flags, 0, 1 << 20, false},
        {"VRT_DEBUG_SKIP_LEVELS", &t.debug_skip_levels, 0, 1, false}, {"VRT_TILE_DEBUG", &t.tile_debug, 0, 1, false},
    };
    for (auto &o : tab)
        if (nm == o.name) {
            if (o.creation_only && created)
                return fail(VRT_EINVAL, "option " + nm + " shapes what plan creation builds: preset it in the environment");
            if (x < o.lo || x > o.hi) return fail(VRT_EINVAL, "option " + nm + " out of range");
            *o.field = (int)x;
            return VRT_OK;
        }
    return fail(VRT_EINVAL, "unknown option " + nm);
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE,
                    std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    return VRT_OK;
}

template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

int use_device(int device)
{
    if (device < 0)
        return fail(VRT_ENODEVICE, "host-only grid handle (device < 0): no compute without a HIP device");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(VRT_ENODEVICE, "no HIP device available (libvrt_hip has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(VRT_EINVAL, "device ordinal out of range");
    VRT_HIP_TRY(hipSetDevice(device));
    // an error some EARLIER call of this thread left behind (the host application's, or a refused argument of
    // ours) must not be reported by the hipGetLastError() checks behind this entry point's own launches
    (void)hipGetLastError();
    return VRT_OK;
}

static void free_grid(vrt_grid *g)
{
    if (!g) return;
    for (PlanCacheEntry *c : g->cache) {
        vrt_plan_destroy(c->plan);
        delete c;
    }
    g->cache.clear();
    dev_free(g->d_pos);
    dev_free(g->d_rowptr);
    dev_free(g->d_col);
    dev_free(g->d_lz);
    dev_free(g->d_lx);
    dev_free(g->d_ly);
    dev_free(g->up.d_order);
    dev_free(g->down.d_order);
    dev_free(g->up.d_rank);
    dev_free(g->down.d_rank);
    dev_free(g->up.d_lay);
    dev_free(g->down.d_lay);
    dev_free(g->d_scalars);
    dev_free(g->d_small);
    if (g->small_ev) (void)hipEventDestroy(g->small_ev);
    if (g->small_copy_ev) (void)hipEventDestroy(g->small_copy_ev);
    if (g->h_small) (void)hipHostFree(g->h_small);
    dev_free(g->up.d_store);
    dev_free(g->down.d_store);
    dev_free(g->up.d_srank);
    dev_free(g->down.d_srank);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static int upload_grid(vrt_grid *g)
{
    const int64_t n = g->n;
    const size_t nnz = g->col.size();
    int rc;
    if ((rc = dev_alloc(&g->d_pos, (size_t)3 * n))) return rc;
    if ((rc = dev_alloc(&g->d_rowptr, (size_t)n + 1))) return rc;
    if ((rc = dev_alloc(&g->d_col, nnz))) return rc;
    if ((rc = dev_alloc(&g->d_lz, nnz))) return rc;
    if ((rc = dev_alloc(&g->d_lx, nnz))) return rc;
    if ((rc = dev_alloc(&g->d_ly, nnz))) return rc;
    VRT_HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    VRT_HIP_TRY(hipMemcpy(g->d_pos, g->pos.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    VRT_HIP_TRY(hipMemcpy(g->d_rowptr, g->rowptr.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
    if (nnz)
        VRT_HIP_TRY(hipMemcpy(g->d_col, g->col.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    for (int d = 0; d < 2; d++) {
        Direction &dir = d == 0 ? g->up : g->down;
        std::vector<int32_t> order((size_t)n);
        for (int64_t i = 0; i < n; i++) order[(size_t)i] = (int32_t)(dir.perm[(size_t)i] - 1);
        if ((rc = dev_alloc(&dir.d_order, (size_t)n))) return rc;
        VRT_HIP_TRY(hipMemcpy(dir.d_order, order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        std::vector<int32_t> rank((size_t)n);
        for (int64_t i = 0; i < n; i++) rank[(size_t)order[(size_t)i]] = (int32_t)i;
        if ((rc = dev_alloc(&dir.d_rank, (size_t)n))) return rc;
        VRT_HIP_TRY(hipMemcpy(dir.d_rank, rank.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        std::vector<int32_t> srank((size_t)n);
        for (int64_t i = 0; i < n; i++) srank[(size_t)dir.store[(size_t)i]] = (int32_t)i;
        if ((rc = dev_alloc(&dir.d_store, (size_t)n))) return rc;
        if ((rc = dev_alloc(&dir.d_srank, (size_t)n))) return rc;
        VRT_HIP_TRY(hipMemcpy(dir.d_store, dir.store.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        VRT_HIP_TRY(hipMemcpy(dir.d_srank, srank.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        std::vector<int32_t> lay(dir.reduced.size());
        for (size_t j = 0; j < lay.size(); j++) lay[j] = (int32_t)(dir.reduced[j] - 1);
        if ((rc = dev_alloc(&dir.d_lay, lay.size()))) return rc;
        VRT_HIP_TRY(hipMemcpy(dir.d_lay, lay.data(), sizeof(int32_t) * lay.size(), hipMemcpyHostToDevice));
    }
    if ((rc = launch_delaunay_lines(g))) return rc;
    VRT_HIP_TRY(hipStreamSynchronize(g->stream));
    return VRT_OK;
}

static int grid_create_impl(int64_t n, const double *pos, const int64_t *nbr, int64_t D1,
                            const double bounds[6], int device, vrt_grid **out)
{
    if (!out) return fail(VRT_EINVAL, "out is NULL");
    *out = nullptr;
    if (!pos || !nbr || !bounds) return fail(VRT_EINVAL, "NULL argument");
    int rc = device >= 0 ? use_device(device) : VRT_OK;   // device < 0: host-only handle
    if (rc) return rc;
    vrt_grid *g = new (std::nothrow) vrt_grid();
    if (!g) return fail(VRT_ENOMEM, "out of host memory");
    g->device = device;
    rc = build_grid_host(g, n, pos, nbr, D1, bounds);
    if (!rc && device >= 0) rc = upload_grid(g);
    if (rc) {
        free_grid(g);
        return rc;
    }
    *out = g;
    return VRT_OK;
}

static const Direction &direction_of(const vrt_grid *g, int dir) { return dir > 0 ? g->up : g->down; }

static void free_plan(vrt_plan *p)
{
    if (!p) return;
    dev_free(p->d_up1);
    dev_free(p->d_up2);
    dev_free(p->d_d1);
    dev_free(p->d_d2);
    dev_free(p->d_w1);
    dev_free(p->d_w2);
    dev_free(p->d_r1);
    dev_free(p->d_r2);
    dev_free(p->d_node_site);
    dev_free(p->d_node_meta);
    dev_free(p->d_node_u1);
    dev_free(p->d_node_u2);
    dev_free(p->d_angles_up);
    dev_free(p->d_angles_down);
    dev_free(p->d_I);
    for (int i = 0; i < 6; i++) dev_free(p->d_stage[i]);
    dev_free(p->t_u1); dev_free(p->t_u2);
    dev_free(p->t_w1); dev_free(p->t_w2); dev_free(p->t_r1); dev_free(p->t_r2);
    dev_free(p->t_vis);
    dev_free(p->t_loc);
    dev_free(p->t_self);
    dev_free(p->t_vis_s);
    dev_free(p->t_loc_s);
    dev_free(p->t_gpos);
    dev_free(p->t_rank_s);
    dev_free(p->t_loc_ss);
    dev_free(p->t_code_ss);
    dev_free(p->d_nlev); dev_free(p->d_angle_dir); dev_free(p->d_task_map);
    for (int d = 0; d < 2; d++) { dev_free(p->ws_S[d]); dev_free(p->ws_A[d]); dev_free(p->ws_J[d]); }
    dev_free(p->ws_AA);
    for (int i = 0; i < 2; i++) dev_free(p->ws_cg[i]);
    dev_free(p->d_step_angles);
    dev_free(p->d_level_map);
    dev_free(p->d_patch_work);
    dev_free(p->d_chain_items); dev_free(p->d_chain_deps); dev_free(p->d_chain_progress); dev_free(p->d_chain_ctrl);
    for (auto &cs : p->chain_cache) { dev_free(cs.items); dev_free(cs.deps); }
    p->chain_cache.clear();
    if (p->h_chain_status) { (void)hipHostFree(p->h_chain_status); p->h_chain_status = nullptr; }
    if (p->d_chain_dev) { (void)hipFree(p->d_chain_dev); p->d_chain_dev = nullptr; }
    if (p->h_chain_dev_pinned) { (void)hipHostFree(p->h_chain_dev_pinned); p->h_chain_dev_pinned = nullptr; }
    if (p->chain_dev_ev) (void)hipEventDestroy(p->chain_dev_ev);
    dev_free(p->e_pos); dev_free(p->e_u1); dev_free(p->e_u2); dev_free(p->e_vis); dev_free(p->e_loc);
    dev_free(p->e_w1); dev_free(p->e_w2); dev_free(p->e_r1); dev_free(p->e_r2);
    if (p->step_fork) (void)hipEventDestroy(p->step_fork);
    for (int i = 0; i < 4; i++) {
        if (p->step_join[i]) (void)hipEventDestroy(p->step_join[i]);
        if (p->step_stream[i]) (void)hipStreamDestroy(p->step_stream[i]);
    }
    for (CopyLane &l : p->copy_lanes) {
        for (int b = 0; b < 2; b++) {
            if (l.pin[b]) (void)hipHostFree(l.pin[b]);
            if (l.ev[b]) (void)hipEventDestroy(l.ev[b]);
        }
        if (l.st) (void)hipStreamDestroy(l.st);
    }
    if (p->copy_done) (void)hipEventDestroy(p->copy_done);
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    delete p;
}

// Global-level schedule of the "levels" path (merged over the active angles), built on first use:
// plan creation only pays for what the default path of the grid needs.
static int ensure_level_schedule(vrt_plan *p)
{
    if (p->level_ready) return VRT_OK;
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int A = p->A;
    const int n_sweeps = p->n_sweeps;
    std::vector<AngleSchedule> sched((size_t)A);
    {
        unsigned hw = std::thread::hardware_concurrency();
        int nthr = (int)std::min<unsigned>(hw ? hw : 4, 16);
        nthr = std::max(1, std::min(nthr, std::max(A, 1)));
        if (!run_workers(nthr, [&](int t) {
                for (int a = t; a < A; a += nthr) {
                    const bool up = p->dir_of_active[(size_t)a] > 0;
                    build_angle_schedule(up ? g->up : g->down, /*ascending=*/up, n, n_sweeps,
                                         p->h_up1.data() + (size_t)a * n, p->h_up2.data() + (size_t)a * n,
                                         sched[(size_t)a]);
                }
            }))
            return fail(VRT_ENOMEM, "out of host memory while building the level schedule");
    }
    int64_t max_levels = 0, total = 0;
    for (int a = 0; a < A; a++) {
        if (sched[(size_t)a].bad_site >= 0) return fail(VRT_EGRID, "site without an upwind neighbour");
        max_levels = std::max<int64_t>(max_levels, (int64_t)sched[(size_t)a].level_off.size() - 1);
        total += (int64_t)sched[(size_t)a].site.size();
    }
    // merge: global level t = union over the angles of their level t
    std::vector<int32_t> srank_up((size_t)n);
    for (int64_t i = 0; i < n; i++) srank_up[(size_t)g->up.store[(size_t)i]] = (int32_t)i;
    std::vector<uint32_t> node_site((size_t)total), node_meta((size_t)total);
    std::vector<int32_t> node_u1((size_t)total), node_u2((size_t)total);   // the node's upwind ids ride along:
                                                                           // one dependent load less per launch
    p->level_off.assign((size_t)max_levels + 1, 0);
    int64_t at = 0;
    for (int64_t t = 0; t < max_levels; t++) {
        p->level_off[(size_t)t] = at;
        for (int a = 0; a < A; a++) {
            const AngleSchedule &s = sched[(size_t)a];
            if (t + 1 >= (int64_t)s.level_off.size()) continue;
            // Nodes of one level are independent, so their order inside the launch is free: sort
            // them along the (layer, Morton(x, y)) storage curve so that neighbouring workgroups
            // work on neighbouring sites and the upwind rows they share are still in L2.
            const int64_t x0 = s.level_off[(size_t)t], x1 = s.level_off[(size_t)t + 1];
            std::vector<std::pair<int32_t, int64_t>> keyed((size_t)(x1 - x0));
            for (int64_t x = x0; x < x1; x++)
                keyed[(size_t)(x - x0)] = {srank_up[(size_t)s.site[(size_t)x]], x};
            std::sort(keyed.begin(), keyed.end());
            for (const auto &kv : keyed) {
                const int64_t x = kv.second;
                node_site[(size_t)at] = s.site[(size_t)x];
                node_meta[(size_t)at] = (uint32_t)a | ((uint32_t)s.zflags[(size_t)x] << 8);
                node_u1[(size_t)at] = p->h_up1[(size_t)a * (size_t)n + s.site[(size_t)x]];
                node_u2[(size_t)at] = p->h_up2[(size_t)a * (size_t)n + s.site[(size_t)x]];
                at++;
            }
        }
    }
    p->level_off[(size_t)max_levels] = at;
    p->n_nodes = total;
    int rc = VRT_OK;
    if (!rc) rc = dev_alloc(&p->d_node_site, (size_t)total);
    if (!rc) rc = dev_alloc(&p->d_node_meta, (size_t)total);
    if (!rc) rc = dev_alloc(&p->d_node_u1, (size_t)total);
    if (!rc) rc = dev_alloc(&p->d_node_u2, (size_t)total);
    if (!rc && total) {
        const size_t b = sizeof(uint32_t) * (size_t)total;
        if (hipMemcpy(p->d_node_u1, node_u1.data(), b, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_node_u2, node_u2.data(), b, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_node_site, node_site.data(), b, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_node_meta, node_meta.data(), b, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(VRT_ENODEVICE, "uploading the level schedule failed");
    }
    if (rc) {                       // leave no half-built schedule behind
        dev_free(p->d_node_site);
        dev_free(p->d_node_meta);
        dev_free(p->d_node_u1);
        dev_free(p->d_node_u2);
        p->level_off.clear();
        p->n_nodes = 0;
        return rc;
    }
    p->level_ready = true;
    return VRT_OK;
}

static int plan_create_impl(vrt_grid *g, int64_t n_angles, const double *k, const int *dirs,
                            int n_sweeps, vrt_plan **out,
                            const std::vector<std::pair<std::string, std::string>> *options = nullptr)
{
    if (!out) return fail(VRT_EINVAL, "out is NULL");
    *out = nullptr;
    if (!g || !k) return fail(VRT_EINVAL, "NULL argument");
    if (n_angles < 1) return fail(VRT_EINVAL, "n_angles must be >= 1");
    if (n_sweeps < 1) return fail(VRT_EINVAL, "n_sweeps must be >= 1");
    int rc = use_device(g->device);
    if (rc) return rc;
#ifdef VRT_DIAG
    auto t_phase = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[plan_create] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };
#else
    auto tick = [](const char *) {};
#endif
    vrt_plan *p = new (std::nothrow) vrt_plan();
    if (!p) return fail(VRT_ENOMEM, "out of host memory");
    p->g = g;
    p->n_sweeps = n_sweeps;
    p->n_angles_user = n_angles;
    tuning_from_env(p->tune);             // the ONLY place the library reads its tuning environment variables
    {
        // The layer paths advance the two sweep directions on two streams: the caller's and ONE of the library's.
        // A process gets four hardware queues by default (GPU_MAX_HW_QUEUES); streams beyond them share queues
        // and serialise.  The library cannot see the host's other streams, but it can see a lowered limit.
        static bool warned = false;
        const char *q = std::getenv("GPU_MAX_HW_QUEUES");
        if (!warned && q && *q && std::atoi(q) < 2) {
            warned = true;
            std::fprintf(stderr, "[libvrt_hip] GPU_MAX_HW_QUEUES=%s: the sweep's two direction streams will share one "
                                 "hardware queue and run one after the other (set it to 2 or more)\n", q);
        }
    }
    if (options)
        for (const auto &o : *options) (void)tuning_set(p->tune, o.first.c_str(), o.second.c_str(), /*created=*/false);
    for (int64_t a = 0; a < n_angles; a++) {
        const double *ka = k + 3 * a;
        const double nrm = std::sqrt(ka[0] * ka[0] + ka[1] * ka[1] + ka[2] * ka[2]);
        if (!(std::fabs(nrm - 1.0) < 1e-6)) {   // functions.jl:432 asserts norm(k) ≈ 1
            free_plan(p);
            return fail(VRT_EINVAL, "direction " + std::to_string(a + 1) + " is not a unit vector");
        }
        // θ>90 up, θ<90 down, θ=90 skipped (lambda_iteration.jl:98,104).  cos(90° π/180) is 6.1e-17,
        // not 0, so a horizontal direction is recognised by |k_z| < 1e-12 (1e-12 rad from horizontal)
        int d = dirs ? (dirs[a] > 0 ? 1 : (dirs[a] < 0 ? -1 : 0))
                     : (std::fabs(ka[0]) < 1e-12 ? 0 : (ka[0] < 0 ? 1 : -1));
        if (d == 0) continue;
        p->user_of_active.push_back((int)a);
        p->dir_of_active.push_back(d);
        p->k.insert(p->k.end(), ka, ka + 3);
    }
    p->A = (int)p->user_of_active.size();
    if (p->A > kMaxAngles) {
        free_plan(p);
        return fail(VRT_EINVAL, "more than 64 active angles in one plan");
    }
    const int64_t n = g->n;
    const int A = p->A;
    const size_t tab = (size_t)A * (size_t)n;
#define VRT_TRY_FREE(expr)      \
    do {                        \
        int _rc = (expr);       \
        if (_rc) {              \
            free_plan(p);       \
            return _rc;         \
        }                       \
    } while (0)
#define VRT_HIP_TRY_FREE(expr)                                                             \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            free_plan(p);                                                                  \
            return fail(VRT_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                  \
    } while (0)
    VRT_TRY_FREE(dev_alloc(&p->d_up1, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_up2, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_d1, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_d2, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_w1, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_w2, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_r1, tab));
    VRT_TRY_FREE(dev_alloc(&p->d_r2, tab));
    VRT_HIP_TRY_FREE(hipEventCreate(&p->ev0));
    VRT_HIP_TRY_FREE(hipEventCreate(&p->ev1));
    for (int a = 0; a < A; a++) VRT_TRY_FREE(launch_upwind_table(p, a));
    std::vector<int32_t> up1(tab), up2(tab);
    VRT_HIP_TRY_FREE(hipStreamSynchronize(g->stream));
    if (tab) {
        VRT_HIP_TRY_FREE(hipMemcpy(up1.data(), p->d_up1, sizeof(int32_t) * tab, hipMemcpyDeviceToHost));
        VRT_HIP_TRY_FREE(hipMemcpy(up2.data(), p->d_up2, sizeof(int32_t) * tab, hipMemcpyDeviceToHost));
    }

    // per-angle layer-local schedules (always needed: they drive the default steps/tiles paths and
    // detect sites without an upwind neighbour), built concurrently on the host.  The global-level
    // schedule of the "levels" path is built on first use (ensure_level_schedule).
    tick("upwind tables (device) + D2H");
    p->h_up1.swap(up1);
    p->h_up2.swap(up2);
    std::vector<LayerSchedule> lsched((size_t)A);
    std::vector<PatchSchedule> psched((size_t)A);                // fused patch path (vrt_patch.cpp)
    {
        // shape of the patch kernel: K entries per thread x NT threads = largest dependency cone of a patch
        if (!patch_shape_exists(p->tune.patch_K, p->tune.patch_Q, p->tune.patch_NT)) {
            p->tune.patch_K = 1; p->tune.patch_NT = 512; p->tune.patch_Q = 1;
        }
        p->patch_K = p->tune.patch_K;
        p->patch_NT = p->tune.patch_NT;
        p->patch_cap = p->patch_K * p->patch_NT;
        // pairs of a site side by side in the patch path's planes: a power of two, and a plane block must stay
        // addressable with 32-bit byte offsets (n sites x 2^lg pairs x 16 bytes)
        int lg = 0;
        while ((2 << lg) <= p->tune.pair_block && lg < 4) lg++;
        while (lg > 0 && ((uint64_t)n << (lg + 4)) > 0xFFFFFFFFull) lg--;
        p->lg_pair_block = lg;
    }
    // a patch owns as many consecutive sites as its dependency cone leaves room for (VRT_PATCH_OWN: at most that many)
    const int patch_own = p->tune.patch_own > 0 ? std::min(p->tune.patch_own, p->patch_cap) : p->patch_cap;
    {
        // one job per angle, dealt to the host threads; a job splits further by layer (the layers of an angle are analysed
        // independently, vrt_patch.cpp) when threads are left over.  The patch builder returns the angle's layer schedule
        // as a by-product -- the same analysis -- so build_layer_schedule itself only runs for an angle whose layers do not
        // fit the packed encoding (it then says so too; the level kernels take over).
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        // (all angles at once when the host has four threads for each: no second round with most threads idle)
        const int nthr = std::max(1, (int)std::min<unsigned>(std::min(hw, (unsigned)A * 4u <= hw ? 64u : 16u), (unsigned)A));
        const int sub_threads = (int)std::max(1u, std::min(16u, hw / (unsigned)nthr));
        std::atomic<int> next_job(0);
        // (every schedule build holds a visit trace of up to n_sweeps n entries and a dozen n-sized arrays)
        if (!run_workers(nthr, [&](int) {
                for (;;) {
                    const int a = next_job.fetch_add(1);
                    if (a >= A) break;
                    const bool up = p->dir_of_active[(size_t)a] > 0;
                    const int32_t *u1 = p->h_up1.data() + (size_t)a * n, *u2 = p->h_up2.data() + (size_t)a * n;
                    build_patch_schedule(up ? g->up : g->down, /*ascending=*/up, n, n_sweeps, u1, u2, patch_own,
                                         p->patch_cap, psched[(size_t)a], sub_threads, &lsched[(size_t)a]);
                    if (!lsched[(size_t)a].ok && lsched[(size_t)a].bad_site < 0)
                        build_layer_schedule(up ? g->up : g->down, /*ascending=*/up, n, n_sweeps, u1, u2, lsched[(size_t)a]);
                }
            })) {
            free_plan(p);
            return fail(VRT_ENOMEM, "out of host memory while building the sweep schedules");
        }
    }
    tick("layer + patch schedules");
    for (int a = 0; a < A; a++) {
        if (lsched[(size_t)a].bad_site >= 0) {
            std::string msg = "site " + std::to_string(lsched[(size_t)a].bad_site + 1) +
                              " has no neighbour with k . line > -1 for angle " +
                              std::to_string(p->user_of_active[(size_t)a] + 1) +
                              " (the reference reads an uninitialised index here)";
            free_plan(p);
            return fail(VRT_EGRID, msg);
        }
    }
    std::vector<int32_t> ups, downs;
    for (int a = 0; a < A; a++) (p->dir_of_active[(size_t)a] > 0 ? ups : downs).push_back(a);
    p->n_up = (int)ups.size();
    p->n_down = (int)downs.size();
    VRT_TRY_FREE(dev_alloc(&p->d_angles_up, ups.size()));
    VRT_TRY_FREE(dev_alloc(&p->d_angles_down, downs.size()));
    if (!ups.empty())
        VRT_HIP_TRY_FREE(hipMemcpy(p->d_angles_up, ups.data(), sizeof(int32_t) * ups.size(), hipMemcpyHostToDevice));
    if (!downs.empty())
        VRT_HIP_TRY_FREE(hipMemcpy(p->d_angles_down, downs.data(), sizeof(int32_t) * downs.size(), hipMemcpyHostToDevice));
    // ---- layer paths: tables in storage order, per-layer level counts, patch schedules -------------
    {
        bool ok = A > 0;
        int64_t max_layer = 0, visits = 0;
        for (int a = 0; a < A; a++) {
            ok = ok && lsched[(size_t)a].ok;
            max_layer = std::max(max_layer, lsched[(size_t)a].max_layer_size);
            visits += lsched[(size_t)a].n_visits;
        }
        // the boundary layer is a layer too: k_sweep_tiles_pre keeps it in LDS as the first "previous" layer
        if (p->n_up > 0) max_layer = std::max(max_layer, g->up.n1);
        if (p->n_down > 0) max_layer = std::max(max_layer, g->down.n1);
        if (n >= ((int64_t)1 << 28)) ok = false;  // the layer kernels index 16-byte pair planes with 32-bit byte offsets
        // the layer-step level kernels hold a whole layer per workgroup: 8192 sites as fp64 wavelength
        // pairs, 12 288 as fp64 single wavelengths, 18 432 as fp32 ones (vrt_step_kernels.h); the fused
        // patch kernel (vrt_patch.hip) has no such limit
        const bool tile_ok = ok && max_layer <= steps_max_layer(/*f32=*/true);
        bool patch_ok = ok;
        int64_t n_patches = 0, n_entries = 0;
        for (int a = 0; a < A && patch_ok; a++) {
            patch_ok = psched[(size_t)a].ok;
            n_patches += (int64_t)psched[(size_t)a].patch_own_lo.size();
            n_entries += (int64_t)psched[(size_t)a].entry_pos.size();
        }
        if (n_entries >= ((int64_t)1 << 31) - 1 || n_patches >= ((int64_t)1 << 31) - 1) patch_ok = false;
        p->tile_ok = tile_ok;
        p->patch_ok = patch_ok;
        p->tile_max_layer_size = max_layer;
        p->tile_visits = visits;
        p->tile_K = max_layer <= 2048 ? 2 : max_layer <= 4096 ? 4 : 8;
        if (tile_ok || patch_ok) {
            VRT_TRY_FREE(dev_alloc(&p->t_u1, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_u2, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_w1, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_w2, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_r1, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_r2, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_vis, tab));
            VRT_TRY_FREE(dev_alloc(&p->t_loc, tab));
            // (the sorted-slot tables of the steps / tiles paths are built when one of those paths first runs:
            // ensure_step_tables -- the default patch path needs none of them)
            const int maxL = (int)std::max(g->up.reduced.size(), g->down.reduced.size()) - 1;
            p->tile_max_layers = maxL;
            if (patch_ok) {
                const size_t ne = (size_t)n_entries;
                VRT_TRY_FREE(dev_alloc(&p->e_pos, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_u1, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_u2, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_vis, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_loc, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_w1, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_w2, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_r1, ne));
                VRT_TRY_FREE(dev_alloc(&p->e_r2, ne));
                p->h_patch_first.assign((size_t)A * (size_t)(maxL + 2), 0);
                p->h_patch_rec.reserve((size_t)n_patches);
                p->n_patches = n_patches;
                p->n_patch_entries = n_entries;
            }
            uint32_t *d_vis_site = nullptr;
            VRT_TRY_FREE(dev_alloc(&d_vis_site, (size_t)n));
            std::vector<int32_t> nlev((size_t)A * (size_t)(maxL + 1), 0), adir((size_t)A);
            std::vector<int2> rec2;
            p->angle_visits.assign((size_t)A, 0);
            int64_t ent_base = 0;
            for (int a = 0; a < A; a++) {
                hipError_t e = hipMemcpy(d_vis_site, lsched[(size_t)a].vis.data(), sizeof(uint32_t) * n,
                                         hipMemcpyHostToDevice);
                int rc2 = e == hipSuccess ? launch_permute_table(p, a, d_vis_site) : VRT_ENODEVICE;
                if (!rc2 && patch_ok) {    // patch records + entry tables of this angle
                    PatchSchedule &ps = psched[(size_t)a];
                    const size_t np_a = ps.patch_own_lo.size(), ne_a = ps.entry_pos.size();
                    const int32_t pbase = (int32_t)p->h_patch_rec.size();
                    int32_t *first = p->h_patch_first.data() + (size_t)a * (size_t)(maxL + 2);
                    for (int l = 0; l <= maxL + 1; l++)
                        first[l] = pbase + ps.layer_patch_off[std::min<size_t>((size_t)l, ps.layer_patch_off.size() - 1)];
                    for (size_t q = 0; q < np_a; q++) {
                        p->h_patch_rec.push_back(make_int4((int)(ent_base + ps.patch_ent_off[q]),
                                                           (int)(ps.patch_ent_off[q + 1] - ps.patch_ent_off[q]),
                                                           ps.patch_own_lo[q], ps.patch_own_cnt[q]));
                        rec2.push_back(make_int2(ps.patch_nlev[q], a));
                        p->h_patch_dep_off.push_back((int64_t)p->h_patch_deps.size());
                        for (int64_t j = ps.dep_off[q]; j < ps.dep_off[q + 1]; j++)
                            p->h_patch_deps.push_back(pbase + ps.dep_list[(size_t)j]);
                    }
                    if (ne_a) {
                        if (hipMemcpy(p->e_pos + ent_base, ps.entry_pos.data(), sizeof(int32_t) * ne_a, hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->e_vis + ent_base, ps.entry_vis.data(), sizeof(uint32_t) * ne_a, hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->e_loc + ent_base, ps.entry_loc.data(), sizeof(uint32_t) * ne_a, hipMemcpyHostToDevice) != hipSuccess)
                            rc2 = VRT_ENODEVICE;
                        if (!rc2) rc2 = launch_patch_entries(p, a, ent_base, (int64_t)ne_a);
                    }
                    ent_base += (int64_t)ne_a;
                    p->n_patch_visits += ps.n_visits;
                    ps = PatchSchedule();
                }
                if (!rc2 && hipStreamSynchronize(g->stream) != hipSuccess) rc2 = VRT_ENODEVICE;
                if (rc2) {
                    dev_free(d_vis_site);
                    free_plan(p);
                    return fail(VRT_ENODEVICE, "building the sweep-order tables failed");
                }
                const std::vector<int32_t> &nl = lsched[(size_t)a].nlev;
                for (size_t l = 0; l < nl.size() && l <= (size_t)maxL; l++)
                    nlev[(size_t)a * (size_t)(maxL + 1) + l] = nl[l];
                adir[(size_t)a] = p->dir_of_active[(size_t)a] > 0 ? 0 : 1;
                p->angle_visits[(size_t)a] = lsched[(size_t)a].n_visits + n;   // + n: phase 1 touches every site
                {
                    double sum = 0.0;
                    int cntl = 0;
                    for (size_t l = 2; l < nl.size(); l++, cntl++) sum += nl[l];
                    p->angle_mean_levels.push_back(cntl ? sum / cntl : 0.0);
                }
            }
            dev_free(d_vis_site);
            VRT_TRY_FREE(dev_alloc(&p->d_nlev, nlev.size()));
            VRT_TRY_FREE(dev_alloc(&p->d_angle_dir, (size_t)A));
            VRT_HIP_TRY_FREE(hipMemcpy(p->d_nlev, nlev.data(), sizeof(int32_t) * nlev.size(), hipMemcpyHostToDevice));
            VRT_HIP_TRY_FREE(hipMemcpy(p->d_angle_dir, adir.data(), sizeof(int32_t) * A, hipMemcpyHostToDevice));
            if (patch_ok) {
                p->h_patch_dep_off.push_back((int64_t)p->h_patch_deps.size());
                p->h_patch_rec2 = rec2;
            }
        }
    }
    tick("tables upload + entry kernels");
#undef VRT_TRY_FREE
#undef VRT_HIP_TRY_FREE
    *out = p;
    return VRT_OK;
}

}  // namespace vrt (reopened below)

// Sorted-slot tables of the steps / tiles paths (thread assignment of their level kernels, compact coupling lists),
// built when one of those paths first runs on the plan: they cost seven [A][n] device arrays and a host pass per
// angle that the default patch path never needs (plan creation at C4 size: 0.95 -> 0.6 s).
int vrt::ensure_step_tables(vrt_plan *p)
{
    using namespace vrt;
    if (p->step_tables_ready) return VRT_OK;
    if (!p->tile_ok) return fail(VRT_EINVAL, "the grid does not fit the steps / tiles kernels");
    vrt_grid *g = p->g;
    const int64_t n = g->n;
    const int A = p->A, n_sweeps = p->n_sweeps;
    const size_t tab = (size_t)A * (size_t)n;
    int rc;
    // (a table an earlier, failed attempt already allocated is kept: a retry neither leaks it nor overwrites its pointer)
    auto need = [&](auto *&ptr) -> int { return ptr ? VRT_OK : dev_alloc(&ptr, tab); };
    if ((rc = need(p->t_self)) || (rc = need(p->t_vis_s)) || (rc = need(p->t_loc_s)) || (rc = need(p->t_gpos)) ||
        (rc = need(p->t_rank_s)) || (rc = need(p->t_loc_ss)))
        return rc;
    if (p->tile_max_layer_size <= 4096 && (rc = need(p->t_code_ss))) return rc;
    std::vector<std::vector<int32_t>> sorted_self((size_t)A);
    unsigned hw = std::thread::hardware_concurrency();
    const int nthr = std::max(1, std::min<int>((int)std::min<unsigned>(hw ? hw : 4, 16), A));
    if (!run_workers(nthr, [&](int t) {
            for (int a = t; a < A; a += nthr) {
                const bool up = p->dir_of_active[(size_t)a] > 0;
                LayerSchedule ls;
                build_layer_schedule(up ? g->up : g->down, /*ascending=*/up, n, n_sweeps, p->h_up1.data() + (size_t)a * n,
                                     p->h_up2.data() + (size_t)a * n, ls);
                build_sorted_slots(up ? g->up : g->down, n, ls.vis, sorted_self[(size_t)a]);
            }
        }))
        return fail(VRT_ENOMEM, "out of host memory while building the layer-step tables");
    for (int a = 0; a < A; a++) {
        VRT_HIP_TRY(hipMemcpy(p->t_self + (size_t)a * n, sorted_self[(size_t)a].data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
        std::vector<int32_t>().swap(sorted_self[(size_t)a]);
        if ((rc = launch_sorted_tables(p, a)) || (rc = launch_gpos(p, a))) return rc;
    }
    VRT_HIP_TRY(hipStreamSynchronize(g->stream));
    p->step_tables_ready = true;
    return VRT_OK;
}

namespace vrt {

static int ensure(double *&buf, size_t &cap, size_t count)
{
    if (count <= cap && buf) return VRT_OK;
    dev_free(buf);
    cap = 0;
    int rc = dev_alloc(&buf, count);
    if (rc) return rc;
    cap = count;
    return VRT_OK;
}

int execute_dev_locked(vrt_plan *p, int64_t nlam, int64_t ld, const void *dS_, const void *dalpha_, int alpha_mode,
                       const void *dI0_up_, const void *dI0_down_, const double *weights, void *dJ_, void *dI_out_,
                       hipStream_t st, bool f32)
{
    const void *dS = dS_, *dalpha = dalpha_;
    void *dJ = dJ_, *dI_out = dI_out_;
    vrt_grid *g = p->g;
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (!dS || !dalpha) return fail(VRT_EINVAL, "S and alpha must not be NULL");
    if (alpha_mode < 0 || alpha_mode > VRT_ALPHA_SITE_LAM_NATIVE) return fail(VRT_EINVAL, "bad alpha_mode");
    if (alpha_mode == VRT_ALPHA_SITE_LAM_NATIVE && !p->nat_mode)
        return fail(VRT_EINVAL, "alpha per (site, wavelength) in sweep order goes with sweep-order S and J (vrt_plan_execute_native_dev)");
    if (dJ && !weights) return fail(VRT_EINVAL, "weights must be given when J is requested");
    int rc = use_device(g->device);
    if (rc) return rc;
    const int64_t n = g->n;
    // the user's per-angle alpha is indexed by USER angle; the plan's by active angle.  They
    // coincide unless a θ = 90 direction was skipped, which per-angle alpha does not support.
    if (alpha_mode == VRT_ALPHA_ANGLE_SITE_LAM && p->A != (int)p->n_angles_user)
        return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
    const bool steps_ok = p->tile_ok && p->tile_max_layer_size <= steps_max_layer(f32);
    if (alpha_mode == VRT_ALPHA_ANGLE_NATIVE && !(p->patch_ok || (steps_ok && !f32)))
        return fail(VRT_EINVAL, "native-layout alpha needs a layer path (at most 4 visits per site and 255 levels per layer)");
    {
        // Three device paths produce the same results:
        //   "levels"  one launch per dependency level over all angles; any grid;
        //   "steps"   two launches per BFS layer: chip-wide coefficient kernel + one workgroup per
        //             (angle, wavelength) running the layer's Gauss-Seidel levels on an LDS tile;
        //   "tiles"   ONE launch: each (angle, wavelength) workgroup walks all layers itself.
        // tiles needs layers of at most 8192 sites, steps of at most 12 288 (fp64) / 18 432 (fp32
        // storage) and both <= 255 levels per layer.  Default when that holds: tiles while the
        // (angle, wavelength) problems fit one round of workgroups (<= 256) AND the layers are small
        // (<= 4096 sites: the one launch has no per-layer launch cost -- C2, 2738-site layers: 1.2 ms
        // vs 1.8 ms on steps, which is host-launch-bound there); steps otherwise (its chip-wide
        // coefficient kernel wins once a layer holds more than a few sites per thread -- 1M sites x
        // 12 angles x 1 λ: 5.1 vs 7.8 ms; C4: 11.8 vs 20.5 ms); levels when the grid does not fit.
        // VRT_PATH selects one explicitly.
        const bool tiles_ok = steps_ok && !f32 && p->tile_max_layer_size <= 8192;
        int path = 1;
        if (tiles_ok && (int64_t)p->A * nlam <= 2 && p->tile_max_layer_size <= 4096)
            path = 2;                 // a single solve on small layers: two launches in all
        else if (p->patch_ok)
            path = 4;
        else if (steps_ok)
            path = 3;
        if (p->tune.path) path = p->tune.path;
        // the native layout IS the storage order of the layer paths
        // (laid out for the patch path when the grid fits it: then the steps path can only read it with one pair per block)
        if (alpha_mode == VRT_ALPHA_ANGLE_NATIVE) {
            const bool steps_can = steps_ok && !f32 && native_lg(p, f32) == 0;
            if (path == 3 && !steps_can) path = 4;
            else if (path != 3 && path != 4) path = p->patch_ok ? 4 : 3;
        }
        if (p->A == 0) path = 1;      // nothing to solve (every direction skipped): J = 0 via the level path
        if ((path == 3 && !steps_ok) || (path == 2 && !tiles_ok) || (path == 4 && !p->patch_ok))
            return fail(VRT_EINVAL, "VRT_PATH = tiles / steps / patches but the grid (or the fp32 storage type) does not fit those kernels");
        if (path != 1) {
            p->last_path = path;
            return execute_tiles(p, nlam, ld, dS_, dalpha_, alpha_mode, dI0_up_, dI0_down_, weights, dJ_,
                                 dI_out_, st, f32);
        }
        p->last_path = 1;
    }
    if ((rc = ensure_level_schedule(p))) return rc;
    const size_t need = (size_t)std::max(1, p->A) * (size_t)n * (size_t)nlam;
    if ((rc = ensure(p->d_I, p->I_cap, f32 ? (need + 1) / 2 : need))) return rc;
    p->I_ld = nlam;
    SweepArgs sa;
    sa.f32 = f32;
    sa.n = n;
    sa.nlam = nlam;
    sa.ldS = ld;
    sa.ldA = alpha_mode == VRT_ALPHA_SITE ? 1 : ld;
    sa.ldI = nlam;
    sa.S = dS_;
    sa.alpha = dalpha_;
    sa.alpha_mode = alpha_mode;
    sa.I = p->d_I;
    if ((rc = launch_boundary(p, sa, dI0_up_, dI0_down_, st))) return rc;
    VRT_HIP_TRY(hipEventRecord(p->ev0, st));
    {
        // The level sequence is hundreds to thousands of short dependent launches.  VRT_GRAPH=1
        // captures it once into a hipGraph and replays it while the arguments stay the same.
        // Measured on MI355X it changes nothing (C2: 7.57 vs 7.56 ms, C4: 24.87 vs 24.91 ms --
        // the launches are bound by the dependent-load latency inside each level, not by the
        // host), so eager launches stay the default.
        const bool use_graph = p->tune.graph == 1;
        SweepKey key;
        key.nlam = sa.nlam; key.ldS = sa.ldS; key.ldA = sa.ldA; key.ldI = sa.ldI;
        key.S = sa.S; key.alpha = sa.alpha; key.I = sa.I; key.alpha_mode = sa.alpha_mode; key.f32 = sa.f32;
        bool replayed = false;
        if (use_graph) {
            if (!(p->graph_exec && p->graph_key == key)) {
                if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
                p->graph_exec = nullptr;
                hipGraph_t graph = nullptr;
                if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const int rc2 = launch_sweep_levels(p, sa, st, &p->last_launches);
                    const hipError_t e = hipStreamEndCapture(st, &graph);
                    if (rc2 == VRT_OK && e == hipSuccess && graph &&
                        hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess)
                        p->graph_key = key;
                    else
                        p->graph_exec = nullptr;
                    if (graph) (void)hipGraphDestroy(graph);
                }
                (void)hipGetLastError();
            }
            if (p->graph_exec && hipGraphLaunch(p->graph_exec, st) == hipSuccess) replayed = true;
        }
        if (!replayed && (rc = launch_sweep_levels(p, sa, st, &p->last_launches))) return rc;
    }
    VRT_HIP_TRY(hipEventRecord(p->ev1, st));
    p->ev_valid = true;
    if (dJ) {
        double wact[kMaxAngles];
        for (int a = 0; a < p->A; a++) wact[a] = weights[p->user_of_active[(size_t)a]];
        if ((rc = launch_reduce_J(p, sa, wact, dJ_, ld, st))) return rc;
    }
    if (dI_out && (rc = launch_copy_I_out(p, sa, dI_out_, ld, st))) return rc;
    return VRT_OK;
}

int execute_native_locked(vrt_plan *p, int64_t nlam, const void *dS_up, const void *dS_down, const void *dalpha, int alpha_mode,
                          const void *dI0_up, const void *dI0_down, const double *weights, void *dJ_up, void *dJ_down,
                          hipStream_t st, bool f32)
{
    int rc = native_planes_ok(p, f32);
    if (rc) return rc;
    if ((p->n_up > 0 && !dS_up) || (p->n_down > 0 && !dS_down)) return fail(VRT_EINVAL, "S of a direction with angles must not be NULL");
    if (!dJ_up != !dJ_down) return fail(VRT_EINVAL, "J_up and J_down must be given together (or both NULL)");
    if (alpha_mode != VRT_ALPHA_SITE && alpha_mode != VRT_ALPHA_ANGLE_NATIVE && alpha_mode != VRT_ALPHA_SITE_LAM_NATIVE)
        return fail(VRT_EINVAL, "sweep-order S goes with alpha per site (0), native per angle (3) or per (site, wavelength) in sweep order (4): the other layouts carry the caller's leading dimension");
    // (the level path -- grids whose schedule does not fit the layer kernels -- keeps the caller's layout)
    const int keep = p->tune.path;
    if (keep == 1 || keep == 2 || (f32 && keep == 3)) p->tune.path = 0;
    p->nat_S[0] = dS_up; p->nat_S[1] = dS_down;
    p->nat_J[0] = dJ_up; p->nat_J[1] = dJ_down;
    p->nat_mode = true;
    const void *anyS = dS_up ? dS_up : dS_down;
    rc = execute_dev_locked(p, nlam, (nlam + 1) / 2 * 2, anyS, dalpha, alpha_mode, dI0_up, dI0_down, weights, dJ_up, nullptr, st, f32);
    p->nat_mode = false;
    p->nat_S[0] = p->nat_S[1] = nullptr;
    p->nat_J[0] = p->nat_J[1] = nullptr;
    p->tune.path = keep;
    if (!rc && p->last_path != 3 && p->last_path != 4) return fail(VRT_EINVAL, "sweep-order S and J: the plan did not run on a layer path");
    return rc;
}

// the float forms of the sweep-order entry points share this: lock, device, layout check, then `fn`
template <typename F>
static int native_f32_call(vrt_plan *p, int64_t nlam, int64_t ld, bool args_ok, F fn)
{
    DeviceScope scope;
    if (!p || !args_ok) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (!rc) rc = native_planes_ok(p, true);
        if (rc) return rc;
        return fn();
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

}  // namespace vrt

using namespace vrt;

extern "C" {

const char *vrt_last_error(void) { return g_err.c_str(); }

int vrt_version(void) { return 100; }

int vrt_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int vrt_grid_create(int64_t n, const double *pos_zxy, const int64_t *nbr, int64_t D1,
                    const double bounds[6], int device, vrt_grid **out)
{
    DeviceScope scope;
    try {
        return grid_create_impl(n, pos_zxy, nbr, D1, bounds, device, out);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_grid_create_from_file(const char *neighbours_file, int64_t n, const double *pos_zxy,
                              const double bounds[6], int device, vrt_grid **out)
{
    DeviceScope scope;
    try {
        if (!neighbours_file) return fail(VRT_EINVAL, "NULL file name");
        if (n < 1) return fail(VRT_EINVAL, "n must be positive");
        std::vector<int64_t> M;
        int64_t D1 = 0;
        int rc = parse_neighbour_file(neighbours_file, n, M, D1);
        if (rc) return rc;
        return grid_create_impl(n, pos_zxy, M.data(), D1, bounds, device, out);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_tessellate(int64_t n, const double *pos_zxy, const double bounds[6], int64_t D1, int64_t *nbr,
                   int64_t *max_count)
{
    if (!pos_zxy || !bounds || !nbr) return fail(VRT_EINVAL, "NULL argument");
    if (n < 2 || n >= ((int64_t)1 << 30)) return fail(VRT_EINVAL, "n must be in [2, 2^30)");
    if (D1 < 5) return fail(VRT_EINVAL, "D1 must leave room for at least four neighbours");
    try {
        unsigned hw = std::thread::hardware_concurrency();
        const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 4), 64, n / 256 + 1}));
        return tessellate_host(n, pos_zxy, bounds, D1, nbr, max_count, nthr);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_write_neighbours_file(const char *path, int64_t n, const int64_t *nbr, int64_t D1)
{
    if (!path || !nbr) return fail(VRT_EINVAL, "NULL argument");
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(VRT_EIO, std::string("cannot write ") + path);
    for (int64_t i = 0; i < n; i++) {       // voro++ "%i %n": id, then the neighbours (output_sites.cc:49)
        std::fprintf(f, "%lld", (long long)(i + 1));
        const int64_t cnt = std::min<int64_t>(nbr[i], D1 - 1);
        for (int64_t q = 1; q <= cnt; q++) std::fprintf(f, " %lld", (long long)nbr[i + n * q]);
        std::fputc('\n', f);
    }
    if (std::fclose(f) != 0) return fail(VRT_EIO, std::string("error writing ") + path);
    return VRT_OK;
}

void vrt_grid_destroy(vrt_grid *g)
{
    DeviceScope scope;
    if (g && g->device >= 0) (void)hipSetDevice(g->device);
    free_grid(g);
}

int64_t vrt_grid_n(const vrt_grid *g) { return g ? g->n : 0; }
int64_t vrt_grid_max_neighbours(const vrt_grid *g) { return g ? g->D : 0; }
int64_t vrt_grid_num_layer_offsets(const vrt_grid *g, int dir)
{
    return g ? (int64_t)direction_of(g, dir).reduced.size() : 0;
}

int vrt_grid_get_layers(const vrt_grid *g, int dir, int64_t *out)
{
    if (!g || !out) return fail(VRT_EINVAL, "NULL argument");
    const Direction &d = direction_of(g, dir);
    std::copy(d.reduced.begin(), d.reduced.end(), out);
    return VRT_OK;
}

int vrt_grid_get_perm(const vrt_grid *g, int dir, int64_t *out)
{
    if (!g || !out) return fail(VRT_EINVAL, "NULL argument");
    const Direction &d = direction_of(g, dir);
    std::copy(d.perm.begin(), d.perm.end(), out);
    return VRT_OK;
}

int vrt_grid_get_delaunay_lines(const vrt_grid *g, double *out)
{
    DeviceScope scope;
    if (!g || !out) return fail(VRT_EINVAL, "NULL argument");
    try {
        int rc = use_device(g->device);
        if (rc) return rc;
        const size_t nnz = g->col.size();
        std::vector<double> lz(nnz), lx(nnz), ly(nnz);
        if (nnz) {
            VRT_HIP_TRY(hipMemcpy(lz.data(), g->d_lz, sizeof(double) * nnz, hipMemcpyDeviceToHost));
            VRT_HIP_TRY(hipMemcpy(lx.data(), g->d_lx, sizeof(double) * nnz, hipMemcpyDeviceToHost));
            VRT_HIP_TRY(hipMemcpy(ly.data(), g->d_ly, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        }
        const int64_t D = g->D;
        std::fill(out, out + (size_t)3 * (size_t)D * (size_t)g->n, 0.0);
        for (int64_t i = 0; i < g->n; i++)
            for (int32_t e = g->rowptr[(size_t)i]; e < g->rowptr[(size_t)i + 1]; e++) {
                const int64_t j = e - g->rowptr[(size_t)i];
                double *o = out + 3 * ((size_t)j + (size_t)D * (size_t)i);
                o[0] = lz[(size_t)e];
                o[1] = lx[(size_t)e];
                o[2] = ly[(size_t)e];
            }
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

void vrt_direction(double theta_deg, double phi_deg, double k[3])
{
    // lambda_iteration.jl:87: degrees -> radians as θ*π/180
    const double pi = 3.14159265358979323846;
    const double th = theta_deg * pi / 180, ph = phi_deg * pi / 180;
    k[0] = std::cos(th);
    k[1] = std::cos(ph) * std::sin(th);
    k[2] = std::sin(ph) * std::sin(th);
}

int vrt_plan_create_ex(vrt_grid *g, int64_t n_angles, const double *k, const int *dirs,
                       int n_sweeps, vrt_plan **out)
{
    DeviceScope scope;
    try {
        return plan_create_impl(g, n_angles, k, dirs, n_sweeps, out);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_create(vrt_grid *g, int64_t n_angles, const double *k, int n_sweeps, vrt_plan **out)
{
    return vrt_plan_create_ex(g, n_angles, k, nullptr, n_sweeps, out);
}

void vrt_plan_destroy(vrt_plan *p)
{
    DeviceScope scope;
    if (p && p->g) (void)hipSetDevice(p->g->device);
    free_plan(p);
}

int64_t vrt_plan_num_levels(const vrt_plan *cp)
{
    if (!cp) return 0;
    vrt_plan *p = const_cast<vrt_plan *>(cp);      // the level schedule is built lazily, under the plan's mutex
    DeviceScope scope;
    std::lock_guard<std::mutex> lock(p->mu);
    if (use_device(p->g->device) || ensure_level_schedule(p)) return -1;
    return (int64_t)p->level_off.size() - 1;
}
int64_t vrt_plan_num_nodes(const vrt_plan *cp)
{
    if (!cp) return 0;
    vrt_plan *p = const_cast<vrt_plan *>(cp);
    DeviceScope scope;
    std::lock_guard<std::mutex> lock(p->mu);
    if (use_device(p->g->device) || ensure_level_schedule(p)) return -1;
    return p->n_nodes;
}

int vrt_plan_get_upwind(const vrt_plan *p, int64_t angle, int64_t *up, double *dots, double *w,
                        double *r)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    if (angle < 0 || angle >= p->n_angles_user) return fail(VRT_EINVAL, "angle out of range");
    int a = -1;
    for (int i = 0; i < p->A; i++)
        if (p->user_of_active[(size_t)i] == (int)angle) a = i;
    if (a < 0) return fail(VRT_EINVAL, "angle is skipped (θ = 90) and has no table");
    try {
        int rc = use_device(p->g->device);
        if (rc) return rc;
        const int64_t n = p->g->n;
        const size_t o = (size_t)a * (size_t)n;
        std::vector<int32_t> a1((size_t)n), a2((size_t)n);
        std::vector<double> x1((size_t)n), x2((size_t)n);
        if (up) {
            VRT_HIP_TRY(hipMemcpy(a1.data(), p->d_up1 + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            VRT_HIP_TRY(hipMemcpy(a2.data(), p->d_up2 + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; i++) {
                up[2 * i] = a1[(size_t)i] < 0 ? 0 : (int64_t)a1[(size_t)i] + 1;
                up[2 * i + 1] = a2[(size_t)i] < 0 ? 0 : (int64_t)a2[(size_t)i] + 1;
            }
        }
        const double *src1[3] = {p->d_d1, p->d_w1, p->d_r1};
        const double *src2[3] = {p->d_d2, p->d_w2, p->d_r2};
        double *dst[3] = {dots, w, r};
        for (int q = 0; q < 3; q++) {
            if (!dst[q]) continue;
            VRT_HIP_TRY(hipMemcpy(x1.data(), src1[q] + o, sizeof(double) * n, hipMemcpyDeviceToHost));
            VRT_HIP_TRY(hipMemcpy(x2.data(), src2[q] + o, sizeof(double) * n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; i++) {
                dst[q][2 * i] = x1[(size_t)i];
                dst[q][2 * i + 1] = x2[(size_t)i];
            }
        }
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

struct vrt_schedule {
    vrt::AngleSchedule s;
};

int vrt_schedule_build(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, vrt_schedule **out)
{
    if (!g || !up || !out) return fail(VRT_EINVAL, "NULL argument");
    if (n_sweeps < 1) return fail(VRT_EINVAL, "n_sweeps must be >= 1");
    *out = nullptr;
    try {
        const int64_t n = g->n;
        std::vector<int32_t> u1((size_t)n), u2((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t a = up[2 * i], b = up[2 * i + 1];
            if (a > n || b > n) return fail(VRT_EINVAL, "upwind id out of range");
            u1[(size_t)i] = a >= 1 ? (int32_t)(a - 1) : kNoUpwind;
            u2[(size_t)i] = b >= 1 ? (int32_t)(b - 1) : kNoUpwind;
        }
        vrt_schedule *sc = new vrt_schedule();
        build_angle_schedule(direction_of(g, dir), dir > 0, n, n_sweeps, u1.data(), u2.data(), sc->s);
        if (sc->s.bad_site >= 0) {
            std::string msg = "site " + std::to_string(sc->s.bad_site + 1) + " has no upwind neighbour";
            delete sc;
            return fail(VRT_EGRID, msg);
        }
        *out = sc;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

int64_t vrt_schedule_num_nodes(const vrt_schedule *s) { return s ? (int64_t)s->s.site.size() : 0; }
int64_t vrt_schedule_num_levels(const vrt_schedule *s) { return s ? (int64_t)s->s.level_off.size() - 1 : 0; }

int vrt_schedule_get(const vrt_schedule *s, int64_t *site, int32_t *zflags, int64_t *level_off)
{
    if (!s) return fail(VRT_EINVAL, "NULL schedule");
    if (site)
        for (size_t i = 0; i < s->s.site.size(); i++) site[i] = (int64_t)s->s.site[i] + 1;
    if (zflags)
        for (size_t i = 0; i < s->s.zflags.size(); i++) zflags[i] = s->s.zflags[i];
    if (level_off) std::copy(s->s.level_off.begin(), s->s.level_off.end(), level_off);
    return VRT_OK;
}

void vrt_schedule_destroy(vrt_schedule *s) { delete s; }

int vrt_layer_schedule(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, uint32_t *vis,
                       int32_t *nlev, int64_t *n_visits)
{
    if (!g || !up || !vis || !nlev) return fail(VRT_EINVAL, "NULL argument");
    if (n_sweeps < 1) return fail(VRT_EINVAL, "n_sweeps must be >= 1");
    try {
        const int64_t n = g->n;
        std::vector<int32_t> u1((size_t)n), u2((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t a = up[2 * i], b = up[2 * i + 1];
            if (a > n || b > n) return fail(VRT_EINVAL, "upwind id out of range");
            u1[(size_t)i] = a >= 1 ? (int32_t)(a - 1) : kNoUpwind;
            u2[(size_t)i] = b >= 1 ? (int32_t)(b - 1) : kNoUpwind;
        }
        LayerSchedule ls;
        build_layer_schedule(direction_of(g, dir), dir > 0, n, n_sweeps, u1.data(), u2.data(), ls);
        if (ls.bad_site >= 0)
            return fail(VRT_EGRID, "site " + std::to_string(ls.bad_site + 1) + " has no upwind neighbour");
        if (!ls.ok) return fail(VRT_EINVAL, "schedule does not fit the packed layer-tile encoding");
        std::copy(ls.vis.begin(), ls.vis.end(), vis);
        std::copy(ls.nlev.begin(), ls.nlev.end(), nlev);
        if (n_visits) *n_visits = ls.n_visits;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

int vrt_layer_sorted_slots(const vrt_grid *g, int dir, const uint32_t *vis, int64_t *store, int64_t *self)
{
    if (!g || !vis || !store || !self) return fail(VRT_EINVAL, "NULL argument");
    try {
        const int64_t n = g->n;
        const Direction &d = direction_of(g, dir);
        std::vector<uint32_t> v(vis, vis + n);
        std::vector<int32_t> s32;
        build_sorted_slots(d, n, v, s32);
        for (int64_t i = 0; i < n; i++) {
            store[i] = (int64_t)d.store[(size_t)i] + 1;
            self[i] = (int64_t)s32[(size_t)i];
        }
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

struct vrt_patch_schedule {
    vrt::PatchSchedule s;
    vrt::LayerSchedule layers;       // the builder's by-product (what plan creation uses as the angle's layer schedule)
};

int vrt_patch_schedule_build(const vrt_grid *g, int dir, const int64_t *up, int n_sweeps, int own_target,
                             int entry_cap, vrt_patch_schedule **out, int64_t counts[6])
{
    if (!g || !up || !out || !counts) return fail(VRT_EINVAL, "NULL argument");
    if (n_sweeps < 1 || own_target < 1 || entry_cap < 1 || entry_cap > 65535)
        return fail(VRT_EINVAL, "need n_sweeps >= 1, own_target >= 1 and 1 <= entry_cap <= 65535");
    *out = nullptr;
    try {
        const int64_t n = g->n;
        std::vector<int32_t> u1((size_t)n), u2((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t a = up[2 * i], b = up[2 * i + 1];
            if (a > n || b > n) return fail(VRT_EINVAL, "upwind id out of range");
            u1[(size_t)i] = a >= 1 ? (int32_t)(a - 1) : kNoUpwind;
            u2[(size_t)i] = b >= 1 ? (int32_t)(b - 1) : kNoUpwind;
        }
        vrt_patch_schedule *ps = new vrt_patch_schedule();
        // (introspection: VRT_HOST_THREADS sets the builder's thread count -- the sanitizer screen drives it with 16)
        int threads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        if (const char *e = std::getenv("VRT_HOST_THREADS")) threads = std::max(1, std::min(64, std::atoi(e)));
        build_patch_schedule(direction_of(g, dir), dir > 0, n, n_sweeps, u1.data(), u2.data(), own_target, entry_cap,
                             ps->s, threads, &ps->layers);
        if (ps->s.bad_site >= 0) {
            const std::string msg = "site " + std::to_string(ps->s.bad_site + 1) + " has no upwind neighbour";
            delete ps;
            return fail(VRT_EGRID, msg);
        }
        if (!ps->s.ok) {
            delete ps;
            return fail(VRT_EINVAL, "schedule does not fit the packed patch encoding");
        }
        counts[0] = (int64_t)ps->s.patch_own_lo.size();
        counts[1] = (int64_t)ps->s.entry_pos.size();
        counts[2] = ps->s.n_visits;
        counts[3] = ps->s.n_live;
        counts[4] = ps->s.max_entries;
        counts[5] = (int64_t)ps->s.layer_patch_off.size();
        *out = ps;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    }
}

int vrt_patch_schedule_get(const vrt_patch_schedule *s, int32_t *layer_patch_off, int32_t *patch_own_lo,
                           int32_t *patch_own_cnt, int32_t *patch_nlev, int64_t *patch_ent_off,
                           int32_t *entry_pos, uint32_t *entry_vis, uint32_t *entry_loc)
{
    if (!s) return fail(VRT_EINVAL, "NULL schedule");
    const PatchSchedule &p = s->s;
    if (layer_patch_off) std::copy(p.layer_patch_off.begin(), p.layer_patch_off.end(), layer_patch_off);
    if (patch_own_lo) std::copy(p.patch_own_lo.begin(), p.patch_own_lo.end(), patch_own_lo);
    if (patch_own_cnt) std::copy(p.patch_own_cnt.begin(), p.patch_own_cnt.end(), patch_own_cnt);
    if (patch_nlev) std::copy(p.patch_nlev.begin(), p.patch_nlev.end(), patch_nlev);
    if (patch_ent_off) std::copy(p.patch_ent_off.begin(), p.patch_ent_off.end(), patch_ent_off);
    if (entry_pos) std::copy(p.entry_pos.begin(), p.entry_pos.end(), entry_pos);
    if (entry_vis) std::copy(p.entry_vis.begin(), p.entry_vis.end(), entry_vis);
    if (entry_loc) std::copy(p.entry_loc.begin(), p.entry_loc.end(), entry_loc);
    return VRT_OK;
}

int vrt_patch_schedule_get_deps(const vrt_patch_schedule *s, int64_t *dep_off, int32_t *dep_list)
{
    if (!s) return fail(VRT_EINVAL, "NULL schedule");
    const PatchSchedule &p = s->s;
    if (dep_off) std::copy(p.dep_off.begin(), p.dep_off.end(), dep_off);
    if (dep_list) std::copy(p.dep_list.begin(), p.dep_list.end(), dep_list);
    return VRT_OK;
}

int vrt_patch_schedule_get_layers(const vrt_patch_schedule *s, uint32_t *vis, int32_t *nlev, int64_t *n_visits)
{
    if (!s) return fail(VRT_EINVAL, "NULL schedule");
    const LayerSchedule &ls = s->layers;
    if (!ls.ok) return fail(VRT_EINVAL, "schedule does not fit the packed layer-tile encoding");
    if (vis) std::copy(ls.vis.begin(), ls.vis.end(), vis);
    if (nlev) std::copy(ls.nlev.begin(), ls.nlev.end(), nlev);
    if (n_visits) *n_visits = ls.n_visits;
    return VRT_OK;
}

void vrt_patch_schedule_destroy(vrt_patch_schedule *s) { delete s; }

int vrt_grid_get_storage_order(const vrt_grid *g, int dir, int64_t *out)
{
    if (!g || !out) return fail(VRT_EINVAL, "NULL argument");
    const Direction &d = direction_of(g, dir);
    for (int64_t i = 0; i < g->n; i++) out[i] = (int64_t)d.store[(size_t)i] + 1;
    return VRT_OK;
}

int vrt_plan_native_pair_block(const vrt_plan *p) { return p ? 1 << native_lg(p, false) : 0; }
int vrt_plan_native_pair_block_f32(const vrt_plan *p) { return p ? 1 << native_lg(p, true) : 0; }

int64_t vrt_plan_native_alpha_count(const vrt_plan *p, int64_t nlam)
{
    if (!p || nlam < 1) return 0;
    return (int64_t)p->A * ((nlam + 1) / 2 * 2) * p->g->n;
}

int vrt_plan_alpha_to_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dalpha,
                                 double *dalpha_native, void *stream)
{
    DeviceScope scope;
    if (!p || !dalpha || !dalpha_native) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (p->A != (int)p->n_angles_user)
        return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (rc) return rc;
        return alpha_to_native(p, nlam, ld, dalpha, dalpha_native, (hipStream_t)stream, false);
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_alpha_to_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dalpha, float *dalpha_native,
                                     void *stream)
{
    DeviceScope scope;
    if (!p || !dalpha || !dalpha_native) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (p->A != (int)p->n_angles_user)
        return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (rc) return rc;
        return alpha_to_native(p, nlam, ld, dalpha, dalpha_native, (hipStream_t)stream, true);
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_execute_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dS,
                         const double *dalpha, int alpha_mode, const double *dI0_up,
                         const double *dI0_down, const double *weights_host, double *dJ,
                         double *dI_out, void *stream)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        return execute_dev_locked(p, nlam, ld, dS, dalpha, alpha_mode, dI0_up, dI0_down,
                                  weights_host, dJ, dI_out, (hipStream_t)stream);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int64_t vrt_plan_native_plane_count(const vrt_plan *p, int64_t nlam)
{
    if (!p || nlam < 1) return 0;
    return ((nlam + 1) / 2 * 2) * p->g->n;
}

int vrt_plan_to_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *d_in, double *d_up, double *d_down, void *stream)
{
    DeviceScope scope;
    if (!p || !d_in || (!d_up && !d_down)) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (!rc) rc = native_planes_ok(p);
        if (rc) return rc;
        return planes_to_native(p, nlam, ld, d_in, d_up, d_down, (hipStream_t)stream);
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_from_native_dev(vrt_plan *p, int dir, int64_t nlam, int64_t ld, const double *d_native, double *d_out, void *stream)
{
    DeviceScope scope;
    if (!p || !d_native || !d_out) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (!rc) rc = native_planes_ok(p);
        if (rc) return rc;
        return plane_from_native(p, dir > 0 ? 0 : 1, nlam, ld, d_native, d_out, (hipStream_t)stream);
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_j_from_native_dev(vrt_plan *p, int64_t nlam, int64_t ld, const double *dJ_up, const double *dJ_down, double *dJ,
                               void *stream)
{
    DeviceScope scope;
    if (!p || !dJ || (!dJ_up && !dJ_down)) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        int rc = use_device(p->g->device);
        if (!rc) rc = native_planes_ok(p);
        if (rc) return rc;
        return J_from_native(p, nlam, ld, dJ_up, dJ_down, dJ, (hipStream_t)stream);
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_execute_native_dev(vrt_plan *p, int64_t nlam, const double *dS_up, const double *dS_down, const double *dalpha,
                                int alpha_mode, const double *dI0_up, const double *dI0_down, const double *weights_host,
                                double *dJ_up, double *dJ_down, void *stream)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        return execute_native_locked(p, nlam, dS_up, dS_down, dalpha, alpha_mode, dI0_up, dI0_down, weights_host, dJ_up, dJ_down,
                                     (hipStream_t)stream);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_to_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *d_in, float *d_up, float *d_down, void *stream)
{
    return native_f32_call(p, nlam, ld, d_in && (d_up || d_down),
                           [&] { return planes_to_native_f32(p, nlam, ld, d_in, d_up, d_down, (hipStream_t)stream); });
}

int vrt_plan_from_native_dev_f32(vrt_plan *p, int dir, int64_t nlam, int64_t ld, const float *d_native, float *d_out, void *stream)
{
    return native_f32_call(p, nlam, ld, d_native && d_out,
                           [&] { return plane_from_native_f32(p, dir > 0 ? 0 : 1, nlam, ld, d_native, d_out, (hipStream_t)stream); });
}

int vrt_plan_j_from_native_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dJ_up, const float *dJ_down, float *dJ,
                                   void *stream)
{
    return native_f32_call(p, nlam, ld, dJ && (dJ_up || dJ_down),
                           [&] { return J_from_native_f32(p, nlam, ld, dJ_up, dJ_down, dJ, (hipStream_t)stream); });
}

int vrt_plan_execute_native_dev_f32(vrt_plan *p, int64_t nlam, const float *dS_up, const float *dS_down, const float *dalpha,
                                    int alpha_mode, const float *dI0_up, const float *dI0_down, const double *weights_host,
                                    float *dJ_up, float *dJ_down, void *stream)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        return execute_native_locked(p, nlam, dS_up, dS_down, dalpha, alpha_mode, dI0_up, dI0_down, weights_host, dJ_up, dJ_down,
                                     (hipStream_t)stream, /*f32=*/true);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_execute_dev_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dS,
                             const float *dalpha, int alpha_mode, const float *dI0_up,
                             const float *dI0_down, const double *weights_host, float *dJ,
                             float *dI_out, void *stream)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        return execute_dev_locked(p, nlam, ld, dS, dalpha, alpha_mode, dI0_up, dI0_down, weights_host,
                                  dJ, dI_out, (hipStream_t)stream, /*f32=*/true);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_execute(vrt_plan *p, int64_t nlam, int64_t ld, const double *S, const double *alpha,
                     int alpha_mode, const double *I0_up, const double *I0_down,
                     const double *weights, double *J, double *I_out)
{
    DeviceScope scope;
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    if (!S || !alpha) return fail(VRT_EINVAL, "S and alpha must not be NULL");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (alpha_mode < 0 || alpha_mode > 2) return fail(VRT_EINVAL, "bad alpha_mode (host arrays: 0, 1 or 2)");
    try {
        std::lock_guard<std::mutex> lock(p->mu);
        vrt_grid *g = p->g;
        int rc = use_device(g->device);
        if (rc) return rc;
        const size_t n = (size_t)g->n;
        const size_t nS = n * (size_t)ld;
        const size_t nA = alpha_mode == VRT_ALPHA_SITE ? n
                          : alpha_mode == VRT_ALPHA_SITE_LAM ? nS
                                                             : nS * (size_t)p->n_angles_user;
        const size_t nU = (size_t)g->up.n1 * (size_t)nlam, nD = (size_t)g->down.n1 * (size_t)nlam;
        hipStream_t st = g->stream;
        if ((rc = ensure(p->d_stage[0], p->stage_cap[0], nS))) return rc;
        if ((rc = ensure(p->d_stage[1], p->stage_cap[1], nA))) return rc;
        VRT_HIP_TRY(hipMemcpyAsync(p->d_stage[0], S, sizeof(double) * nS, hipMemcpyHostToDevice, st));
        VRT_HIP_TRY(hipMemcpyAsync(p->d_stage[1], alpha, sizeof(double) * nA, hipMemcpyHostToDevice, st));
        double *dU = nullptr, *dD = nullptr, *dJ = nullptr;
        if (I0_up && nU) {
            if ((rc = ensure(p->d_stage[2], p->stage_cap[2], nU))) return rc;
            dU = p->d_stage[2];
            VRT_HIP_TRY(hipMemcpyAsync(dU, I0_up, sizeof(double) * nU, hipMemcpyHostToDevice, st));
        }
        if (I0_down && nD) {
            if ((rc = ensure(p->d_stage[3], p->stage_cap[3], nD))) return rc;
            dD = p->d_stage[3];
            VRT_HIP_TRY(hipMemcpyAsync(dD, I0_down, sizeof(double) * nD, hipMemcpyHostToDevice, st));
        }
        if (J) {
            if ((rc = ensure(p->d_stage[4], p->stage_cap[4], nS))) return rc;
            dJ = p->d_stage[4];
        }
        double *dIo = nullptr;
        if (I_out) {
            if ((rc = ensure(p->d_stage[5], p->stage_cap[5], nS * (size_t)p->n_angles_user))) return rc;
            dIo = p->d_stage[5];
        }
        rc = execute_dev_locked(p, nlam, ld, p->d_stage[0], p->d_stage[1], alpha_mode, dU, dD,
                                weights, dJ, dIo, st);
        if (rc) return rc;
        if (J) VRT_HIP_TRY(hipMemcpyAsync(J, dJ, sizeof(double) * nS, hipMemcpyDeviceToHost, st));
        if (I_out)   // (nlam, n, n_angles) with leading dimension ld, skipped angles already zeroed
            VRT_HIP_TRY(hipMemcpyAsync(I_out, dIo, sizeof(double) * nS * (size_t)p->n_angles_user,
                                       hipMemcpyDeviceToHost, st));
        VRT_HIP_TRY(hipStreamSynchronize(st));
        return patch_chain_check(p);         // a chained sweep that gave up waiting: THIS call's results are invalid
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_plan_check(vrt_plan *p)
{
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    std::lock_guard<std::mutex> lock(p->mu);
    return patch_chain_check(p);
}

int vrt_plan_last_sweep_timing(const vrt_plan *p, double *ms, int64_t *launches)
{
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    if (!p->ev_valid) return fail(VRT_EINVAL, "no execute has run on this plan yet");
    VRT_HIP_TRY(hipEventSynchronize(p->ev1));
    float t = 0.f;
    VRT_HIP_TRY(hipEventElapsedTime(&t, p->ev0, p->ev1));
    if (ms) *ms = (double)t;
    if (launches) *launches = p->last_launches;
    // the sweep has finished: a chained launch that gave up waiting for a dependency is reported here (or by the next execute)
    return patch_chain_check(const_cast<vrt_plan *>(p));
}

int vrt_plan_last_path(const vrt_plan *p) { return p ? p->last_path : 0; }

int vrt_plan_set_option(vrt_plan *p, const char *name, const char *value)
{
    if (!p) return fail(VRT_EINVAL, "NULL plan");
    std::lock_guard<std::mutex> lock(p->mu);
    return tuning_set(p->tune, name, value, /*created=*/true);
}

int vrt_grid_set_option(vrt_grid *g, const char *name, const char *value)
{
    if (!g) return fail(VRT_EINVAL, "NULL grid");
    Tuning probe;
    int rc = tuning_set(probe, name, value, /*created=*/false);      // validates name and value
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g->mu);
    bool found = false;
    for (auto &o : g->options)
        if (o.first == name) { o.second = value; found = true; }
    if (!found) g->options.emplace_back(name, value);
    for (PlanCacheEntry *c : g->cache) {                             // cached single-solve plans follow where they can
        std::lock_guard<std::mutex> plock(c->plan->mu);
        (void)tuning_set(c->plan->tune, name, value, /*created=*/true);
    }
    return VRT_OK;
}

int vrt_lambda_update_dev(vrt_grid *g, int64_t nlam, int64_t ld, const double *dJ, const double *dB,
                          const double *deps, const double *dS_old, double *dS_new, double *max_rel_change,
                          void *stream)
{
    DeviceScope scope;
    if (!g || !dJ || !dB || !deps || !dS_old || !dS_new || !max_rel_change)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    int rc = use_device(g->device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g->mu);
    if (!g->d_scalars) VRT_HIP_TRY(hipMalloc((void **)&g->d_scalars, 2 * sizeof(unsigned long long)));
    unsigned long long *d_res = g->d_scalars;
    rc = launch_lambda_update(g->n, nlam, ld, dJ, dB, deps, dS_old, dS_new, d_res, st);
    unsigned long long h[2] = {0, 0};
    if (!rc && hipMemcpyAsync(h, d_res, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) rc = VRT_ENODEVICE;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = VRT_ENODEVICE;
    if (rc) return rc == VRT_ENODEVICE ? fail(rc, "HIP error in vrt_lambda_update_dev") : rc;
    double d;
    std::memcpy(&d, &h[0], sizeof(double));
    *max_rel_change = h[1] ? std::nan("") : d;
    return VRT_OK;
}

int vrt_lambda_update_native_dev(vrt_grid *g, int64_t nlam, const double *dJ_up, const double *dJ_down, const double *dB_up,
                                 const double *deps, double *dS_up, double *dS_down, double *max_rel_change, void *stream)
{
    DeviceScope scope;
    if (!g || (!dJ_up && !dJ_down) || !dB_up || !deps || !dS_up || !dS_down || !max_rel_change)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1) return fail(VRT_EINVAL, "need nlam >= 1");
    int rc = use_device(g->device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g->mu);
    if (!g->d_scalars) VRT_HIP_TRY(hipMalloc((void **)&g->d_scalars, 2 * sizeof(unsigned long long)));
    unsigned long long *d_res = g->d_scalars;
    rc = launch_lambda_update_native(g, nlam, dJ_up, dJ_down, dB_up, deps, dS_up, dS_down, d_res, st);
    unsigned long long h[2] = {0, 0};
    if (!rc && hipMemcpyAsync(h, d_res, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) rc = VRT_ENODEVICE;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = VRT_ENODEVICE;
    if (rc) return rc == VRT_ENODEVICE ? fail(rc, "HIP error in vrt_lambda_update_native_dev") : rc;
    double d;
    std::memcpy(&d, &h[0], sizeof(double));
    *max_rel_change = h[1] ? std::nan("") : d;
    return VRT_OK;
}

// wavelength-sized host arrays -> the grid's device scratch (caller holds g->mu).  The scratch is
// shared by successive calls, possibly on different streams: the copy first waits for the kernel of
// the previous call that read it (small_ev, recorded by small_done after that launch).  The values
// are staged in a pinned host buffer of the grid, so nothing here drains the caller's stream; the
// host only waits for its OWN previous copy out of that buffer (small_copy_ev, long done by then).
static int upload_small(vrt_grid *g, const std::vector<double> &h, hipStream_t st)
{
    if (g->small_ev_valid) VRT_HIP_TRY(hipStreamWaitEvent(st, g->small_ev, 0));
    if (g->small_copy_valid) VRT_HIP_TRY(hipEventSynchronize(g->small_copy_ev));
    if (!g->d_small || g->small_cap < h.size()) {
        if (g->small_ev_valid) VRT_HIP_TRY(hipEventSynchronize(g->small_ev));   // about to free what it read
        dev_free(g->d_small);
        if (g->h_small) { (void)hipHostFree(g->h_small); g->h_small = nullptr; }
        g->small_cap = 0;
        const size_t cap = std::max<size_t>(2 * h.size(), 256);
        int rc = dev_alloc(&g->d_small, cap);
        if (rc) return rc;
        VRT_HIP_TRY(hipHostMalloc((void **)&g->h_small, sizeof(double) * cap, hipHostMallocDefault));
        g->small_cap = cap;
    }
    std::memcpy(g->h_small, h.data(), sizeof(double) * h.size());
    VRT_HIP_TRY(hipMemcpyAsync(g->d_small, g->h_small, sizeof(double) * h.size(), hipMemcpyHostToDevice, st));
    if (!g->small_copy_ev) VRT_HIP_TRY(hipEventCreateWithFlags(&g->small_copy_ev, hipEventDisableTiming));
    VRT_HIP_TRY(hipEventRecord(g->small_copy_ev, st));
    g->small_copy_valid = true;
    return VRT_OK;
}

static int small_done(vrt_grid *g, hipStream_t st)
{
    if (!g->small_ev) VRT_HIP_TRY(hipEventCreateWithFlags(&g->small_ev, hipEventDisableTiming));
    VRT_HIP_TRY(hipEventRecord(g->small_ev, st));
    g->small_ev_valid = true;
    return VRT_OK;
}

static int line_opacity_impl(vrt_plan *p, int64_t nlam, const double *lambda, double lambda0, double c0,
                             const double *d_velocity, const double *d_doppler_width, const double *d_gamma,
                             const double *d_line_strength, const double *d_alpha_cont, void *d_alpha_native,
                             void *stream, bool f32_out)
{
    DeviceScope scope;
    if (!p || !lambda || !d_velocity || !d_doppler_width || !d_gamma || !d_line_strength || !d_alpha_cont ||
        !d_alpha_native)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1) return fail(VRT_EINVAL, "nlam must be >= 1");
    if (!(lambda0 > 0) || !(c0 > 0)) return fail(VRT_EINVAL, "lambda0 and c0 must be positive");
    try {
        vrt_grid *g = p->g;
        int rc = use_device(g->device);
        if (rc) return rc;
        if (!p->patch_ok && (!p->tile_ok || p->tile_max_layer_size > steps_max_layer(false)))
            return fail(VRT_EINVAL, "the native alpha layout needs a layer path (at most 4 visits per site and 255 levels per layer)");
        std::lock_guard<std::mutex> lock(g->mu);
        std::vector<double> h(lambda, lambda + nlam);
        if ((rc = upload_small(g, h, (hipStream_t)stream))) return rc;
        rc = launch_line_opacity(p, nlam, g->d_small, lambda0, c0, d_velocity, d_doppler_width, d_gamma,
                                 d_line_strength, d_alpha_cont, d_alpha_native, (hipStream_t)stream, f32_out);
        return rc ? rc : small_done(g, (hipStream_t)stream);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_line_opacity_dev(vrt_plan *p, int64_t nlam, const double *lambda, double lambda0, double c0,
                         const double *d_velocity, const double *d_doppler_width, const double *d_gamma,
                         const double *d_line_strength, const double *d_alpha_cont, double *d_alpha_native,
                         void *stream)
{
    return line_opacity_impl(p, nlam, lambda, lambda0, c0, d_velocity, d_doppler_width, d_gamma, d_line_strength,
                             d_alpha_cont, d_alpha_native, stream, false);
}

int vrt_line_opacity_dev_f32(vrt_plan *p, int64_t nlam, const double *lambda, double lambda0, double c0,
                             const double *d_velocity, const double *d_doppler_width, const double *d_gamma,
                             const double *d_line_strength, const double *d_alpha_cont, float *d_alpha_native,
                             void *stream)
{
    return line_opacity_impl(p, nlam, lambda, lambda0, c0, d_velocity, d_doppler_width, d_gamma, d_line_strength,
                             d_alpha_cont, d_alpha_native, stream, true);
}

static int rates_populations_impl(vrt_grid *g, int64_t nlam, int64_t ld, const double *lambda,
                              const int64_t blocks[6], const double *dJ, const double *dJ_up, const double *dJ_down, const double *planck2,
                              double lambda0, double c0, const double *d_doppler_width, const double *d_gamma,
                              double sigma_bb_const, const double *sigma_bf1, const double *sigma_bf2,
                              const double *d_temperature, const double *d_lte_populations, double hc_over_kB,
                              double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                              double *d_R, double *d_populations, void *stream)
{
    DeviceScope scope;
    if (!g || !lambda || !blocks || (!dJ && !dJ_up && !dJ_down) || !planck2 || !d_doppler_width || !d_gamma || !sigma_bf1 || !sigma_bf2 ||
        !d_temperature || !d_lte_populations || !d_C || !d_atom_density || !d_R || !d_populations)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 2 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 2 and ld >= nlam");
    for (int b = 0; b < 3; b++)
        if (blocks[2 * b] < 0 || blocks[2 * b + 1] > nlam || blocks[2 * b + 1] - blocks[2 * b] < 2)
            return fail(VRT_EINVAL, "each wavelength block needs at least two wavelengths inside [0, nlam)");
    try {
        int rc = use_device(g->device);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(g->mu);
        std::vector<double> h;
        h.insert(h.end(), lambda, lambda + nlam);
        h.insert(h.end(), planck2, planck2 + nlam);
        h.insert(h.end(), sigma_bf1, sigma_bf1 + (blocks[3] - blocks[2]));
        h.insert(h.end(), sigma_bf2, sigma_bf2 + (blocks[5] - blocks[4]));
        if ((rc = upload_small(g, h, (hipStream_t)stream))) return rc;
        rc = launch_rates_populations(g, nlam, ld, blocks, g->d_small, dJ, lambda0, c0, d_doppler_width,
                                      d_gamma, sigma_bb_const, d_temperature, d_lte_populations, hc_over_kB,
                                      pref_ij, pref_ji, d_C, d_atom_density, d_R, d_populations,
                                      (hipStream_t)stream, dJ_up, dJ_down);
        return rc ? rc : small_done(g, (hipStream_t)stream);
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_rates_populations_dev(vrt_grid *g, int64_t nlam, int64_t ld, const double *lambda,
                              const int64_t blocks[6], const double *dJ, const double *planck2,
                              double lambda0, double c0, const double *d_doppler_width, const double *d_gamma,
                              double sigma_bb_const, const double *sigma_bf1, const double *sigma_bf2,
                              const double *d_temperature, const double *d_lte_populations, double hc_over_kB,
                              double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                              double *d_R, double *d_populations, void *stream)
{
    if (!dJ) return fail(VRT_EINVAL, "NULL argument");
    return rates_populations_impl(g, nlam, ld, lambda, blocks, dJ, nullptr, nullptr, planck2, lambda0, c0, d_doppler_width, d_gamma,
                                  sigma_bb_const, sigma_bf1, sigma_bf2, d_temperature, d_lte_populations, hc_over_kB, pref_ij, pref_ji,
                                  d_C, d_atom_density, d_R, d_populations, stream);
}

int vrt_rates_populations_native_dev(vrt_grid *g, int64_t nlam, const double *lambda, const int64_t blocks[6],
                                     const double *dJ_up, const double *dJ_down, const double *planck2,
                                     double lambda0, double c0, const double *d_doppler_width, const double *d_gamma,
                                     double sigma_bb_const, const double *sigma_bf1, const double *sigma_bf2,
                                     const double *d_temperature, const double *d_lte_populations, double hc_over_kB,
                                     double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                                     double *d_R, double *d_populations, void *stream)
{
    if (!dJ_up && !dJ_down) return fail(VRT_EINVAL, "NULL argument");
    return rates_populations_impl(g, nlam, nlam, lambda, blocks, nullptr, dJ_up, dJ_down, planck2, lambda0, c0, d_doppler_width, d_gamma,
                                  sigma_bb_const, sigma_bf1, sigma_bf2, d_temperature, d_lte_populations, hc_over_kB, pref_ij, pref_ji,
                                  d_C, d_atom_density, d_R, d_populations, stream);
}

static int single_solve(vrt_grid *g, int dir, const double k[3], const double *S, const double *I0,
                        int64_t nI0, const double *alpha, int n_sweeps, double *I_out)
{
    DeviceScope scope;
    if (!g || !k || !S || !alpha || !I_out) return fail(VRT_EINVAL, "NULL argument");
    const Direction &d = direction_of(g, dir);
    if (nI0 != d.n1)   // Julia: DimensionMismatch at irregular_ray_tracing.jl:35 / :116
        return fail(VRT_EINVAL, "I_0 has " + std::to_string(nI0) + " elements, the boundary layer has " +
                                    std::to_string(d.n1));
    if (nI0 > 0 && !I0) return fail(VRT_EINVAL, "I_0 is NULL");
    try {
        // Only the cache lookup / insertion / eviction runs under the grid's mutex; the solve itself
        // takes the plan's own mutex (vrt_plan_execute), so concurrent callers with different
        // directions (the reference calls these from Threads.@threads) overlap, and callers of the
        // same direction queue on that plan's workspaces.  An entry in use is never evicted.
        PlanCacheEntry *entry = nullptr;
        {
            std::lock_guard<std::mutex> lock(g->mu);
            for (PlanCacheEntry *c : g->cache)
                if (c->n_sweeps == n_sweeps * dir && c->k[0] == k[0] && c->k[1] == k[1] && c->k[2] == k[2])
                    entry = c;
            if (!entry) {
                int dirs[1] = {dir};
                vrt_plan *plan = nullptr;
                int rc = plan_create_impl(g, 1, k, dirs, n_sweeps, &plan, &g->options);
                if (rc) return rc;
                if (g->cache.size() >= 64)      // drop the oldest entry nobody is using
                    for (size_t i = 0; i < g->cache.size(); i++)
                        if (g->cache[i]->users == 0) {
                            vrt_plan_destroy(g->cache[i]->plan);
                            delete g->cache[i];
                            g->cache.erase(g->cache.begin() + (long)i);
                            break;
                        }
                entry = new PlanCacheEntry{{k[0], k[1], k[2]}, n_sweeps * dir, plan, 0};
                g->cache.push_back(entry);
            }
            entry->users++;
        }
        const double one = 1.0;
        // I_out doubles as J with weight 1: J = 0 + 1*I is exact
        const int rc = vrt_plan_execute(entry->plan, 1, 1, S, alpha, VRT_ALPHA_SITE, dir > 0 ? I0 : nullptr,
                                        dir > 0 ? nullptr : I0, &one, I_out, nullptr);
        {
            std::lock_guard<std::mutex> lock(g->mu);
            entry->users--;
        }
        return rc;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_delaunay_up(vrt_grid *g, const double k[3], const double *S, const double *I0, int64_t nI0,
                    const double *alpha, int n_sweeps, double *I_out)
{
    return single_solve(g, +1, k, S, I0, nI0, alpha, n_sweeps, I_out);
}

int vrt_delaunay_down(vrt_grid *g, const double k[3], const double *S, const double *I0,
                      int64_t nI0, const double *alpha, int n_sweeps, double *I_out)
{
    return single_solve(g, -1, k, S, I0, nI0, alpha, n_sweeps, I_out);
}

}  // extern "C"
