// Internal declarations of libvrt_hip.so (not part of the C ABI; see include/voronoirt.h).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "voronoirt.h"

namespace vrt {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define VRT_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::vrt::fail(VRT_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Restores the caller's current HIP device when an entry point returns (a multi-GPU host such as
// torch keeps its own notion of the current device; the library must not change it behind its back).
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceScope()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// Runs body(t) for t = 0 .. count-1 on `count` host threads.  A worker must not throw (an exception that leaves a
// std::thread's function is std::terminate for the whole host process -- Julia or Python): every body runs inside a
// try block, and a thread that cannot be created has its share done by the calling thread.  Returns false when a
// body threw (std::bad_alloc in practice: the callers report VRT_ENOMEM).
template <typename Body>
inline bool run_workers(int count, Body body)
{
    std::vector<char> threw((size_t)std::max(count, 1), 0);
    auto guarded = [&](int t) {
        try {
            body(t);
        } catch (...) {
            threw[(size_t)t] = 1;
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve((size_t)std::max(count, 1));
    } catch (...) {
    }
    for (int t = 0; t < count; t++) {
        try {
            pool.emplace_back(guarded, t);
        } catch (...) {
            guarded(t);
        }
    }
    for (auto &th : pool) th.join();
    for (int t = 0; t < count; t++)
        if (threw[(size_t)t]) return false;
    return true;
}

constexpr int kMaxAngles = 64;          // active angles per plan (kernel-argument weight table)
constexpr int64_t kMaxGuess = 70;       // read_cell's neighbour cap, voronoi_utils.jl:42
constexpr int32_t kNoUpwind = -1;

// One sweep direction of the grid (up: from the z_min wall, down: from the z_max wall).
struct Direction {
    std::vector<int32_t> layer_of;   // BFS layer per site, 1-based (voronoi_utils.jl:93-174)
    std::vector<int64_t> perm;       // stable sortperm, 1-based site ids (:72,77)
    std::vector<int64_t> reduced;    // reduce_layers offsets, 1-based, r[end] = n (:253-269)
    int64_t n1 = 0;                  // reduced[1] - 1: sites that receive I_0
    int32_t *d_order = nullptr;      // device copy of perm, 0-based int32: sweep position -> site
    int32_t *d_rank = nullptr;       // inverse: site -> sweep position
    int32_t *d_lay = nullptr;        // 0-based layer boundaries: layer l = [lay[l-1], lay[l]), L+1 entries
    // STORAGE order of the layer-tile path: layers contiguous exactly like the sweep order, but
    // inside a layer the sites are sorted in strips of rows (vrt_grid.cpp; VRT_STORE_ORDER=morton: along a Morton curve over (x, y)) so that spatial
    // neighbours (and therefore upwind gathers) share cache lines; the never-visited last site
    // perm[n] stays at position n-1.  The Gauss-Seidel ORDER is unaffected (it lives in the schedule).
    std::vector<int32_t> store;      // storage position -> site (0-based)
    int32_t *d_store = nullptr;
    int32_t *d_srank = nullptr;      // site -> storage position
};

struct PlanCacheEntry;

// Host <-> device lanes of the host-pointer entry points (vrt_lambda.cpp: vrt_plan_execute_line): per lane a copy stream
// and two pinned staging buffers, filled / drained by a host thread of its own, so that the caller's pageable arrays
// travel at PCIe speed (a pageable hipMemcpy stages through ONE thread: 0.4 GB of S took 20 ms each way)
struct CopyLane {
    hipStream_t st = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
constexpr size_t kCopyChunk = (size_t)8 << 20;      // bytes per staging buffer

// Tuning options of a plan (vrt_plan_set_option / vrt_grid_set_option).  Defaults; an environment variable of the
// option's name presets it, read ONCE when the plan is created (never during an execute).  Results do not
// depend on any of them (the parity tests run the paths and shapes against each other).
struct Tuning {
    int path = 0;                 // VRT_PATH: 0 auto, 1 levels, 2 tiles, 3 steps, 4 patches
    int step_K = 0;               // VRT_STEP_K: sites per thread of the layer-step level kernels (0: fewest that fit)
    int step_single = 0;          // VRT_STEP_SINGLE: 1 = the single-wavelength level kernel on any grid
    int step_pairs = 0;           // VRT_STEP_PAIRS: wavelength pairs per coefficient thread (0: 4 to 6)
    int step_xcd = 2;             // VRT_STEP_XCD: block -> XCD map of the coefficient kernel
    int step_streams = 2;         // VRT_STEP_STREAMS: internal streams of the layer paths (1..4)
    int step_level_map = 1;       // VRT_STEP_LEVEL_MAP: level workgroups of an angle on one XCD
    int step_group_dir = 1;       // VRT_STEP_GROUP_DIR: one direction per stream when balanced
    int tile_wide = 1;            // VRT_TILE_WIDE, VRT_TILE_PRE: variants of the persistent tile path
    int tile_pre = 1;
    int graph = 0;                // VRT_GRAPH: replay the level launches as a hipGraph
    int patch_K = 1, patch_NT = 512;   // VRT_PATCH_K, VRT_PATCH_NT: entries per thread, threads (plan creation only)
    int patch_own = 0;            // VRT_PATCH_OWN: owned sites per patch at most (0: as many as fit; creation only)
    int patch_Q = 1;              // VRT_PATCH_Q: wavelength pairs a patch workgroup solves at a time
    int pair_block = 1;           // VRT_PAIR_BLOCK: wavelength pairs of a site kept side by side in the patch path's
                                  //   storage layout, 1 / 2 / 4 / 8 / 16 (vrt_device.h; creation only: the native
                                  //   per-angle alpha of the plan is laid out with it)
    int patch_quad = 1;           // VRT_PATCH_QUAD: fp32 storage in blocks of >= 2 pairs, four wavelengths per lane
                                  //   (k_patch_quad; creation only: the native float alpha is laid out with it)
    int patch_lean = 1;           // VRT_PATCH_LEAN: the 64-register form of the (1, 1, NT) kernel (four workgroups per CU;
                                  //   default for two and more wavelength pairs)
    int patch_chain = 2;          // VRT_PATCH_CHAIN: every layer inside ONE chained launch (k_patch_chain) instead of one launch
                                  //   per layer and direction: 0 never, 1 wherever it exists, 2 = auto: where a layer
                                  //   alone cannot fill the chip (patch_chain_possible)
    int chain_pairs = 9;          // VRT_CHAIN_PAIRS: wavelength-pair blocks an item of the chained launch solves at most
    int chain_spin = 2048;        // VRT_CHAIN_SPIN: polls (x 1024) after which a waiting workgroup gives up (~2 s)
    int chain_dataflag = 2;       // VRT_CHAIN_DATAFLAG: the chained launch's intensities as their own flags: 0 never, 1 wherever
                                  //   the kernel exists, 2 auto (one or two wavelength pairs: the planes are filled per step)
    int patch_split = 0;          // VRT_PATCH_SPLIT: workgroups per item of a per-layer launch, fixed (0: from VRT_PATCH_TARGET)
    int patch_target = 0;         // VRT_PATCH_TARGET: workgroups per launch aimed at when the pair steps of an item are split
                                  //   (0: a balanced split chosen per launch, launch_patch_layer)
    int chain_static = 1;         // VRT_CHAIN_STATIC: the progress-word form of the chained launch maps block -> item statically (0: tickets)
    int lambda_native = 1;        // VRT_LAMBDA_NATIVE: the Λ-iteration session keeps S and J in sweep order between its steps
                                  //   (read when a session is created; 0: the caller's layout, two layout changes per iteration)
    int debug_flags = 0, debug_skip_levels = 0, tile_debug = 0;   // timing diagnostics (-DVRT_DIAG build only)
};
void tuning_from_env(Tuning &t);
// VRT_OK, or VRT_EINVAL for an unknown name / bad value; `created`: the plan exists already (creation-only options fail)
int tuning_set(Tuning &t, const char *name, const char *value, bool created);

// arguments a captured level-launch graph was recorded with
struct SweepKey {
    int64_t nlam = -1, ldS = 0, ldA = 0, ldI = 0;
    const void *S = nullptr, *alpha = nullptr, *I = nullptr;
    int alpha_mode = -1;
    bool f32 = false;
    bool operator==(const SweepKey &o) const
    {
        return nlam == o.nlam && ldS == o.ldS && ldA == o.ldA && ldI == o.ldI && S == o.S &&
               alpha == o.alpha && I == o.I && alpha_mode == o.alpha_mode && f32 == o.f32;
    }
};

}  // namespace vrt

struct vrt_grid {
    int device = 0;
    int64_t n = 0;
    int64_t D = 0;                   // maximum(neighbours[:,1])
    double bounds[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> pos;         // (3, n) z,x,y
    std::vector<int32_t> rowptr;     // CSR over the neighbour lists, n+1
    std::vector<int32_t> col;        // 1-based ids, walls <= 0, row order preserved
    vrt::Direction up, down;
    // device mirrors
    double *d_pos = nullptr;
    int32_t *d_rowptr = nullptr;
    int32_t *d_col = nullptr;
    double *d_lz = nullptr, *d_lx = nullptr, *d_ly = nullptr;   // Delaunay lines, CSR-packed SoA
    hipStream_t stream = nullptr;
    unsigned long long *d_scalars = nullptr;   // scratch of the Λ-iteration epilogue's reduction
    double *d_small = nullptr;                 // wavelength-sized host arrays of the physics kernels (λ, 2hc²/λ⁵, σ_bf)
    size_t small_cap = 0;
    double *h_small = nullptr;                 // pinned staging buffer of the same capacity
    hipEvent_t small_ev = nullptr;             // last kernel that read d_small
    bool small_ev_valid = false;
    hipEvent_t small_copy_ev = nullptr;        // last copy out of h_small
    bool small_copy_valid = false;
    // options applied to the single-solve plans cached below (vrt_grid_set_option)
    std::vector<std::pair<std::string, std::string>> options;
    // cache of single-angle plans for vrt_delaunay_up/down
    std::mutex mu;
    std::vector<vrt::PlanCacheEntry *> cache;
};

struct vrt_plan {
    vrt_grid *g = nullptr;
    vrt::Tuning tune;
    int n_sweeps = 3;
    int64_t n_angles_user = 0;
    int A = 0;                          // active angles (k[0] != 0)
    std::vector<int> user_of_active;    // active index -> user angle index
    std::vector<int> dir_of_active;     // +1 up, -1 down
    std::vector<double> k;              // (3, A)
    // per-angle upwind tables, [A][n]
    int32_t *d_up1 = nullptr, *d_up2 = nullptr;     // 0-based ids, kNoUpwind if none
    double *d_d1 = nullptr, *d_d2 = nullptr;         // dot products (smallest_angle's `dots`)
    double *d_w1 = nullptr, *d_w2 = nullptr;         // dot_weights, irregular_ray_tracing.jl:51
    double *d_r1 = nullptr, *d_r2 = nullptr;         // euclidean path lengths, :66
    // level schedule, merged over the active angles
    uint32_t *d_node_site = nullptr;    // site id (0-based)
    uint32_t *d_node_meta = nullptr;    // active angle | zero-read flags
    int32_t *d_node_u1 = nullptr, *d_node_u2 = nullptr;   // the node's upwind site ids (copied next to it)
    std::vector<int64_t> level_off;     // nodes of level t are [level_off[t], level_off[t+1])
    bool level_ready = false;           // the level schedule is built on first use
    std::vector<int32_t> h_up1, h_up2;  // host copies of the upwind ids (schedule building)
    int64_t n_nodes = 0;
    // per-direction lists of active angle indices (for the boundary kernel)
    int32_t *d_angles_up = nullptr, *d_angles_down = nullptr;
    int n_up = 0, n_down = 0;
    std::vector<int64_t> skip_site;     // per active angle: never-updated site perm[n] (0-based)
    // workspaces (grow-only)
    double *d_I = nullptr;
    size_t I_cap = 0;
    int64_t I_ld = 0;
    double *d_stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // S, alpha, I0up, I0down, J, I_out
    size_t stage_cap[6] = {0, 0, 0, 0, 0, 0};
    // layer paths (vrt_layers.hip, vrt_tables.hip): tables in storage order, per-layer level counts
    bool tile_ok = false;
    bool step_tables_ready = false;      // t_self ... t_code_ss exist (ensure_step_tables)
    int tile_K = 8;                      // sites per thread of the 1024-thread workgroup
    int tile_max_layers = 0;
    int64_t tile_max_layer_size = 0;
    int64_t tile_visits = 0;
    int32_t *t_u1 = nullptr, *t_u2 = nullptr;
    double *t_w1 = nullptr, *t_w2 = nullptr, *t_r1 = nullptr, *t_r2 = nullptr;
    uint32_t *t_vis = nullptr, *t_loc = nullptr;
    // layer-step level kernel: sites of a layer dealt to threads sorted by visit pattern
    int32_t *t_self = nullptr;           // [A][n] sorted index (absolute) -> storage position
    uint32_t *t_vis_s = nullptr, *t_loc_s = nullptr;   // t_vis / t_loc in sorted order
    uint32_t *t_gpos = nullptr;          // [A][n] compact in-layer coupling list (k_gpos)
    int32_t *t_rank_s = nullptr;         // [A][n] storage position -> sorted index (inverse of t_self)
    uint32_t *t_loc_ss = nullptr;        // [A][n] upwind tile slots of the sorted entries, in SORTED terms
    uint32_t *t_code_ss = nullptr;       // [A][n] two-launch tile path: upwind slot + kind codes (layers <= 4096 sites)
    int32_t *d_nlev = nullptr, *d_angle_dir = nullptr;
    std::vector<int64_t> angle_visits;   // surviving visits per active angle (task cost)
    std::vector<int32_t> h_task_map;     // block -> angle | wavelength << 8
    int32_t *d_task_map = nullptr;
    size_t task_map_cap = 0;
    int task_map_nlam = -1;
    double *ws_S[2] = {nullptr, nullptr}, *ws_A[2] = {nullptr, nullptr}, *ws_J[2] = {nullptr, nullptr};
    size_t ws_S_cap[2] = {0, 0}, ws_A_cap[2] = {0, 0}, ws_J_cap[2] = {0, 0};
    double *ws_AA = nullptr;
    size_t ws_AA_cap = 0;
    double *ws_cg[2] = {nullptr, nullptr};   // layer-step coefficient buffers: constant terms, compact couplings
    size_t ws_cg_cap[2] = {0, 0};
    // layer-step path: internal streams and the angle groups they advance through the layers
    int step_groups = 0;
    int32_t *d_step_angles = nullptr;
    std::vector<int> step_group_off;
    std::vector<double> angle_mean_levels;   // mean in-layer level count per active angle (level-kernel task cost)
    int32_t *d_level_map = nullptr;      // block -> task of the layer-step level kernels, per stream group (build_level_map)
    std::vector<int> level_map_off;      //   offsets of the groups' maps (+ end); 8 ceil-blocks each
    int level_map_units = 0, level_map_groups = 0;
    std::vector<int32_t> h_step_angles;  // host copy of d_step_angles
    // fused patch path (vrt_patch.hip): per-angle patch schedules, concatenated over the active angles
    bool patch_ok = false;
    int lg_pair_block = 0;           // log2 of the pairs per block of the patch path's storage layout (vrt_device.h)
    int patch_cap = 0, patch_K = 0, patch_NT = 0;   // entries per patch <= cap = K * NT (fixed at creation)
    int64_t n_patches = 0, n_patch_entries = 0, n_patch_visits = 0;
    std::vector<int32_t> h_patch_first;  // [A][tile_max_layers + 2]: index of the first patch of (angle, layer)
    std::vector<int4> h_patch_rec;       // per patch: first entry, entries, first owned position, owned sites
    int32_t *e_pos = nullptr, *e_u1 = nullptr, *e_u2 = nullptr;
    uint32_t *e_vis = nullptr, *e_loc = nullptr;
    double *e_w1 = nullptr, *e_w2 = nullptr, *e_r1 = nullptr, *e_r2 = nullptr;
    std::vector<int2> h_patch_rec2;      // per patch: levels, active angle
    std::vector<int64_t> h_patch_dep_off;   // per patch: the patches (plan-wide indices) whose stored intensities it gathers
    std::vector<int32_t> h_patch_deps;      //   (vrt_patch.cpp: dep_list) -- what the chained launch waits on
    // chained launch (vrt_patch.hip: k_patch_chain): items per XCD queue, dependency lists, progress words
    int4 *d_chain_items = nullptr;
    int32_t *d_chain_deps = nullptr;
    uint32_t *d_chain_progress = nullptr, *d_chain_ctrl = nullptr;
    uint32_t *h_chain_status = nullptr, *d_chain_status = nullptr;   // mapped host word: a launch gave up
    void *d_chain_dev = nullptr, *h_chain_dev_pinned = nullptr;       // the launch's argument block (ChainDev) and its staging copy
    std::vector<char> h_chain_dev;                                    //   what the device copy holds
    hipEvent_t chain_dev_ev = nullptr;
    bool chain_dev_ev_valid = false;
    size_t chain_progress_cap = 0;
    bool chain_progress_fresh = false;   // allocated since the last launch: zero it on the launch stream
    int chain_q_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int chain_npair = -1, chain_lgB = -1, chain_nsplit = -1, chain_reduce = -1;
    int64_t chain_items = 0;
    // item sets of other (pair count, block, split, reduce) combinations this plan has run: a caller that alternates
    // wavelength counts, or J and no J, switches between them instead of rebuilding (and freeing: a device synchronisation)
    struct ChainSet {
        int npair = -1, lgB = -1, nsplit = -1, reduce = -1;
        int4 *items = nullptr;
        int32_t *deps = nullptr;
        int q_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int64_t n_items = 0;
    };
    std::vector<ChainSet> chain_cache;
    uint32_t chain_epoch = 0;
    int4 *d_patch_work = nullptr;        // work lists of the launches: two int4 per slot (ensure_patch_work, PatchArgs::wrec)
    std::vector<int64_t> patch_work_off; //   [group][layer] offsets into it
    int patch_work_groups = 0;
    hipStream_t step_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t step_fork = nullptr, step_join[4] = {nullptr, nullptr, nullptr, nullptr};
    int last_path = 0;                   // 1 = level kernels, 2 = layer tiles
    // hipGraph of the level-launch sequence, replayed while the arguments stay the same
    hipGraphExec_t graph_exec = nullptr;
    vrt::SweepKey graph_key;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    std::vector<vrt::CopyLane> copy_lanes;      // host-pointer entry points (ensure_copy_lanes)
    hipEvent_t copy_done = nullptr;
    int64_t last_launches = 0;
    // sweep-order ("native") S and J of the call in progress (vrt_plan_execute_native_dev; set and cleared under `mu`):
    // per sweep direction (0 up, 1 down) a plane set [pairs][n][2] in that direction's storage order, read / written in
    // place by the layer paths instead of the workspaces behind the layout changes
    const void *nat_S[2] = {nullptr, nullptr};
    void *nat_J[2] = {nullptr, nullptr};
    bool nat_mode = false;
    std::mutex mu;
};

namespace vrt {

struct PlanCacheEntry {
    double k[3];
    int n_sweeps;
    vrt_plan *plan;
    int users;          // single solves running on the plan right now (guarded by the grid's mutex)
};

// ---- host-side grid preparation (vrt_grid.cpp) ---------------------------------------------
int parse_neighbour_file(const char *path, int64_t n, std::vector<int64_t> &matrix, int64_t &D1);
int build_grid_host(vrt_grid *g, int64_t n, const double *pos, const int64_t *nbr, int64_t D1,
                    const double bounds[6]);

// ---- in-process tessellation (vrt_tessellate.cpp) ------------------------------------------------
int tessellate_host(int64_t n, const double *pos, const double bounds[6], int64_t D1, int64_t *nbr_out,
                    int64_t *max_count, int nthreads);

// ---- schedule (vrt_schedule.cpp) -------------------------------------------------------------
struct AngleSchedule {
    std::vector<uint32_t> site;      // live nodes sorted by level
    std::vector<uint8_t> zflags;     // bit0: I(upwind 1) reads as zero, bit1: upwind 2
    std::vector<int64_t> level_off;  // per-angle level offsets (level 1 first)
    int64_t bad_site = -1;           // visited site without an upwind neighbour
};
void build_angle_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps,
                          const int32_t *up1, const int32_t *up2, AngleSchedule &out);
struct LayerSchedule {
    std::vector<uint32_t> vis;       // per site (original id): 4 x 8-bit in-layer visit levels
    std::vector<int32_t> nlev;       // per layer (index = 1-based layer): number of levels
    int64_t n_visits = 0;
    int64_t max_layer_size = 0;
    int64_t bad_site = -1;
    bool ok = false;                 // fits the tile kernel's encoding
};
void build_layer_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps,
                          const int32_t *up1, const int32_t *up2, LayerSchedule &out);
void build_sorted_slots(const Direction &dir, int64_t n, const std::vector<uint32_t> &vis_site,
                        std::vector<int32_t> &self);
// Patch schedule of the fused layer kernel (vrt_patch.cpp): every layer cut into ranges of
// consecutive storage positions, each with the in-layer dependency cone of its sites.
struct PatchSchedule {
    std::vector<int32_t> layer_patch_off;   // patches of layer l (1-based): [off[l], off[l+1]); L + 2 entries
    std::vector<int32_t> patch_own_lo;      // first owned storage position
    std::vector<int32_t> patch_own_cnt;     // owned sites (consecutive storage positions)
    std::vector<int32_t> patch_nlev;        // in-layer levels the patch has to run
    std::vector<int64_t> patch_ent_off;     // entries of patch q: [off[q], off[q+1]); owned first, then the halo
    std::vector<int32_t> entry_pos;         // storage position of the entry's site
    std::vector<uint32_t> entry_vis;        // up to 4 x 8-bit visit levels the patch executes for it, increasing
    std::vector<uint32_t> entry_loc;        // patch-local tile slots of its two upwinds, 16 bits each; 0xFFFF: reads 0
    std::vector<int64_t> dep_off;           // patches of EARLIER layers whose stored intensities patch q gathers:
    std::vector<int32_t> dep_list;          //   dep_list[dep_off[q] .. dep_off[q+1]), sorted (chained launch: what q waits for)
    int64_t n_visits = 0;                   // visits executed over all patches (halo visits counted per patch)
    int64_t n_live = 0;                     // live visits of the unsplit schedule
    int64_t max_entries = 0;
    int64_t bad_site = -1;
    bool ok = false;
};
// `threads`: host threads the layers of the angle are dealt to (they are analysed independently); may throw std::bad_alloc
void build_patch_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps, const int32_t *up1,
                          const int32_t *up2, int own_target, int entry_cap, PatchSchedule &out, int threads = 1,
                          LayerSchedule *layers = nullptr);
// (`layers`: also filled with what build_layer_schedule returns for the angle -- the same analysis, done layer by layer
//  here -- whenever its packed encoding fits; !layers->ok with bad_site < 0 means "ask build_layer_schedule")

// ---- device launchers (vrt_kernels.hip) ------------------------------------------------------
int launch_delaunay_lines(vrt_grid *g);
int launch_upwind_table(vrt_plan *p, int a);
struct SweepArgs {
    int64_t n, nlam, ldS, ldA, ldI;
    const void *S, *alpha;     // double, or float when f32
    int alpha_mode;
    void *I;
    bool f32 = false;          // fp32 storage of S, α, I, J (arithmetic stays fp64)
};
int launch_boundary(vrt_plan *p, const SweepArgs &sa, const void *dI0_up, const void *dI0_down,
                    hipStream_t st);
int launch_sweep_levels(vrt_plan *p, const SweepArgs &sa, hipStream_t st, int64_t *launches);
int launch_reduce_J(vrt_plan *p, const SweepArgs &sa, const double *weights_active, void *dJ,
                    int64_t ldJ, hipStream_t st);
int launch_copy_I_out(vrt_plan *p, const SweepArgs &sa, void *dI_out, int64_t ldO, hipStream_t st);

int launch_axpy(size_t count, const double *d_src, double *d_dst, hipStream_t st);
int launch_gather_rows(int64_t rows, int64_t nlam, int64_t ld, const int32_t *d_order, const double *d_src, double *d_dst,
                       hipStream_t st);
int launch_lambda_update(int64_t n, int64_t nlam, int64_t ld, const double *dJ, const double *dB,
                         const double *deps, const double *dS_old, double *dS_new,
                         unsigned long long *d_result, hipStream_t st);
// the same on sweep-order planes (J_up, J_down in, B in the up order, S_up updated in place, S_down written)
int launch_lambda_update_native(vrt_grid *g, int64_t nlam, const double *dJ_up, const double *dJ_down, const double *dB_up,
                                const double *deps, double *dS_up, double *dS_down, unsigned long long *d_result, hipStream_t st);

// ---- physics either side of the formal solve (vrt_physics.hip) -------------------------------------
int launch_line_opacity(vrt_plan *p, int64_t nlam, const double *d_lambda, double lambda0, double c0,
                        const double *d_velocity, const double *d_doppler, const double *d_gamma,
                        const double *d_strength, const double *d_alpha_cont, void *d_out, hipStream_t st,
                        bool f32_out = false);
int launch_line_terms(int64_t n, const double *d_gamma_static, const double *d_gamma_unsold, const double *d_pops,
                      double strength_const, double Bij, double Bji, double *d_gamma, double *d_strength, hipStream_t st);
int launch_rates_populations(vrt_grid *g, int64_t nlam, int64_t ld, const int64_t blocks[6],
                             const double *d_small, const double *dJ, double lambda0, double c0,
                             const double *d_doppler, const double *d_gamma, double sigma_bb_const,
                             const double *d_temperature, const double *d_lte, double hc_over_kB,
                             double pref_ij, double pref_ji, const double *d_C, const double *d_atom_density,
                             double *d_R, double *d_populations, hipStream_t st,
                             const double *dJ_up = nullptr, const double *dJ_down = nullptr);   // sweep-order J instead of dJ

int launch_rates_partial(vrt_grid *g, int64_t nlam, int64_t l0, int64_t l1, int64_t ld, const int64_t blocks[6],
                         const double *d_small, const double *dJ, double lambda0, double c0, const double *d_doppler,
                         const double *d_gamma, double sigma_bb_const, const double *d_temperature, const double *d_lte,
                         double hc_over_kB, double pref_ij, double pref_ji, double *d_shares, hipStream_t st,
                         const double *dJ_up = nullptr, const double *dJ_down = nullptr);   // this device's columns as sweep-order planes
int launch_populations_from_shares(vrt_grid *g, const double *d_shares, const double *d_C, const double *d_atom_density,
                                   double *d_R, double *d_populations, hipStream_t st);

// ---- entry-point internals shared with vrt_lambda.cpp (vrt_api.cpp) ---------------------------------
int use_device(int device);
int execute_dev_locked(vrt_plan *p, int64_t nlam, int64_t ld, const void *dS, const void *dalpha, int alpha_mode,
                       const void *dI0_up, const void *dI0_down, const double *weights, void *dJ, void *dI_out,
                       hipStream_t st, bool f32 = false);

// ---- layer paths (vrt_tables.hip, vrt_layers.hip) ----------------------------------------------------------
int ensure_step_tables(vrt_plan *p);     // sorted-slot tables of the steps / tiles paths, on first use (vrt_api.cpp)
int launch_permute_table(vrt_plan *p, int a, const uint32_t *d_vis_site);
int launch_sorted_tables(vrt_plan *p, int a);
int launch_gpos(vrt_plan *p, int a);
// log2 of the pairs per block of the plan's NATIVE per-angle alpha (and of every plane of the patch path)
// (fp32 storage: at least two pairs, so that k_patch_quad can take them as one 16-byte access)
inline int native_lg(const vrt_plan *p, bool f32)
{
    if (!p->patch_ok) return 0;
    return f32 && p->tune.patch_quad != 0 && p->patch_K == 1 ? std::max(p->lg_pair_block, 1) : p->lg_pair_block;
}
int alpha_to_native(vrt_plan *p, int64_t nlam, int64_t ld, const void *dalpha, void *out, hipStream_t st, bool f32);
// sweep-order S and J (per direction [pairs][n][2] in that direction's storage order)
int native_planes_ok(const vrt_plan *p, bool f32 = false);
int planes_to_native_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *din, float *out_up, float *out_down, hipStream_t st);
int plane_from_native_f32(vrt_plan *p, int dir_index, int64_t nlam, int64_t ld, const float *din, float *dout, hipStream_t st);
int J_from_native_f32(vrt_plan *p, int64_t nlam, int64_t ld, const float *dJ_up, const float *dJ_down, float *dJ, hipStream_t st);
int planes_to_native(vrt_plan *p, int64_t nlam, int64_t ld, const double *din, double *out_up, double *out_down, hipStream_t st);
int plane_from_native(vrt_plan *p, int dir_index, int64_t nlam, int64_t ld, const double *din, double *dout, hipStream_t st);
int J_from_native(vrt_plan *p, int64_t nlam, int64_t ld, const double *dJ_up, const double *dJ_down, double *dJ, hipStream_t st);
// the sweep with S read from / J reduced into the caller's sweep-order planes (caller holds p->mu)
int execute_native_locked(vrt_plan *p, int64_t nlam, const void *dS_up, const void *dS_down, const void *dalpha, int alpha_mode,
                          const void *dI0_up, const void *dI0_down, const double *weights, void *dJ_up, void *dJ_down,
                          hipStream_t st, bool f32 = false);
int execute_tiles(vrt_plan *p, int64_t nlam, int64_t ld, const void *dS, const void *dalpha,
                  int alpha_mode, const void *dI0_up, const void *dI0_down,
                  const double *weights_user, void *dJ, void *dI_out, hipStream_t st, bool f32);
int64_t steps_max_layer(bool f32);   // largest layer the layer-step level kernels hold

// ---- fused patch path (vrt_patch.hip) ----------------------------------------------------------------
struct TileArgs;
struct PatchReduce;
bool patch_shape_exists(int K, int Q, int NT);
int launch_patch_entries(vrt_plan *p, int a, int64_t first, int64_t count);
int ensure_patch_work(vrt_plan *p, int G, const std::vector<int32_t> &group_angles, const std::vector<int> &group_off);
int launch_patch_layer(vrt_plan *p, const TileArgs &ta, int npair, int layer, int group, int Q, hipStream_t st,
                       bool f32, const PatchReduce *reduce);
bool patch_chain_possible(const vrt_plan *p, int npair, bool f32);
bool patch_chain_dataflag(const vrt_plan *p, int npair, bool f32);
// ctrl_zeroed: the launch's control words (chain_ctrl_words() of them at p->d_chain_ctrl) were zeroed on `st` by the caller
int launch_patch_chain(vrt_plan *p, const TileArgs &ta, int npair, hipStream_t st, bool f32, const PatchReduce *reduce,
                       bool dataflag, bool ctrl_zeroed = false);
int chain_ctrl_words();
int patch_chain_check(vrt_plan *p);

}  // namespace vrt
