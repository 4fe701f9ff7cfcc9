// Patch schedule of the fused layer kernel (vrt_patch.hip): a BFS layer is cut into PATCHES of
// consecutive storage positions (a piece of a strip of the storage order, vrt_grid.cpp: compact in x, y) and every patch is solved by its
// own workgroup, with no communication between workgroups.
//
// The reference's serial sweep (src/irregular_ray_tracing.jl:37-80, :118-161) couples the sites of a
// layer through their in-layer upwind neighbours.  build_layer_schedule (vrt_schedule.cpp) turns
// that into live visits with in-layer levels; every live visit reads, per in-layer upwind, the value
// written by ONE specific earlier visit of that neighbour (or the tile's initial zero).  Following
// those read-from edges backwards from the final visit of every site a patch OWNS gives the exact
// set of visits the patch's results depend on -- its dependency cone inside the layer.  A patch's
// workgroup executes that set (its own sites plus a HALO of neighbouring sites' visits, recomputed
// redundantly) in level order on its private LDS tile: the same arithmetic on the same values as
// the serial sweep, so the results equal those of the unsplit layer bit for bit, for any layer size.
//
// Everything here is LOCAL to a layer: the trace of a layer's visits, which of them are redundant, live, at which
// level they run and how the layer is cut into patches depend on sites of other layers only through constants
// (an upwind in another layer never changes while the layer is swept).  So the layers of an angle are analysed
// independently -- by several host threads when the caller has them to spare -- with arrays of the layer's size, and
// stitched together in layer order.  Only the dependency lists BETWEEN patches (what the chained launch waits on) look
// across layers; they are formed at the end.
#include <algorithm>
#include <atomic>
#include <new>

#include "vrt_internal.h"

namespace vrt {

namespace {

struct LayerPatches {
    std::vector<int32_t> own_lo, own_cnt, nlev;
    std::vector<int64_t> ent_off;            // local: entries of local patch q are [ent_off[q], ent_off[q+1])
    std::vector<int32_t> entry_pos;
    std::vector<uint32_t> entry_vis, entry_loc;
    std::vector<uint32_t> vis;               // by local id: the site's 4 x 8-bit in-layer visit levels (LayerSchedule::vis)
    int32_t layer_nlev = 0;                  // levels of the unsplit layer (LayerSchedule::nlev)
    int64_t n_visits = 0, n_live = 0, max_entries = 0, bad_site = -1;
    bool ok = true;
    bool levels_ok = false;                  // passes A-C went through: vis / layer_nlev / n_live are the layer schedule's
};

// one layer: sites at storage positions [lo, hi); `pos_of` = storage position of a site (global array)
void build_layer_patches(const Direction &dir, bool ascending, int n_sweeps, const int32_t *up1, const int32_t *up2,
                         int own_target, int entry_cap, int64_t layer, const int32_t *pos_of, LayerPatches &out)
{
    const std::vector<int64_t> &r = dir.reduced;
    const int32_t lo = (int32_t)(r[(size_t)layer - 1] - 1), hi = (int32_t)(r[(size_t)layer] - 1);
    const int32_t m = hi - lo;
    out.ent_off.push_back(0);
    if (m <= 0) return;
    // local id of an in-layer site = its storage position - lo; -1 for a site of another layer
    // (the never-visited last site is stored just past the last layer's range whatever layer it is counted to: a constant too)
    auto local = [&](int32_t site) -> int32_t {
        const int32_t q = pos_of[(size_t)site];
        return dir.layer_of[(size_t)site] == layer && q >= lo && q < hi ? q - lo : -1;
    };
    std::vector<int32_t> l1((size_t)m), l2((size_t)m);       // local ids of the two upwinds (by local id of the site)
    for (int32_t p = lo; p < hi; p++) {
        const int32_t i = dir.store[(size_t)p];
        const int32_t u1 = up1[i], u2 = up2[i];
        if (u1 < 0 || u2 < 0) {
            if (out.bad_site < 0 || i < out.bad_site) out.bad_site = i;
            l1[(size_t)(p - lo)] = l2[(size_t)(p - lo)] = -1;
            continue;
        }
        l1[(size_t)(p - lo)] = local(u1);
        l2[(size_t)(p - lo)] = local(u2);
    }
    if (out.bad_site >= 0) {
        out.ok = false;
        return;
    }
    // ---- pass A: the layer's trace, redundant visits dropped, with read-from edges --------------------------
    std::vector<uint32_t> t_site;            // local id of the visit's site
    std::vector<int32_t> t_src1, t_src2;     // trace index of the in-layer visit whose value input r reads; -1: none
    t_site.reserve((size_t)m * 2);
    t_src1.reserve((size_t)m * 2);
    t_src2.reserve((size_t)m * 2);
    std::vector<uint32_t> ver((size_t)m, 0), seen1((size_t)m, UINT32_MAX), seen2((size_t)m, UINT32_MAX);
    std::vector<int32_t> lastw((size_t)m, -1);
    for (int sweep = 0; sweep < n_sweeps; sweep++)
        for (int32_t t = 0; t < m; t++) {
            const int64_t posn = ascending ? (int64_t)lo + t : (int64_t)hi - 1 - t;      // sweep order (irregular_ray_tracing.jl:41 / :122)
            const int32_t i = (int32_t)(dir.perm[(size_t)posn] - 1);
            const int32_t li = pos_of[(size_t)i] - lo;
            const int32_t a = l1[(size_t)li], b = l2[(size_t)li];
            // an upwind of another layer never changes while this layer is swept: version 0 throughout
            const uint32_t v1 = a >= 0 ? ver[(size_t)a] : 0u, v2 = b >= 0 ? ver[(size_t)b] : 0u;
            if (seen1[(size_t)li] == v1 && seen2[(size_t)li] == v2) continue;
            seen1[(size_t)li] = v1;
            seen2[(size_t)li] = v2;
            ver[(size_t)li]++;
            t_src1.push_back(a >= 0 ? lastw[(size_t)a] : -1);
            t_src2.push_back(b >= 0 ? lastw[(size_t)b] : -1);
            lastw[(size_t)li] = (int32_t)t_site.size();
            t_site.push_back((uint32_t)li);
        }
    const size_t T = t_site.size();
    // ---- pass B: liveness (the final value of every site is needed) -------------------------------------------
    std::vector<uint8_t> live(T, 0);
    {
        std::vector<uint8_t> needed((size_t)m, 1);
        for (size_t x = T; x-- > 0;) {
            const uint32_t li = t_site[x];
            if (!needed[li]) continue;
            live[x] = 1;
            needed[li] = 0;
            if (t_src1[x] >= 0) needed[(size_t)l1[li]] = 1;
            if (t_src2[x] >= 0) needed[(size_t)l2[li]] = 1;
        }
    }
    // ---- pass C: in-layer levels (as build_layer_schedule): only their fit into the packed encoding matters here ---
    {
        std::vector<int32_t> lw((size_t)m, 0), lr((size_t)m, 0);
        std::vector<uint8_t> nvis((size_t)m, 0);
        out.vis.assign((size_t)m, 0u);
        for (size_t x = 0; x < T; x++) {
            if (!live[x]) continue;
            const uint32_t li = t_site[x];
            const int32_t a = l1[li], b = l2[li];
            int32_t lv = std::max(lw[li], lr[li]);
            if (a >= 0) lv = std::max(lv, lw[(size_t)a]);
            if (b >= 0) lv = std::max(lv, lw[(size_t)b]);
            lv += 1;
            if (a >= 0) lr[(size_t)a] = std::max(lr[(size_t)a], lv);
            if (b >= 0) lr[(size_t)b] = std::max(lr[(size_t)b], lv);
            lw[li] = lv;
            lr[li] = 0;
            if (lv > 255 || nvis[li] >= 4) {
                out.ok = false;              // does not fit the packed encoding (the level kernels handle it)
                return;
            }
            out.vis[li] |= (uint32_t)lv << (8 * nvis[li]);
            nvis[li]++;
            out.n_live++;
            out.layer_nlev = std::max(out.layer_nlev, lv);
        }
        out.levels_ok = true;
    }
    // ---- patches and their dependency cones ---------------------------------------------------------------------
    // Greedy along the storage order: a patch takes the next site as long as the union of the cones of
    // its sites stays within entry_cap entries (and it owns at most own_target sites), so every
    // workgroup's lanes are filled whatever the halo width of the angle and the layer.
    std::vector<int32_t> last_live((size_t)m, -1);   // trace index of the site's final visit
    for (size_t x = 0; x < T; x++)
        if (live[x]) last_live[t_site[x]] = (int32_t)x;
    std::vector<int32_t> stamp(T, -1);               // patch that has marked the visit
    std::vector<int32_t> slot_of((size_t)m, -1), in_patch((size_t)m, -1);   // local tile slot of a site / patch it is an entry of
    std::vector<int32_t> stack, marked, sites, new_marks, new_sites, halo, llw, llr;
    int32_t patch_id = 0;
    int32_t p = 0;                                    // local position
    while (p < m) {
        const int32_t own0 = p;
        marked.clear();
        sites.clear();
        while (p < m && p - own0 < own_target) {
            // cone of the next site: its final visit + everything it (transitively) reads inside the layer
            new_marks.clear();
            new_sites.clear();
            stack.clear();
            const int32_t xs = last_live[(size_t)p];
            if (xs >= 0 && stamp[(size_t)xs] != patch_id) {
                stamp[(size_t)xs] = patch_id;
                stack.push_back(xs);
            }
            while (!stack.empty()) {
                const int32_t x = stack.back();
                stack.pop_back();
                new_marks.push_back(x);
                const int32_t li = (int32_t)t_site[(size_t)x];
                if (in_patch[(size_t)li] != patch_id) {
                    in_patch[(size_t)li] = patch_id;
                    new_sites.push_back(li);
                }
                const int32_t s1 = t_src1[(size_t)x], s2 = t_src2[(size_t)x];
                if (s1 >= 0 && stamp[(size_t)s1] != patch_id) { stamp[(size_t)s1] = patch_id; stack.push_back(s1); }
                if (s2 >= 0 && stamp[(size_t)s2] != patch_id) { stamp[(size_t)s2] = patch_id; stack.push_back(s2); }
            }
            if ((int64_t)sites.size() + (int64_t)new_sites.size() > entry_cap) {
                for (int32_t x : new_marks) stamp[(size_t)x] = -1;          // does not fit any more: next patch
                for (int32_t li : new_sites) in_patch[(size_t)li] = -1;
                break;
            }
            marked.insert(marked.end(), new_marks.begin(), new_marks.end());
            sites.insert(sites.end(), new_sites.begin(), new_sites.end());
            p++;
        }
        const int32_t own_cnt = p - own0;
        if (own_cnt == 0) {                              // the cone of a single site exceeds the cap: the
            out.ok = false;                              // patch kernel cannot hold it (other paths take over)
            return;
        }
        // entries: owned sites in storage order, then the halo sites by storage position
        halo.clear();
        for (int32_t li : sites)
            if (li < own0 || li >= p) halo.push_back(li);
        std::sort(halo.begin(), halo.end());
        const int64_t entries = (int64_t)own_cnt + (int64_t)halo.size();
        const int64_t e0 = (int64_t)out.entry_pos.size();
        for (int32_t j = 0; j < own_cnt; j++) {
            slot_of[(size_t)(own0 + j)] = j;
            in_patch[(size_t)(own0 + j)] = patch_id;
            out.entry_pos.push_back(lo + own0 + j);
        }
        for (size_t j = 0; j < halo.size(); j++) {
            slot_of[(size_t)halo[j]] = own_cnt + (int32_t)j;
            out.entry_pos.push_back(lo + halo[j]);
        }
        out.entry_vis.resize(out.entry_pos.size(), 0u);
        out.entry_loc.resize(out.entry_pos.size(), 0u);
        // Levels INSIDE the patch: the cone's visits in trace order, ordered by the same read-after-write,
        // write-after-read and write-after-write rules as the layer's levels but among themselves only --
        // a patch does not wait for levels in which nothing of its cone happens.  Packed increasing.
        std::sort(marked.begin(), marked.end());
        llw.assign((size_t)entries, 0);
        llr.assign((size_t)entries, 0);
        int32_t nlev = 0;
        for (int32_t x : marked) {
            const int32_t li = (int32_t)t_site[(size_t)x];
            const int32_t si = slot_of[(size_t)li];
            const int32_t a = l1[(size_t)li], b = l2[(size_t)li];
            const int32_t c1 = (a >= 0 && in_patch[(size_t)a] == patch_id) ? slot_of[(size_t)a] : -1;
            const int32_t c2 = (b >= 0 && in_patch[(size_t)b] == patch_id) ? slot_of[(size_t)b] : -1;
            int32_t lv = std::max(llw[(size_t)si], llr[(size_t)si]);
            if (c1 >= 0) lv = std::max(lv, llw[(size_t)c1]);
            if (c2 >= 0) lv = std::max(lv, llw[(size_t)c2]);
            lv += 1;
            if (c1 >= 0) llr[(size_t)c1] = std::max(llr[(size_t)c1], lv);
            if (c2 >= 0) llr[(size_t)c2] = std::max(llr[(size_t)c2], lv);
            llw[(size_t)si] = lv;
            llr[(size_t)si] = 0;
            uint32_t &v = out.entry_vis[(size_t)(e0 + si)];
            int sh = 0;
            while (sh < 32 && ((v >> sh) & 0xFFu)) sh += 8;
            v |= (uint32_t)lv << sh;
            nlev = std::max(nlev, lv);
        }
        for (int64_t e = e0; e < (int64_t)out.entry_pos.size(); e++) {
            const int32_t li = out.entry_pos[(size_t)e] - lo;
            uint32_t l[2];
            for (int q = 0; q < 2; q++) {
                const int32_t u = q == 0 ? l1[(size_t)li] : l2[(size_t)li];
                l[q] = (u >= 0 && in_patch[(size_t)u] == patch_id) ? (uint32_t)slot_of[(size_t)u] : 0xFFFFu;
            }
            out.entry_loc[(size_t)e] = l[0] | (l[1] << 16);
        }
        out.own_lo.push_back(lo + own0);
        out.own_cnt.push_back(own_cnt);
        out.nlev.push_back(nlev);
        out.ent_off.push_back((int64_t)out.entry_pos.size());
        out.n_visits += (int64_t)marked.size();
        out.max_entries = std::max<int64_t>(out.max_entries, entries);
        patch_id++;
    }
}

}  // namespace

void build_patch_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps, const int32_t *up1,
                          const int32_t *up2, int own_target, int entry_cap, PatchSchedule &out, int threads,
                          LayerSchedule *layers)
{
    const std::vector<int64_t> &r = dir.reduced;
    const int64_t nl = (int64_t)r.size();           // layers 1 .. nl-1
    out = PatchSchedule();
    out.layer_patch_off.assign((size_t)nl + 1, 0);
    out.ok = true;
    if (own_target < 1) own_target = 1;
    if (entry_cap < 1) entry_cap = 1;
    if (entry_cap > 65535) entry_cap = 65535;
    std::vector<int32_t> pos_of((size_t)n);          // site -> storage position
    for (int64_t p = 0; p < n; p++) pos_of[(size_t)dir.store[(size_t)p]] = (int32_t)p;

    // ---- the layers, independently (dealt to the threads by an atomic counter: layers differ in size) -----------
    const int64_t first = 2, last = nl - 1;          // layers first .. last are swept
    const int64_t count = std::max<int64_t>(0, last - first + 1);
    std::vector<LayerPatches> lp((size_t)count);
    std::atomic<int64_t> next(0);
    auto worker = [&](int) {
        for (;;) {
            const int64_t j = next.fetch_add(1);
            if (j >= count) break;
            build_layer_patches(dir, ascending, n_sweeps, up1, up2, own_target, entry_cap, first + j, pos_of.data(), lp[(size_t)j]);
        }
    };
    const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>(threads, count));
    if (nthr <= 1) worker(0);
    else if (!run_workers(nthr, worker)) throw std::bad_alloc();
    // ---- stitched together in layer order ----------------------------------------------------------------------------
    for (int64_t j = 0; j < count; j++)
        if (lp[(size_t)j].bad_site >= 0 && (out.bad_site < 0 || lp[(size_t)j].bad_site < out.bad_site)) out.bad_site = lp[(size_t)j].bad_site;
    if (layers) {
        // by-product: passes A-C above ARE build_layer_schedule's analysis (vrt_schedule.cpp), layer by layer
        LayerSchedule &ls = *layers;
        ls = LayerSchedule();
        ls.bad_site = out.bad_site;
        ls.nlev.assign((size_t)nl, 0);
        bool all = out.bad_site < 0;
        for (int64_t j = 0; j < count && all; j++) all = lp[(size_t)j].levels_ok || lp[(size_t)j].vis.empty();
        for (int64_t layer = first; layer <= last; layer++)
            ls.max_layer_size = std::max(ls.max_layer_size, r[(size_t)layer] - r[(size_t)layer - 1]);
        if (all) {
            ls.vis.assign((size_t)n, 0u);
            for (int64_t j = 0; j < count; j++) {
                const LayerPatches &l = lp[(size_t)j];
                const int64_t lo = r[(size_t)(first + j) - 1] - 1;
                for (size_t li = 0; li < l.vis.size(); li++) ls.vis[(size_t)dir.store[(size_t)lo + li]] = l.vis[li];
                ls.nlev[(size_t)(first + j)] = l.layer_nlev;
                ls.n_visits += l.n_live;
            }
            ls.ok = true;
        }
    }
    if (out.bad_site >= 0) {
        out.ok = false;
        return;
    }
    for (int64_t j = 0; j < count; j++)
        if (!lp[(size_t)j].ok) {
            out.ok = false;
            return;
        }
    int32_t patch_id = 0;
    for (int64_t j = 0; j < count; j++) {
        LayerPatches &l = lp[(size_t)j];
        out.layer_patch_off[(size_t)(first + j)] = patch_id;
        const int64_t e_base = (int64_t)out.entry_pos.size();
        for (size_t q = 0; q < l.own_lo.size(); q++) {
            out.patch_own_lo.push_back(l.own_lo[q]);
            out.patch_own_cnt.push_back(l.own_cnt[q]);
            out.patch_nlev.push_back(l.nlev[q]);
            out.patch_ent_off.push_back(e_base + l.ent_off[q]);
            patch_id++;
        }
        out.entry_pos.insert(out.entry_pos.end(), l.entry_pos.begin(), l.entry_pos.end());
        out.entry_vis.insert(out.entry_vis.end(), l.entry_vis.begin(), l.entry_vis.end());
        out.entry_loc.insert(out.entry_loc.end(), l.entry_loc.begin(), l.entry_loc.end());
        out.n_visits += l.n_visits;
        out.n_live += l.n_live;
        out.max_entries = std::max(out.max_entries, l.max_entries);
        l = LayerPatches();
    }
    for (int64_t layer = std::max<int64_t>(nl, 2); layer <= nl; layer++) out.layer_patch_off[(size_t)layer] = patch_id;
    out.patch_ent_off.push_back((int64_t)out.entry_pos.size());

    // ---- dependencies BETWEEN patches: the owners of the earlier-layer intensities a patch gathers ---------------------
    // (an upwind in an EARLIER layer is read from memory as the final value its owner patch stored; one in a later
    // layer reads as 0, irregular_ray_tracing.jl:23, :75; the boundary layer and the never-visited last site have no
    // owner): the patches a patch waits for when the layers are chained inside one launch (vrt_patch.hip: k_patch_chain)
    std::vector<int32_t> owner((size_t)n, -1);       // storage position -> patch
    for (int32_t q = 0; q < patch_id; q++)
        for (int32_t j = 0; j < out.patch_own_cnt[(size_t)q]; j++) owner[(size_t)(out.patch_own_lo[(size_t)q] + j)] = q;
    // (patches are independent here too: contiguous ranges of them per thread, concatenated in patch order)
    const int dthr = (int)std::max<int64_t>(1, std::min<int64_t>(threads, patch_id / 256 + 1));
    std::vector<std::vector<int32_t>> part_list((size_t)dthr);
    std::vector<std::vector<int32_t>> part_cnt((size_t)dthr);
    auto dep_worker = [&](int t) {
        const int32_t q0 = (int32_t)((int64_t)patch_id * t / dthr), q1 = (int32_t)((int64_t)patch_id * (t + 1) / dthr);
        std::vector<int32_t> deps;
        std::vector<int32_t> &list = part_list[(size_t)t], &cnt = part_cnt[(size_t)t];
        cnt.reserve((size_t)(q1 - q0));
        int64_t layer = first;
        for (int32_t q = q0; q < q1; q++) {
            while (layer < last && q >= out.layer_patch_off[(size_t)layer + 1]) layer++;
            deps.clear();
            for (int64_t e = out.patch_ent_off[(size_t)q]; e < out.patch_ent_off[(size_t)q + 1]; e++) {
                const int32_t i = dir.store[(size_t)out.entry_pos[(size_t)e]];
                for (int qq = 0; qq < 2; qq++) {
                    const int32_t u = qq == 0 ? up1[i] : up2[i];
                    if (dir.layer_of[(size_t)u] < layer && owner[(size_t)pos_of[(size_t)u]] >= 0) deps.push_back(owner[(size_t)pos_of[(size_t)u]]);
                }
            }
            std::sort(deps.begin(), deps.end());
            deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
            list.insert(list.end(), deps.begin(), deps.end());
            cnt.push_back((int32_t)deps.size());
        }
    };
    if (dthr <= 1) dep_worker(0);
    else if (!run_workers(dthr, dep_worker)) throw std::bad_alloc();
    out.dep_off.reserve((size_t)patch_id + 1);
    out.dep_off.push_back(0);
    for (int t = 0; t < dthr; t++) {
        for (int32_t c : part_cnt[(size_t)t]) out.dep_off.push_back(out.dep_off.back() + c);
        out.dep_list.insert(out.dep_list.end(), part_list[(size_t)t].begin(), part_list[(size_t)t].end());
    }
}

}  // namespace vrt
