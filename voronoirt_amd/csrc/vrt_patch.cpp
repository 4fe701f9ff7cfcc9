// Patch schedule of the fused layer kernel (vrt_patch.hip): a BFS layer is cut into PATCHES of
// consecutive storage positions (Morton order: compact in x, y) and every patch is solved by its
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
#include <algorithm>

#include "vrt_internal.h"

namespace vrt {

namespace {

struct Trace {
    std::vector<uint32_t> site;
    std::vector<int32_t> src1, src2;     // trace index of the visit whose value input r reads; -1: initial value
    std::vector<uint8_t> lv;             // in-layer level (1..255), 0 = dead
    std::vector<int64_t> layer_off;      // visits of layer l: [layer_off[l], layer_off[l+1]), l 1-based
};

}  // namespace

void build_patch_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps, const int32_t *up1,
                          const int32_t *up2, int own_target, int entry_cap, PatchSchedule &out)
{
    const std::vector<int64_t> &r = dir.reduced;
    const int64_t nl = (int64_t)r.size();           // layers 1 .. nl-1
    out = PatchSchedule();
    out.layer_patch_off.assign((size_t)nl + 1, 0);
    out.ok = true;
    if (own_target < 1) own_target = 1;
    if (entry_cap < 1) entry_cap = 1;
    if (entry_cap > 65535) entry_cap = 65535;

    // ---- passes A (trace, redundant visits dropped) with read-from edges ----------------------
    Trace tr;
    tr.layer_off.assign((size_t)nl + 1, 0);
    std::vector<uint32_t> ver((size_t)n, 0), seen1((size_t)n, UINT32_MAX), seen2((size_t)n, UINT32_MAX);
    std::vector<int32_t> lastw((size_t)n, -1);       // trace index of the site's latest kept visit
    for (int64_t layer = 2; layer <= nl - 1; layer++) {
        tr.layer_off[(size_t)layer] = (int64_t)tr.site.size();
        const int64_t lo = r[(size_t)layer - 1] - 1, hi = r[(size_t)layer] - 1;
        for (int sweep = 0; sweep < n_sweeps; sweep++)
            for (int64_t t = 0; t < hi - lo; t++) {
                const int64_t posn = ascending ? lo + t : hi - 1 - t;
                const int64_t i = dir.perm[(size_t)posn] - 1;
                const int32_t u1 = up1[i], u2 = up2[i];
                if (u1 < 0 || u2 < 0) {
                    if (out.bad_site < 0) out.bad_site = i;
                    continue;
                }
                const uint32_t v1 = ver[(size_t)u1], v2 = ver[(size_t)u2];
                if (seen1[(size_t)i] == v1 && seen2[(size_t)i] == v2) continue;
                seen1[(size_t)i] = v1;
                seen2[(size_t)i] = v2;
                ver[(size_t)i]++;
                tr.src1.push_back(lastw[(size_t)u1]);
                tr.src2.push_back(lastw[(size_t)u2]);
                lastw[(size_t)i] = (int32_t)tr.site.size();
                tr.site.push_back((uint32_t)i);
            }
    }
    for (int64_t layer = std::max<int64_t>(nl, 2); layer <= nl; layer++) tr.layer_off[(size_t)layer] = (int64_t)tr.site.size();
    if (nl >= 1) tr.layer_off[1] = 0;
    if (out.bad_site >= 0) {
        out.ok = false;
        return;
    }
    const size_t T = tr.site.size();
    // ---- pass B: liveness ------------------------------------------------------------------------
    std::vector<uint8_t> live(T, 0);
    {
        std::vector<uint8_t> needed((size_t)n, 1);
        for (size_t x = T; x-- > 0;) {
            const uint32_t i = tr.site[x];
            if (!needed[i]) continue;
            live[x] = 1;
            needed[i] = 0;
            if (tr.src1[x] >= 0) needed[(size_t)up1[i]] = 1;
            if (tr.src2[x] >= 0) needed[(size_t)up2[i]] = 1;
        }
    }
    // ---- pass C: in-layer levels (as build_layer_schedule) -----------------------------------------
    tr.lv.assign(T, 0);
    {
        std::vector<int32_t> lw((size_t)n, 0), lr((size_t)n, 0);
        std::vector<uint8_t> nvis((size_t)n, 0);
        for (int64_t layer = 2; layer <= nl - 1; layer++)
            for (int64_t x = tr.layer_off[(size_t)layer]; x < tr.layer_off[(size_t)layer + 1]; x++) {
                if (!live[(size_t)x]) continue;
                const uint32_t i = tr.site[(size_t)x];
                const int32_t u1 = up1[i], u2 = up2[i];
                const bool in1 = dir.layer_of[(size_t)u1] == layer, in2 = dir.layer_of[(size_t)u2] == layer;
                int32_t lv = std::max(lw[i], lr[i]);
                if (in1) lv = std::max(lv, lw[(size_t)u1]);
                if (in2) lv = std::max(lv, lw[(size_t)u2]);
                lv += 1;
                if (in1) lr[(size_t)u1] = std::max(lr[(size_t)u1], lv);
                if (in2) lr[(size_t)u2] = std::max(lr[(size_t)u2], lv);
                lw[i] = lv;
                lr[i] = 0;
                if (lv > 255 || nvis[i] >= 4) {
                    out.ok = false;          // does not fit the packed encoding (the level kernels handle it)
                    return;
                }
                nvis[i]++;
                tr.lv[(size_t)x] = (uint8_t)lv;
            }
    }

    // ---- patches and their dependency cones ---------------------------------------------------------
    // Greedy along the storage order: a patch takes the next site as long as the union of the cones of
    // its sites stays within entry_cap entries (and it owns at most own_target sites), so every
    // workgroup's lanes are filled whatever the halo width of the angle and the layer.
    std::vector<int32_t> srank((size_t)n);          // site -> storage position
    for (int64_t p = 0; p < n; p++) srank[(size_t)dir.store[(size_t)p]] = (int32_t)p;
    std::vector<int32_t> last_live((size_t)n, -1);  // trace index of the site's final visit
    for (size_t x = 0; x < T; x++)
        if (live[x]) last_live[tr.site[x]] = (int32_t)x;
    std::vector<int32_t> stamp(T, -1);               // patch that has marked the visit
    std::vector<int32_t> slot_of((size_t)n, -1), in_patch((size_t)n, -1);   // local tile slot of a site / patch it is an entry of
    std::vector<int32_t> stack, marked, sites, new_marks, new_sites, halo, llw, llr, deps;
    std::vector<int32_t> owner((size_t)n, -1);      // site -> patch that owns it (stores its final intensity); -1: the
                                                    //   boundary layer and the never-visited last site (nobody stores them)
    int32_t patch_id = 0;
    out.dep_off.push_back(0);

    for (int64_t layer = 2; layer <= nl - 1; layer++) {
        out.layer_patch_off[(size_t)layer] = patch_id;
        const int32_t lo = (int32_t)(r[(size_t)layer - 1] - 1), hi = (int32_t)(r[(size_t)layer] - 1);
        const int64_t x0 = tr.layer_off[(size_t)layer];
        int32_t p = lo;
        while (p < hi) {
            const int32_t own_lo = p;
            marked.clear();
            sites.clear();
            while (p < hi && p - own_lo < own_target) {
                // cone of the next site: its final visit + everything it (transitively) reads inside the layer
                new_marks.clear();
                new_sites.clear();
                stack.clear();
                const int32_t xs = last_live[(size_t)dir.store[(size_t)p]];
                if (xs >= 0 && stamp[(size_t)xs] != patch_id) {
                    stamp[(size_t)xs] = patch_id;
                    stack.push_back(xs);
                }
                while (!stack.empty()) {
                    const int32_t x = stack.back();
                    stack.pop_back();
                    new_marks.push_back(x);
                    const int32_t i = (int32_t)tr.site[(size_t)x];
                    if (in_patch[(size_t)i] != patch_id) {
                        in_patch[(size_t)i] = patch_id;
                        new_sites.push_back(i);
                    }
                    const int32_t s1 = tr.src1[(size_t)x], s2 = tr.src2[(size_t)x];
                    if (s1 >= x0 && stamp[(size_t)s1] != patch_id) { stamp[(size_t)s1] = patch_id; stack.push_back(s1); }
                    if (s2 >= x0 && stamp[(size_t)s2] != patch_id) { stamp[(size_t)s2] = patch_id; stack.push_back(s2); }
                }
                if ((int64_t)sites.size() + (int64_t)new_sites.size() > entry_cap) {
                    for (int32_t x : new_marks) stamp[(size_t)x] = -1;          // does not fit any more: next patch
                    for (int32_t i : new_sites) in_patch[(size_t)i] = -1;
                    break;
                }
                marked.insert(marked.end(), new_marks.begin(), new_marks.end());
                sites.insert(sites.end(), new_sites.begin(), new_sites.end());
                p++;
            }
            const int32_t own_cnt = p - own_lo;
            if (own_cnt == 0) {                              // the cone of a single site exceeds the cap: the
                out.ok = false;                              // patch kernel cannot hold it (other paths take over)
                return;
            }
            // entries: owned sites in storage order, then the halo sites by storage position
            halo.clear();
            for (int32_t i : sites) {
                const int32_t ps = srank[(size_t)i];
                if (ps < own_lo || ps >= p) halo.push_back(ps);
            }
            std::sort(halo.begin(), halo.end());
            const int64_t entries = (int64_t)own_cnt + (int64_t)halo.size();
            const int64_t e0 = (int64_t)out.entry_pos.size();
            for (int32_t j = 0; j < own_cnt; j++) {
                const int32_t i = dir.store[(size_t)(own_lo + j)];
                slot_of[(size_t)i] = j;
                in_patch[(size_t)i] = patch_id;
                out.entry_pos.push_back(own_lo + j);
            }
            for (size_t j = 0; j < halo.size(); j++) {
                slot_of[(size_t)dir.store[(size_t)halo[j]]] = own_cnt + (int32_t)j;
                out.entry_pos.push_back(halo[j]);
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
                const int32_t i = (int32_t)tr.site[(size_t)x];
                const int32_t si = slot_of[(size_t)i];
                const int32_t u1 = up1[i], u2 = up2[i];
                const int32_t c1 = (dir.layer_of[(size_t)u1] == layer && in_patch[(size_t)u1] == patch_id) ? slot_of[(size_t)u1] : -1;
                const int32_t c2 = (dir.layer_of[(size_t)u2] == layer && in_patch[(size_t)u2] == patch_id) ? slot_of[(size_t)u2] : -1;
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
            deps.clear();
            for (int64_t e = e0; e < (int64_t)out.entry_pos.size(); e++) {
                const int32_t i = dir.store[(size_t)out.entry_pos[(size_t)e]];
                uint32_t l[2];
                for (int q = 0; q < 2; q++) {
                    const int32_t u = q == 0 ? up1[i] : up2[i];
                    l[q] = (dir.layer_of[(size_t)u] == layer && in_patch[(size_t)u] == patch_id) ? (uint32_t)slot_of[(size_t)u]
                                                                                                  : 0xFFFFu;
                    // an upwind in an EARLIER layer is read from memory as the final value its owner patch stored
                    // (one in a later layer reads as 0, :23): the patches this one waits for when the layers are
                    // chained inside one launch (vrt_patch.hip: k_patch_chain)
                    if (dir.layer_of[(size_t)u] < layer && owner[(size_t)u] >= 0) deps.push_back(owner[(size_t)u]);
                }
                out.entry_loc[(size_t)e] = l[0] | (l[1] << 16);
            }
            std::sort(deps.begin(), deps.end());
            deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
            out.dep_list.insert(out.dep_list.end(), deps.begin(), deps.end());
            out.dep_off.push_back((int64_t)out.dep_list.size());
            for (int32_t j = 0; j < own_cnt; j++) owner[(size_t)dir.store[(size_t)(own_lo + j)]] = patch_id;
            out.patch_own_lo.push_back(own_lo);
            out.patch_own_cnt.push_back(own_cnt);
            out.patch_ent_off.push_back(e0);
            out.patch_nlev.push_back(nlev);
            out.n_visits += (int64_t)marked.size();
            out.max_entries = std::max<int64_t>(out.max_entries, entries);
            patch_id++;
        }
    }
    for (int64_t layer = std::max<int64_t>(nl, 2); layer <= nl; layer++) out.layer_patch_off[(size_t)layer] = patch_id;
    out.patch_ent_off.push_back((int64_t)out.entry_pos.size());
    for (size_t x = 0; x < T; x++) out.n_live += live[x];
}

}  // namespace vrt
