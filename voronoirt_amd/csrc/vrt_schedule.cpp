// Exact parallel schedule of the reference's serial Gauss-Seidel sweep.
//
// The reference (src/irregular_ray_tracing.jl:37-80, :118-161) visits, per layer, per sweep, the
// sites of the layer one after the other and overwrites I[site] in place, so a site sees
//   - the FINAL value of an upwind neighbour in an earlier layer (or I_0 in layer 1),
//   - THIS sweep's value of an in-layer neighbour that precedes it in the visiting order,
//   - the PREVIOUS sweep's value (0 on the first sweep) of one that follows it,
//   - 0 for a neighbour in a later layer, and 0 for the never-visited last site (:23, and the
//     reduce_layers quirk voronoi_utils.jl:266).
// That serial trace of 3(n - n1 - 1) read-read-write "instructions" is analysed once per angle:
//   pass A  drops a visit whose two inputs are unchanged since the site's previous visit (it
//           would rewrite the identical value),
//   pass B  drops visits whose result is overwritten before anything reads it,
//   pass C  assigns each remaining visit the earliest level that respects read-after-write,
//           write-after-write and write-after-read on the single I array.
// Visits of one level are independent and run as one kernel launch; executing the levels in
// order reproduces the serial result bit for bit (same arithmetic on the same values).
#include <algorithm>

#include "vrt_internal.h"

namespace vrt {

void build_angle_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps,
                          const int32_t *up1, const int32_t *up2, AngleSchedule &out)
{
    out.site.clear();
    out.zflags.clear();
    out.level_off.assign(1, 0);
    out.bad_site = -1;
    const std::vector<int64_t> &r = dir.reduced;
    const int64_t nl = (int64_t)r.size();

    // ---- pass A: trace generation + redundancy elimination --------------------------------
    struct Visit { uint32_t site; uint8_t z; };
    std::vector<Visit> trace;
    trace.reserve((size_t)std::max<int64_t>(0, (n - dir.n1)) * 2);
    std::vector<uint32_t> ver((size_t)n, 0);             // number of value-changing writes so far
    std::vector<uint32_t> seen1((size_t)n, UINT32_MAX);  // input versions at the last kept visit
    std::vector<uint32_t> seen2((size_t)n, UINT32_MAX);
    for (int64_t layer = 2; layer <= nl - 1; layer++) {          // irregular_ray_tracing.jl:37
        const int64_t lo = r[(size_t)layer - 1] - 1;             // 0-based first position
        const int64_t hi = r[(size_t)layer] - 1;                 // 0-based one-past-last
        for (int sweep = 0; sweep < n_sweeps; sweep++) {         // :40
            for (int64_t t = 0; t < hi - lo; t++) {
                const int64_t posn = ascending ? lo + t : hi - 1 - t;   // :41 / :122
                const int64_t i = dir.perm[(size_t)posn] - 1;
                const int32_t u1 = up1[i], u2 = up2[i];
                if (u1 < 0 || u2 < 0) {
                    if (out.bad_site < 0) out.bad_site = i;
                    continue;
                }
                const uint32_t v1 = ver[(size_t)u1], v2 = ver[(size_t)u2];
                if (seen1[(size_t)i] == v1 && seen2[(size_t)i] == v2) continue;  // same inputs
                seen1[(size_t)i] = v1;
                seen2[(size_t)i] = v2;
                ver[(size_t)i]++;
                // a read of version 0 sees the initial array: I_0 in layer 1, otherwise 0
                uint8_t z = 0;
                if (v1 == 0 && dir.layer_of[(size_t)u1] != 1) z |= 1;
                if (v2 == 0 && dir.layer_of[(size_t)u2] != 1) z |= 2;
                trace.push_back({(uint32_t)i, z});
            }
        }
    }
    if (out.bad_site >= 0) return;

    // ---- pass B: liveness, backwards --------------------------------------------------------
    const size_t T = trace.size();
    std::vector<uint8_t> live(T, 0);
    {
        std::vector<uint8_t> needed((size_t)n, 1);     // final values are outputs
        for (size_t x = T; x-- > 0;) {
            const uint32_t i = trace[x].site;
            if (!needed[i]) continue;
            live[x] = 1;
            needed[i] = 0;
            if (!(trace[x].z & 1)) needed[(size_t)up1[i]] = 1;
            if (!(trace[x].z & 2)) needed[(size_t)up2[i]] = 1;
        }
    }

    // ---- pass C: levels ----------------------------------------------------------------------
    std::vector<int32_t> lw((size_t)n, 0);    // level of the last write (0: initial / boundary)
    std::vector<int32_t> lr((size_t)n, 0);    // highest level that read the current value
    std::vector<int32_t> level(T, 0);
    int32_t max_level = 0;
    for (size_t x = 0; x < T; x++) {
        if (!live[x]) continue;
        const uint32_t i = trace[x].site;
        const int32_t u1 = up1[i], u2 = up2[i];
        int32_t lv = std::max(lw[i], lr[i]);                       // WAW, WAR
        if (!(trace[x].z & 1)) lv = std::max(lv, lw[(size_t)u1]);  // RAW
        if (!(trace[x].z & 2)) lv = std::max(lv, lw[(size_t)u2]);
        lv += 1;
        level[x] = lv;
        if (!(trace[x].z & 1)) lr[(size_t)u1] = std::max(lr[(size_t)u1], lv);
        if (!(trace[x].z & 2)) lr[(size_t)u2] = std::max(lr[(size_t)u2], lv);
        lw[i] = lv;
        lr[i] = 0;
        if (lv > max_level) max_level = lv;
    }

    // ---- counting sort by level (stable: keeps trace order inside a level) -------------------
    out.level_off.assign((size_t)max_level + 1, 0);
    for (size_t x = 0; x < T; x++)
        if (live[x]) out.level_off[(size_t)level[x]]++;     // count of level l at index l
    // convert counts (index 1..max) to offsets: level_off[l-1] = start of level l
    {
        int64_t run = 0;
        for (int32_t l = 1; l <= max_level; l++) {
            int64_t c = out.level_off[(size_t)l];
            out.level_off[(size_t)l - 1] = run;
            run += c;
        }
        out.level_off[(size_t)max_level] = run;
        out.site.resize((size_t)run);
        out.zflags.resize((size_t)run);
    }
    {
        std::vector<int64_t> cur(out.level_off.begin(), out.level_off.end());
        for (size_t x = 0; x < T; x++) {
            if (!live[x]) continue;
            const int64_t at = cur[(size_t)level[x] - 1]++;
            out.site[(size_t)at] = trace[x].site;
            out.zflags[(size_t)at] = trace[x].z;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Layer-local schedule for the LDS layer-tile kernel (k_sweep_tiles): one workgroup owns one
// (angle, wavelength) problem and walks the layers itself, so only dependencies INSIDE a layer
// need ordering (earlier layers are final, later layers read as 0).  Same passes A and B as
// above; pass C assigns levels per layer, restarting at 1, over the layer's single-slot tile:
//   RAW  a visit runs after the last write of an in-layer upwind it must see,
//   WAR  a write runs after every earlier-in-trace read of the value it overwrites -- including
//        reads that must still see the tile's initial 0 (the reference's I = zeros),
//   WAW  writes of one site stay ordered.
// Output per site: up to 4 visit levels packed 8 bits each (0 = none); per layer: level count.
// ---------------------------------------------------------------------------------------------
void build_layer_schedule(const Direction &dir, bool ascending, int64_t n, int n_sweeps,
                          const int32_t *up1, const int32_t *up2, LayerSchedule &out)
{
    const std::vector<int64_t> &r = dir.reduced;
    const int64_t nl = (int64_t)r.size();
    out.vis.assign((size_t)n, 0);
    out.nlev.assign((size_t)nl, 0);
    out.ok = true;
    out.n_visits = 0;
    out.max_layer_size = 0;
    out.bad_site = -1;

    struct Visit { uint32_t site; int32_t layer; };
    std::vector<Visit> trace;
    std::vector<uint32_t> ver((size_t)n, 0), seen1((size_t)n, UINT32_MAX), seen2((size_t)n, UINT32_MAX);
    std::vector<uint8_t> z1, z2;   // read sees the never-written initial value
    for (int64_t layer = 2; layer <= nl - 1; layer++) {
        const int64_t lo = r[(size_t)layer - 1] - 1, hi = r[(size_t)layer] - 1;
        out.max_layer_size = std::max(out.max_layer_size, hi - lo);
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
                trace.push_back({(uint32_t)i, (int32_t)layer});
                z1.push_back(v1 == 0);
                z2.push_back(v2 == 0);
            }
    }
    if (out.bad_site >= 0) {
        out.ok = false;
        return;
    }
    const size_t T = trace.size();
    std::vector<uint8_t> live(T, 0);
    {
        std::vector<uint8_t> needed((size_t)n, 1);
        for (size_t x = T; x-- > 0;) {
            const uint32_t i = trace[x].site;
            if (!needed[i]) continue;
            live[x] = 1;
            needed[i] = 0;
            if (!z1[x]) needed[(size_t)up1[i]] = 1;
            if (!z2[x]) needed[(size_t)up2[i]] = 1;
        }
    }
    std::vector<int32_t> lw((size_t)n, 0), lr((size_t)n, 0);
    std::vector<uint8_t> nvis((size_t)n, 0);
    for (size_t x = 0; x < T; x++) {
        if (!live[x]) continue;
        const uint32_t i = trace[x].site;
        const int32_t layer = trace[x].layer;
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
            out.ok = false;      // does not fit the packed encoding: the level kernels handle it
            return;
        }
        out.vis[i] |= (uint32_t)lv << (8 * nvis[i]);
        nvis[i]++;
        out.n_visits++;
        if (lv > out.nlev[(size_t)layer]) out.nlev[(size_t)layer] = lv;
    }
}

// Thread assignment of the layer-step level kernel.  A thread of k_step_levels keeps the
// coefficients of K sites in registers and visits them under a per-site `if (level == t)`; with
// the sites dealt to threads in storage (Morton) order every wave holds sites of every level and
// runs almost all of those serialised branches at every level.  Dealing the sites of a layer in
// the order of their visit patterns (first visit level, then second, ...; stable, so runs keep
// their storage order) makes the waves nearly uniform: a wave then executes about a third of
// the branches (C4: ~36 instead of ~105 per layer).  The Gauss-Seidel order is unaffected --
// it lives in the visit levels; only who holds which site changes.
//   self[i]  : sorted index i (absolute: layer offset + index in the layer) -> storage position
void build_sorted_slots(const Direction &dir, int64_t n, const std::vector<uint32_t> &vis_site,
                        std::vector<int32_t> &self)
{
    self.resize((size_t)n);
    for (int64_t p = 0; p < n; p++) self[(size_t)p] = (int32_t)p;
    const std::vector<int64_t> &r = dir.reduced;
    const int64_t nl = (int64_t)r.size();
    std::vector<uint32_t> key, key2;
    std::vector<int32_t> idx, idx2;
    for (int64_t layer = 1; layer <= nl - 1; layer++) {
        const int64_t lo = r[(size_t)layer - 1] - 1, hi = r[(size_t)layer] - 1;
        const int64_t cnt = hi - lo;
        if (cnt <= 0) continue;
        key.resize((size_t)cnt); key2.resize((size_t)cnt);
        idx.resize((size_t)cnt); idx2.resize((size_t)cnt);
        for (int64_t t = 0; t < cnt; t++) {
            key[(size_t)t] = vis_site[(size_t)dir.store[(size_t)(lo + t)]];
            idx[(size_t)t] = (int32_t)t;
        }
        // LSD radix sort, one pass per visit byte, last visit first => lexicographic by
        // (first, second, third, fourth) visit level
        for (int j = 3; j >= 0; j--) {
            size_t count[257] = {0};
            for (int64_t t = 0; t < cnt; t++) count[((key[(size_t)t] >> (8 * j)) & 0xFFu) + 1]++;
            if (count[1] == (size_t)cnt) continue;          // every site: no such visit
            for (int b = 0; b < 256; b++) count[b + 1] += count[b];
            for (int64_t t = 0; t < cnt; t++) {
                const size_t dst = count[(key[(size_t)t] >> (8 * j)) & 0xFFu]++;
                key2[dst] = key[(size_t)t];
                idx2[dst] = idx[(size_t)t];
            }
            key.swap(key2);
            idx.swap(idx2);
        }
        for (int64_t i = 0; i < cnt; i++) self[(size_t)(lo + i)] = (int32_t)(lo + idx[(size_t)i]);
    }
}

}  // namespace vrt
