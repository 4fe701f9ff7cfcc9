// Host-side grid preparation: what the reference's read_cell does once per grid
// (src/voronoi_utils.jl:36-85), re-designed for a device-resident CSR layout.
//
// Results are identical to the reference's (layers, stable permutation, reduced offsets --
// they define the Gauss-Seidel order, so they must be), but the algorithms are not its
// O(L * n * d) scans: layering is a frontier BFS over the TRANSPOSED neighbour graph, which
// gives exactly "unassigned cells that list a cell of the previous layer" (voronoi_utils.jl:
// 109-119) also for asymmetric neighbour lists, and detects unreachable cells (where the
// reference loops forever) instead of hanging.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "vrt_internal.h"

namespace vrt {

// Parse the voro++ "%i %n" text (rt_preprocessing/output_sites.cc:49): one line per cell,
// "id nb1 ... nbk", lines in any order.  Produces the reference's NeighbourMatrix (column-major
// n x D1, column 0 = count) trimmed to the widest row (voronoi_utils.jl:43-70).
int parse_neighbour_file(const char *path, int64_t n, std::vector<int64_t> &matrix, int64_t &D1)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(VRT_EIO, std::string("cannot open neighbour file ") + path);
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    size_t got = std::fread(buf.data(), 1, (size_t)sz, f);
    std::fclose(f);
    buf[got] = '\n';

    const int64_t cap = kMaxGuess + 1;
    std::vector<int64_t> full((size_t)n * cap, 0);
    int64_t widest = 0;
    const char *p = buf.data(), *end = buf.data() + got + 1;
    while (p < end) {
        // one line
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        int64_t id = 0, count = 0;
        bool have_id = false;
        const char *q = p;
        while (q < eol) {
            while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
            if (q >= eol) break;
            bool neg = false;
            if (*q == '-') { neg = true; q++; }
            else if (*q == '+') q++;
            if (q >= eol || *q < '0' || *q > '9')
                return fail(VRT_EIO, "neighbour file: non-integer token");
            int64_t v = 0;
            while (q < eol && *q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
            if (neg) v = -v;
            if (!have_id) {
                id = v;
                have_id = true;
                if (id < 1 || id > n) return fail(VRT_EGRID, "neighbour file: cell id out of range");
            } else {
                if (count >= kMaxGuess)
                    return fail(VRT_EGRID, "neighbour file: more than 70 neighbours in a cell");
                full[(size_t)(id - 1) + (size_t)n * (size_t)(count + 1)] = v;
                count++;
            }
        }
        if (have_id) {
            full[(size_t)(id - 1)] = count;
            if (count > widest) widest = count;
        }
        p = eol + 1;
    }
    D1 = widest + 1;
    matrix.assign(full.begin(), full.begin() + (size_t)n * (size_t)D1);
    return VRT_OK;
}

static int layer_direction(const vrt_grid *g, int64_t wall, const std::vector<int32_t> &rev_ptr,
                           const std::vector<int32_t> &rev_idx, Direction &d)
{
    const int64_t n = g->n;
    d.layer_of.assign((size_t)n, 0);
    std::vector<int32_t> frontier, next;
    for (int64_t i = 0; i < n; i++) {
        for (int32_t e = g->rowptr[i]; e < g->rowptr[i + 1]; e++)
            if (g->col[e] == wall) { d.layer_of[i] = 1; break; }
        if (d.layer_of[i] == 1) frontier.push_back((int32_t)i);
    }
    int64_t assigned = (int64_t)frontier.size();
    int32_t layer = 1;
    while (assigned < n) {
        if (frontier.empty())
            return fail(VRT_EGRID, "grid has cells that are not connected to the " +
                                       std::string(wall == -5 ? "bottom (-5)" : "top (-6)") +
                                       " wall (the reference's layering would not terminate)");
        next.clear();
        for (int32_t u : frontier)
            for (int32_t e = rev_ptr[u]; e < rev_ptr[u + 1]; e++) {
                int32_t i = rev_idx[e];       // cell i lists u as a neighbour
                if (d.layer_of[i] == 0) {
                    d.layer_of[i] = layer + 1;
                    next.push_back(i);
                }
            }
        assigned += (int64_t)next.size();
        frontier.swap(next);
        layer++;
    }
    const int64_t L = layer;
    // stable sortperm by layer == counting sort (Julia's sortperm is stable: voronoi_utils.jl:72)
    std::vector<int64_t> start((size_t)L + 2, 0);
    for (int64_t i = 0; i < n; i++) start[(size_t)d.layer_of[i] + 1]++;
    for (int64_t l = 1; l <= L + 1; l++) start[(size_t)l] += start[(size_t)l - 1];
    // start[l] = number of sites in layers < l  -> 0-based first position of layer l
    d.perm.assign((size_t)n, 0);
    {
        std::vector<int64_t> cur(start);
        for (int64_t i = 0; i < n; i++) d.perm[(size_t)cur[(size_t)d.layer_of[i]]++] = i + 1;
    }
    // reduce_layers (voronoi_utils.jl:253-269): r[1] = 1, r[l] = first position of layer l,
    // r[end] = n.  Length L + 1.
    d.reduced.assign((size_t)L + 1, 0);
    d.reduced[0] = 1;
    for (int64_t l = 2; l <= L; l++) d.reduced[(size_t)l - 1] = start[(size_t)l] + 1;
    d.reduced[(size_t)L] = n;
    d.n1 = d.reduced[1] - 1;

    // Storage order of the layer paths: layers contiguous, perm[n] kept last; inside a layer
    //   strips (default): the (x, y) plane cut into strips of `W` lattice columns along x; inside a strip rows -- bands of y
    //     half a site spacing high -- from low to high y, the sites of a row by x.  A patch (consecutive positions) is a piece of
    //     a strip, W columns wide and as many rows high as its dependency cone allows: compact, and NEIGHBOURING LANES of
    //     its workgroup hold neighbouring sites of a row.  On a lattice-like grid their upwind neighbours are neighbouring sites
    //     of a row too, so a wave's gather is a handful of contiguous runs for the address unit and the L1 instead of 64 separate
    //     16-byte requests -- the per-lane request, not the byte or the line, is what the patch kernel's gathers cost (DESIGN.md
    //     section 5).  The spacing is taken from the layer's own site count (a = sqrt(2 area / sites): two sites per a x a cell as
    //     on the body-centred lattice of the benchmark grids; on an irregular tessellation it only sets the band height).
    //   morton (VRT_STORE_ORDER=morton): a Morton curve over (x, y) -- rounds 1 to 4.
    // VRT_STORE_ORDER = strips[:W] | morton, read when a grid is created (results do not depend on it).
    {
        auto spread = [](uint32_t v) {            // 16 bits -> every other bit of 32
            v &= 0xFFFFu;
            v = (v | (v << 8)) & 0x00FF00FFu;
            v = (v | (v << 4)) & 0x0F0F0F0Fu;
            v = (v | (v << 2)) & 0x33333333u;
            v = (v | (v << 1)) & 0x55555555u;
            return v;
        };
        int strip_w = 20;                          // columns per strip: C4 7.55 -> 7.01 ms at 20 (16: 7.05, 24: 7.07, 12: 7.18, 30: 7.25)
        if (const char *e = std::getenv("VRT_STORE_ORDER")) {
            const std::string v(e);
            if (v == "morton") strip_w = 0;
            else if (v.rfind("strips", 0) == 0) {
                const size_t c = v.find(':');
                if (c != std::string::npos) strip_w = std::max(1, std::min(4096, std::atoi(v.c_str() + c + 1)));
            }
        }
        if (L >= ((int64_t)1 << 22)) strip_w = 0;     // (the strip key keeps 22 bits for the layer; the Morton key 32)
        const double x0 = g->bounds[2], xs = g->bounds[3] - g->bounds[2];
        const double y0 = g->bounds[4], ys = g->bounds[5] - g->bounds[4];
        std::vector<double> cell((size_t)L + 2, 1.0);           // lattice constant a of every layer (physical units)
        if (strip_w > 0 && xs > 0 && ys > 0)
            for (int64_t l = 1; l <= L; l++) {
                const double m = (double)(start[(size_t)l + 1] - start[(size_t)l]);
                cell[(size_t)l] = std::sqrt(2.0 * xs * ys / std::max(m, 1.0));
            }
        std::vector<uint64_t> key((size_t)n);
        const int64_t last_site = d.perm[(size_t)n - 1] - 1;
        for (int64_t i = 0; i < n; i++) {
            double fx = xs > 0 ? (g->pos[3 * (size_t)i + 1] - x0) / xs : 0.0;
            double fy = ys > 0 ? (g->pos[3 * (size_t)i + 2] - y0) / ys : 0.0;
            fx = fx < 0 ? 0 : (fx > 1 ? 1 : fx);
            fy = fy < 0 ? 0 : (fy > 1 ? 1 : fy);
            const uint64_t layer = (uint64_t)(uint32_t)d.layer_of[(size_t)i];
            if (strip_w > 0 && xs > 0 && ys > 0) {
                // layer 22 bits | strip 14 | row 14 | x 14; (+ 0.25: a jittered lattice row stays inside its band)
                const double a = cell[(size_t)d.layer_of[(size_t)i]];
                const uint64_t row = std::min<uint64_t>(0x3FFFu, (uint64_t)(fy * ys / (0.5 * a) + 0.25));
                const uint64_t strip = std::min<uint64_t>(0x3FFFu, (uint64_t)(fx * xs / a) / (uint64_t)strip_w);
                key[(size_t)i] = (layer << 42) | (strip << 28) | (row << 14) | (uint64_t)(fx * 16383.0);
                if (i == last_site) key[(size_t)i] = (layer << 42) | 0x3FFFFFFFFFFull;
            } else {
                const uint32_t m = spread((uint32_t)(fx * 65535.0)) | (spread((uint32_t)(fy * 65535.0)) << 1);
                key[(size_t)i] = (layer << 32) | m;
                if (i == last_site) key[(size_t)i] = (layer << 32) | 0xFFFFFFFFull;
            }
        }
        d.store.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) d.store[(size_t)i] = (int32_t)i;
        std::stable_sort(d.store.begin(), d.store.end(), [&](int32_t a, int32_t b) {
            return key[(size_t)a] < key[(size_t)b];
        });
        // make sure the never-visited site really is last inside its (last) layer
        for (int64_t q = n - 1; q >= 0; q--)
            if (d.store[(size_t)q] == (int32_t)last_site) {
                for (int64_t t = q; t < n - 1; t++) d.store[(size_t)t] = d.store[(size_t)t + 1];
                d.store[(size_t)n - 1] = (int32_t)last_site;
                break;
            }
    }
    return VRT_OK;
}

int build_grid_host(vrt_grid *g, int64_t n, const double *pos, const int64_t *nbr, int64_t D1,
                    const double bounds[6])
{
    if (n < 2 || D1 < 2) return fail(VRT_EINVAL, "grid needs n >= 2 and D1 >= 2");
    if (n >= (int64_t)1 << 30) return fail(VRT_EINVAL, "n must be below 2^30");
    g->n = n;
    std::memcpy(g->bounds, bounds, sizeof(double) * 6);
    g->pos.assign(pos, pos + 3 * n);
    for (int64_t i = 0; i < 3 * n; i++)
        if (!(pos[i] == pos[i])) return fail(VRT_EGRID, "NaN in positions");

    // CSR pack, row order preserved (the upwind search is order dependent, voronoi_utils.jl:370)
    g->rowptr.assign((size_t)n + 1, 0);
    int64_t nnz = 0, D = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t c = nbr[i];
        if (c < 0 || c > D1 - 1)
            return fail(VRT_EGRID, "neighbour count out of range at site " + std::to_string(i + 1));
        nnz += c;
        if (c > D) D = c;
        if (nnz >= (int64_t)INT32_MAX) return fail(VRT_EINVAL, "too many neighbour entries");
        g->rowptr[(size_t)i + 1] = (int32_t)nnz;
    }
    g->D = D;
    g->col.resize((size_t)nnz);
    std::vector<int32_t> rev_ptr((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) {
        int64_t c = nbr[i];
        for (int64_t j = 0; j < c; j++) {
            int64_t v = nbr[(size_t)i + (size_t)n * (size_t)(j + 1)];
            if (v > n) return fail(VRT_EGRID, "neighbour id out of range at site " + std::to_string(i + 1));
            if (v == i + 1) return fail(VRT_EGRID, "site lists itself as a neighbour: " + std::to_string(i + 1));
            if (v < -(int64_t)1000000) v = -1000000;
            g->col[(size_t)g->rowptr[i] + (size_t)j] = (int32_t)v;
            if (v > 0) rev_ptr[(size_t)v]++;      // count for transposed graph (slot v-1 -> v)
        }
    }
    // transposed graph: rev[u] = cells that list u
    for (int64_t u = 0; u < n; u++) rev_ptr[(size_t)u + 1] += rev_ptr[(size_t)u];
    std::vector<int32_t> rev_idx((size_t)rev_ptr[(size_t)n]);
    {
        std::vector<int32_t> cur(rev_ptr.begin(), rev_ptr.end() - 1);
        for (int64_t i = 0; i < n; i++)
            for (int32_t e = g->rowptr[i]; e < g->rowptr[i + 1]; e++) {
                int32_t v = g->col[e];
                if (v > 0) rev_idx[(size_t)cur[(size_t)v - 1]++] = (int32_t)i;
            }
    }
    int rc = layer_direction(g, -5, rev_ptr, rev_idx, g->up);      // voronoi_utils.jl:97
    if (rc) return rc;
    rc = layer_direction(g, -6, rev_ptr, rev_idx, g->down);        // voronoi_utils.jl:141
    return rc;
}

}  // namespace vrt
