// In-process Voronoi tessellation: replaces the reference's preprocessing
// step -- fork/exec of the voro++ wrapper rt_preprocessing/output_sites.cc (container periodic in
// x and y, walls in z, `con.print_custom("%i %n", ...)`, :35-49), the text round trip
// (src/io.jl:8-40, src/functions.jl:13-23) and its parse (src/voronoi_utils.jl:42-63) -- by a
// library call that returns the neighbour matrix `read_cell` builds, ready for vrt_grid_create.
//
// voro++ is not in the reference checkout, so this is an independent implementation of the same
// geometric object, the way voro++ computes it: every cell on its own, as a convex polyhedron
// (here a list of convex face polygons) that starts as the periodic / wall box around the site
// and is cut by the bisector planes of nearby sites, nearest cells of a search grid first, until
// no unvisited grid cell can hold a site close enough to cut it.  A face that survives names a
// neighbour (1-based id) or a wall (-5 = z_min, -6 = z_max, as voro++ numbers them for this
// container: voronoi_utils.jl:97,141).
//
// NOT reproducible from here: the ORDER of a cell's neighbours in voro++'s "%n" output (its
// internal face order), which the reference's order-dependent upwind rule
// (voronoi_utils.jl:378-386) is sensitive to.  Rows here list the walls first, then the
// neighbours by increasing distance -- deterministic, but a different (equally arbitrary) order.
// tests/test_host.py checks the neighbour SETS against scipy/Qhull's Delaunay triangulation.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "vrt_internal.h"

namespace vrt {
namespace {

struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Face {
    int64_t id;                 // neighbour (1-based), wall (-5 / -6), or 0: periodic-limit face of the start box
    std::vector<V3> v;          // convex polygon, vertices in order (site at the origin)
};

// the cell polyhedron of one site, in coordinates relative to the site
struct Cell {
    std::vector<Face> faces;
    double r2max = 0;           // largest squared vertex distance from the site

    void box(double hx, double hy, double zlo, double zhi)
    {
        const V3 c[8] = {{-hx, -hy, zlo}, {hx, -hy, zlo}, {hx, hy, zlo}, {-hx, hy, zlo},
                         {-hx, -hy, zhi}, {hx, -hy, zhi}, {hx, hy, zhi}, {-hx, hy, zhi}};
        faces.clear();
        faces.push_back({-5, {c[0], c[3], c[2], c[1]}});      // z = zlo (bottom wall)
        faces.push_back({-6, {c[4], c[5], c[6], c[7]}});      // z = zhi (top wall)
        faces.push_back({0, {c[0], c[1], c[5], c[4]}});
        faces.push_back({0, {c[1], c[2], c[6], c[5]}});
        faces.push_back({0, {c[2], c[3], c[7], c[6]}});
        faces.push_back({0, {c[3], c[0], c[4], c[7]}});
        update_r2();
    }
    void update_r2()
    {
        r2max = 0;
        for (const Face &f : faces)
            for (const V3 &p : f.v) r2max = std::max(r2max, dot(p, p));
    }
    // keep the half space  n . x <= d  (n = direction to the neighbour, d = |n|^2 / 2); returns
    // true if the plane cut the cell
    bool cut(V3 nrm, double d, int64_t id, double eps)
    {
        bool any_out = false;
        for (const Face &f : faces) {
            for (const V3 &p : f.v)
                if (dot(nrm, p) - d > eps) { any_out = true; break; }
            if (any_out) break;
        }
        if (!any_out) return false;
        std::vector<Face> out;
        std::vector<V3> cap;                       // intersection points on the cutting plane
        for (const Face &f : faces) {
            Face g;
            g.id = f.id;
            const size_t m = f.v.size();
            for (size_t i = 0; i < m; i++) {       // Sutherland-Hodgman against one plane
                const V3 &a = f.v[i], &b = f.v[(i + 1) % m];
                const double da = dot(nrm, a) - d, db = dot(nrm, b) - d;
                const bool ina = da <= eps, inb = db <= eps;
                if (ina) g.v.push_back(a);
                if (ina != inb) {
                    const double t = da / (da - db);
                    const V3 x = a + (b - a) * t;
                    g.v.push_back(x);
                    cap.push_back(x);
                }
            }
            if (g.v.size() >= 3) out.push_back(std::move(g));
        }
        if (cap.size() >= 3) {
            // the cap is a convex polygon in the cutting plane: order its points by angle around
            // their centroid, drop duplicates
            V3 c = {0, 0, 0};
            for (const V3 &p : cap) c = c + p;
            c = c * (1.0 / (double)cap.size());
            V3 e1 = cap[0] - c;
            for (const V3 &p : cap)
                if (dot(p - c, p - c) > dot(e1, e1)) e1 = p - c;
            const V3 e2 = cross(nrm, e1);
            std::vector<std::pair<double, size_t>> ang(cap.size());
            for (size_t i = 0; i < cap.size(); i++) ang[i] = {std::atan2(dot(cap[i] - c, e2), dot(cap[i] - c, e1)), i};
            std::sort(ang.begin(), ang.end());
            Face g;
            g.id = id;
            for (const auto &a : ang) {
                const V3 &p = cap[a.second];
                if (!g.v.empty()) {
                    const V3 q = p - g.v.back();
                    if (dot(q, q) <= eps * eps) continue;
                }
                g.v.push_back(p);
            }
            while (g.v.size() >= 2) {
                const V3 q = g.v.front() - g.v.back();
                if (dot(q, q) <= eps * eps) g.v.pop_back(); else break;
            }
            if (g.v.size() >= 3) out.push_back(std::move(g));
        }
        faces.swap(out);
        update_r2();
        return true;
    }
    static double area(const Face &f)
    {
        V3 s = {0, 0, 0};
        for (size_t i = 1; i + 1 < f.v.size(); i++) s = s + cross(f.v[i] - f.v[0], f.v[i + 1] - f.v[0]);
        return 0.5 * std::sqrt(dot(s, s));
    }
};

struct SearchGrid {
    int gx, gy, gz;
    double x0, y0, z0, Lx, Ly, Lz, cx, cy, cz;
    std::vector<int32_t> start, item;       // counting-sort buckets
    int cell_of(double v, double lo, double c, int g) const
    {
        int i = (int)std::floor((v - lo) / c);
        return std::min(std::max(i, 0), g - 1);
    }
};

}  // namespace

// nbr_out: (n, D1) column-major like read_cell's matrix (column 0 = count), zero-filled here;
// returns the largest count in *max_count, VRT_EGRID if a row would not fit D1 - 1 entries
int tessellate_host(int64_t n, const double *pos, const double bounds[6], int64_t D1, int64_t *nbr_out,
                    int64_t *max_count, int nthreads)
{
    const double z_min = bounds[0], z_max = bounds[1], x_min = bounds[2], x_max = bounds[3], y_min = bounds[4],
                 y_max = bounds[5];
    const double Lz = z_max - z_min, Lx = x_max - x_min, Ly = y_max - y_min;
    if (!(Lz > 0 && Lx > 0 && Ly > 0)) return fail(VRT_EINVAL, "empty box");
    // search grid with ~4 sites per cell
    SearchGrid G;
    const double vol = Lx * Ly * Lz, h = std::cbrt(vol * 4.0 / (double)std::max<int64_t>(n, 1));
    G.gx = std::max(1, (int)(Lx / h)); G.gy = std::max(1, (int)(Ly / h)); G.gz = std::max(1, (int)(Lz / h));
    G.x0 = x_min; G.y0 = y_min; G.z0 = z_min; G.Lx = Lx; G.Ly = Ly; G.Lz = Lz;
    G.cx = Lx / G.gx; G.cy = Ly / G.gy; G.cz = Lz / G.gz;
    const size_t ncell = (size_t)G.gx * G.gy * G.gz;
    G.start.assign(ncell + 1, 0);
    std::vector<int32_t> cell_of_site((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const double z = pos[3 * i], x = pos[3 * i + 1], y = pos[3 * i + 2];
        if (!(z >= z_min && z <= z_max && x >= x_min && x <= x_max && y >= y_min && y <= y_max))
            return fail(VRT_EINVAL, "site " + std::to_string(i + 1) + " lies outside the box");
        const int c = (G.cell_of(x, x_min, G.cx, G.gx) * G.gy + G.cell_of(y, y_min, G.cy, G.gy)) * G.gz +
                      G.cell_of(z, z_min, G.cz, G.gz);
        cell_of_site[(size_t)i] = c;
        G.start[(size_t)c + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) G.start[c + 1] += G.start[c];
    G.item.resize((size_t)n);
    {
        std::vector<int32_t> cur(G.start.begin(), G.start.end() - 1);
        for (int64_t i = 0; i < n; i++) G.item[(size_t)cur[(size_t)cell_of_site[(size_t)i]]++] = (int32_t)i;
    }
    std::fill(nbr_out, nbr_out + (size_t)n * (size_t)D1, (int64_t)0);
    const double eps = 1e-11 * std::max({Lx, Ly, Lz});
    std::vector<int64_t> worst((size_t)nthreads, 0);
    std::vector<int64_t> bad((size_t)nthreads, -1);
    std::vector<int> oom((size_t)nthreads, 0);       // a worker must not throw: std::terminate would take the host down
    auto work_body = [&](int t) {
        Cell cell;
        std::vector<std::pair<double, std::pair<int32_t, V3>>> cand;
        for (int64_t i = t; i < n; i += nthreads) {
            const double zi = pos[3 * i], xi = pos[3 * i + 1], yi = pos[3 * i + 2];
            // periodic in x, y: a cell cannot reach beyond half a period; walls in z
            cell.box(0.5 * Lx, 0.5 * Ly, z_min - zi, z_max - zi);
            const int ix = G.cell_of(xi, x_min, G.cx, G.gx), iy = G.cell_of(yi, y_min, G.cy, G.gy),
                      iz = G.cell_of(zi, z_min, G.cz, G.gz);
            // shells of grid cells at Chebyshev distance s; stop when the nearest point of the next
            // shell is farther than twice the farthest vertex
            const int smax = std::max({G.gx, G.gy, G.gz});
            for (int s = 0; s <= smax; s++) {
                if (s > 0) {
                    const double reach = (s - 1) * std::min({G.cx, G.cy, G.cz});   // lower bound of the distance to shell s
                    if (reach * reach > 4.0 * cell.r2max) break;
                }
                cand.clear();
                for (int dx = -s; dx <= s; dx++) {
                    for (int dy = -s; dy <= s; dy++) {
                        for (int dz = -s; dz <= s; dz++) {
                            if (std::max({std::abs(dx), std::abs(dy), std::abs(dz)}) != s) continue;
                            const int cz = iz + dz;
                            if (cz < 0 || cz >= G.gz) continue;
                            int cxw = ix + dx, cyw = iy + dy;
                            double sx = 0, sy = 0;                     // shift of the periodic image
                            while (cxw < 0) { cxw += G.gx; sx -= Lx; }
                            while (cxw >= G.gx) { cxw -= G.gx; sx += Lx; }
                            while (cyw < 0) { cyw += G.gy; sy -= Ly; }
                            while (cyw >= G.gy) { cyw -= G.gy; sy += Ly; }
                            const size_t c = ((size_t)cxw * G.gy + cyw) * G.gz + cz;
                            for (int32_t e = G.start[c]; e < G.start[c + 1]; e++) {
                                const int32_t j = G.item[(size_t)e];
                                if (j == i && sx == 0 && sy == 0) continue;
                                const V3 dlt = {pos[3 * j + 1] + sx - xi, pos[3 * j + 2] + sy - yi, pos[3 * j] - zi};
                                cand.push_back({dot(dlt, dlt), {j, dlt}});
                            }
                        }
                    }
                }
                std::sort(cand.begin(), cand.end(), [](const auto &a, const auto &b) {
                    return a.first < b.first || (a.first == b.first && a.second.first < b.second.first);
                });
                for (const auto &cd : cand) {
                    if (cd.first > 4.0 * cell.r2max) break;           // cannot cut: bisector beyond every vertex
                    cell.cut(cd.second.second, 0.5 * cd.first, (int64_t)cd.second.first + 1, eps);
                }
            }
            // the row: walls first, then neighbours by increasing distance (= order of the cuts)
            int64_t cnt = 0;
            const double amin = 1e-14 * (Lx * Ly);
            // A face of the start box that survives (id 0), or a cut by one of the site's own periodic
            // images, is a face towards the site ITSELF across the period (sparse sites in a small
            // box): not a neighbour, left out of the row.
            bool overflow = false;
            for (int pass = 0; pass < 2; pass++)
                for (const Face &f : cell.faces) {
                    if (f.id == 0 || f.id == i + 1) continue;
                    if ((pass == 0) != (f.id < 0)) continue;
                    if (Cell::area(f) <= amin) continue;
                    bool dup = false;                                  // two images of one neighbour: keep one entry
                    for (int64_t q = 1; q <= cnt; q++)
                        if (nbr_out[(size_t)i + (size_t)n * (size_t)q] == f.id) dup = true;
                    if (dup) continue;
                    if (cnt + 1 >= D1) { overflow = true; break; }
                    cnt++;
                    nbr_out[(size_t)i + (size_t)n * (size_t)cnt] = f.id;
                }
            nbr_out[(size_t)i] = cnt;
            worst[(size_t)t] = std::max(worst[(size_t)t], cnt);
            if (overflow && bad[(size_t)t] < 0) bad[(size_t)t] = i;
        }
    };
    auto work = [&](int t) {
        try {
            work_body(t);
        } catch (...) {
            oom[(size_t)t] = 1;
        }
    };
    std::vector<std::thread> pool;
    bool spawn_failed = false;
    for (int t = 0; t < nthreads; t++) {
        try {
            pool.emplace_back(work, t);
        } catch (...) {                              // thread creation failed: this thread does the share
            spawn_failed = true;
            work(t);
        }
    }
    for (auto &th : pool) th.join();
    (void)spawn_failed;
    for (int t = 0; t < nthreads; t++)
        if (oom[(size_t)t]) return fail(VRT_ENOMEM, "out of host memory in the tessellation");
    int64_t mx = 0;
    for (int t = 0; t < nthreads; t++) {
        mx = std::max(mx, worst[(size_t)t]);
        if (bad[(size_t)t] >= 0)
            return fail(VRT_EGRID, "site " + std::to_string(bad[(size_t)t] + 1) + " has more neighbours than the matrix holds");
    }
    if (max_count) *max_count = mx;
    return VRT_OK;
}

}  // namespace vrt
