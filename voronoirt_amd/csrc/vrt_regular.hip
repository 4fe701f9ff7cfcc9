// Regular-grid short characteristics on gfx950 -- SURVEY.md 8f row 1, the comparison solver of all
// three reference drivers: short_characteristics_up / _down (src/characteristics.jl:19-95,
// :110-180) and the six per-plane kernels xy_/yz_/xz_{up,down}_ray (:191-835).
//
// The method marches plane by plane in z, and inside a plane the yz/xz variants carry a 1-D
// recurrence (the row/column solved just before, 3 periodic sweeps), so one solve has little
// parallelism; the parallel axis is the batch of independent solves (angles x wavelengths, as in
// J_λ_regular, src/lambda_iteration.jl:1-58).  One 1024-thread workgroup owns one solve and walks
// all planes: xy planes are solved point-parallel, yz/xz planes row by row with the carried row
// in LDS and `s_barrier` between rows.  Data is held plane-major with x fastest
// (`[iz][iy][ix]`) so that a plane is contiguous; LDS-free transposes convert from/to the
// caller's Julia layout (nz, nx, ny) = `a[iz + nz*(ix + nx*iy)]`.
//
// Every reference quirk is kept (ghost-zone refresh inside the sweep loop only in yz_up_ray
// :480-482, xz_down_ray's centre values from the upper plane :794,804, the carried row is not
// reset between sweeps); expressions are evaluated in the reference's order without FMA
// contraction, so results agree with the oracle to the last bits (exp() aside).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "vrt_internal.h"

namespace vrt {

// (nz, nx, ny) Julia order <-> [iz][iy][ix]
__global__ void __launch_bounds__(256)
k_reg_to_planes(int nz, int nx, int ny, const double *__restrict__ in, double *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)nz * nx * ny;
    if (t >= total) return;
    const int ix = (int)(t % nx);
    const int iy = (int)((t / nx) % ny);
    const int iz = (int)(t / ((int64_t)nx * ny));
    out[t] = in[iz + (int64_t)nz * (ix + (int64_t)nx * iy)];
}

__global__ void __launch_bounds__(256)
k_reg_from_planes(int nz, int nx, int ny, const double *__restrict__ in, double *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)nz * nx * ny;
    if (t >= total) return;
    const int iz = (int)(t % nz);
    const int ix = (int)((t / nz) % nx);
    const int iy = (int)(t / ((int64_t)nz * nx));
    out[t] = in[ix + (int64_t)nx * (iy + (int64_t)ny * iz)];
}

__device__ __forceinline__ void reg_linear_weights(double dtau, double &a, double &b, double &e)
{
    if (dtau < 5e-4) {                                   // functions.jl:484-500
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

__device__ __forceinline__ double reg_bilinear(double xm, double ym, double x1, double x2, double y1,
                                               double y2, double Q11, double Q12, double Q21, double Q22)
{
    const double dx = x2 - x1, dy = y2 - y1;             // functions.jl:332-355
    const double f1 = ((x2 - xm) * Q11 + (xm - x1) * Q21) / dx;
    const double f2 = ((x2 - xm) * Q12 + (xm - x1) * Q22) / dx;
    return ((y2 - ym) * f1 + (ym - y1) * f2) / dy;
}

// n / d given r = 1.0 / d (correctly rounded): the quotient estimate n r, its exact residual, one correction -- three
// instructions that return the correctly rounded quotient (Markstein's theorem; checked against IEEE division on 3e5
// random and grid-like operand pairs), where the division itself is a dozen: k_reg_xy_march runs ONE compute unit
// per solve at the fp64 rate and is bound by exactly these.  (Not for d = 0 or results near the over/underflow limits.)
__device__ __forceinline__ double reg_div(double n, double d, double r)
{
    const double q0 = n * r;
    const double e = fma(-q0, d, n);
    return fma(e, r, q0);
}

// reg_bilinear with the three divisions done by reg_div: bit-identical results
__device__ __forceinline__ double reg_bilinear_div(double xm, double ym, double x1, double x2, double y1, double y2,
                                                   double rdx, double rdy, double Q11, double Q12, double Q21, double Q22)
{
    const double dx = x2 - x1, dy = y2 - y1;
    const double f1 = reg_div((x2 - xm) * Q11 + (xm - x1) * Q21, dx, rdx);
    const double f2 = reg_div((x2 - xm) * Q12 + (xm - x1) * Q22, dx, rdx);
    return reg_div((y2 - ym) * f1 + (ym - y1) * f2, dy, rdy);
}

// the same interpolation with the two interval lengths inverted once per plane and thread: the
// row march is a dependent chain per row, and an fp64 division costs as much as the rest of a step.
// Differs from reg_bilinear in the last bits only (parity contract of the regular solver: 1e-12).
__device__ __forceinline__ double reg_bilinear_rcp(double wx2, double wx1, double wy2, double wy1, double rdx,
                                                   double rdy, double Q11, double Q12, double Q21, double Q22)
{
    const double f1 = (wx2 * Q11 + wx1 * Q21) * rdx;       // wx2 = x2 - xm, wx1 = xm - x1
    const double f2 = (wx2 * Q12 + wx1 * Q22) * rdx;
    return (wy2 * f1 + wy1 * f2) * rdy;                     // wy2 = y2 - ym, wy1 = ym - y1
}

struct RegArgs {
    int nz, nx, ny, n_sweeps;
    const double *z, *x, *y;
    const double *k;          // (3, n_solve)
    const int *up;            // 1 = up, 0 = down
    const double *S, *alpha;  // plane-major, per solve stride below (0 = shared)
    int64_t S_stride, A_stride;
    int64_t field_period;     // solve s reads field s % field_period (0: field s)
    double *coef;             // per solve 5 * nx * ny doubles: row-march coefficients of one plane
    const double *I0;         // (nx, ny) Julia order per solve: I0[ix + nx*iy]
    double *I;                // plane-major [solve][iz][iy][ix]
};

#define PL(p, ix, iy) (p)[(ix) + nx * (iy)]

// Row march of the yz_/xz_ kernels for rows that fit one point per thread: the rows of a plane are
// solved one after the other (each interpolates in the row solved just before: `I_upper`), so the
// time of a plane is n_ser dependent steps.  Everything of a step that does not depend on the
// carried row -- the interpolated upwind opacity and source function, linear_weights, the
// upwind-plane intensities -- is the same in every sweep and is computed once per plane in a
// row-parallel pass into a per-solve scratch (5 doubles per point); a step of the sweeps then
// streams those (two rows ahead) and its critical path is two LDS reads, a dozen flops, one LDS
// write and one barrier.
//   YZ: serial index = ix, parallel index = iy (yz_up_ray :383-487, yz_down_ray :497-604)
//   XZ: serial index = iy, parallel index = ix (xz_up_ray :614-716, xz_down_ray :726-835)
// The arithmetic is the reference's ((e I_u + a S_u) + b S_c with the two products formed a row
// early) except that the bilinear interpolations multiply by reciprocals (reg_bilinear_rcp).
template <bool YZ>
__device__ __forceinline__ void row_march(int nx, int n_ser, int n_par, int s0, int sgn, int h, bool up, double r,
                                          double z_up, double inc, double zb1, double zb2, const double *__restrict__ par,
                                          const double *__restrict__ A_lo, const double *__restrict__ A_hi,
                                          const double *__restrict__ A_cen, const double *__restrict__ S_lo,
                                          const double *__restrict__ S_hi, const double *__restrict__ S_cen,
                                          const double *__restrict__ Ip, double *__restrict__ Ic, double *&row,
                                          double *&row_nxt, int n_sweeps, bool ghost_in_sweeps, int tid, int T,
                                          double *__restrict__ coef)
{
#define AT(arr, sidx, pidx) (YZ ? (arr)[(sidx) + nx * (pidx)] : (arr)[(pidx) + nx * (sidx)])
    const bool act = tid < n_par - 2;
    const int p = 1 + (act ? tid : 0);
    const int pl = p - h, pu = pl + 1;
    const double c_up = par[p] + inc, c_l = par[pl], c_u = par[pu];
    const double wz2 = zb2 - z_up, wz1 = z_up - zb1, wc2 = c_u - c_up, wc1 = c_up - c_l;
    const double rdz = 1.0 / (zb2 - zb1), rdc = 1.0 / (c_u - c_l);
    struct Raw { double a[5], s[5], q[2]; };
    struct Pre { double e, aS, bS, q1, q2; };
    auto load = [&](int qq, Raw &w) {
        const int sc = s0 + qq * sgn, su = sc + sgn;
        w.a[0] = AT(A_lo, su, pl); w.a[1] = AT(A_lo, su, pu); w.a[2] = AT(A_hi, su, pl); w.a[3] = AT(A_hi, su, pu);
        w.a[4] = AT(A_cen, sc, p);
        w.s[0] = AT(S_lo, su, pl); w.s[1] = AT(S_lo, su, pu); w.s[2] = AT(S_hi, su, pl); w.s[3] = AT(S_hi, su, pu);
        w.s[4] = AT(S_cen, sc, p);
        w.q[0] = AT(Ip, su, pl); w.q[1] = AT(Ip, su, pu);
    };
    auto prepare = [&](const Raw &w, Pre &o) {
        const double a_u = reg_bilinear_rcp(wz2, wz1, wc2, wc1, rdz, rdc, w.a[0], w.a[1], w.a[2], w.a[3]);
        const double dtau = r * (w.a[4] + a_u) / 2.0;
        const double S_u = reg_bilinear_rcp(wz2, wz1, wc2, wc1, rdz, rdc, w.s[0], w.s[1], w.s[2], w.s[3]);
        double a, b, e;
        reg_linear_weights(dtau, a, b, e);
        o.e = e; o.aS = a * S_u; o.bS = b * w.s[4]; o.q1 = w.q[0]; o.q2 = w.q[1];
    };
    // ---- pass 0: the coefficients of every row of the plane.  They do not depend on the carried
    // row, so the rows are independent here (the loads of row q + 1 fly while row q is computed),
    // and they are the same in every sweep: written once to a per-solve scratch, [5][n_ser][n_par - 2].
    const size_t cs = (size_t)n_ser * (size_t)(n_par - 2);
    double *__restrict__ sc_e = coef, *__restrict__ sc_aS = coef + cs, *__restrict__ sc_bS = coef + 2 * cs,
           *__restrict__ sc_q1 = coef + 3 * cs, *__restrict__ sc_q2 = coef + 4 * cs;
    if (act) {
        Raw raw, raw2, raw3;
        load(0, raw);
        if (n_ser > 1) load(1, raw2);
        for (int q = 0; q < n_ser; q++) {
            if (q + 2 < n_ser) load(q + 2, raw3);          // two rows of loads in flight
            Pre o;
            prepare(raw, o);
            const size_t at = (size_t)q * (size_t)(n_par - 2) + (size_t)tid;
            sc_e[at] = o.e; sc_aS[at] = o.aS; sc_bS[at] = o.bS; sc_q1[at] = o.q1; sc_q2[at] = o.q2;
            raw = raw2;
            raw2 = raw3;
        }
    }
    // (each thread reads back only what it wrote itself: no barrier needed)
    auto fetch = [&](int qq, Pre &o) {
        const size_t at = (size_t)qq * (size_t)(n_par - 2) + (size_t)tid;
        o.e = sc_e[at]; o.aS = sc_aS[at]; o.bS = sc_bS[at]; o.q1 = sc_q1[at]; o.q2 = sc_q2[at];
    };
    // ---- the sweeps: per row two LDS reads, a dozen flops, one LDS write, one barrier; the
    // coefficients of row q + 2 are in flight across the barrier
    for (int sweep = 0; sweep < n_sweeps; sweep++) {
        Pre cur, nxt, nxt2;
        if (act) {
            fetch(0, cur);
            if (n_ser > 1) fetch(1, nxt);
        }
        for (int q = 0; q < n_ser; q++) {
            const int sc = s0 + q * sgn;
            if (act) {
                if (q + 2 < n_ser) fetch(q + 2, nxt2);
                const double I_u = up ? reg_bilinear_rcp(wz2, wz1, wc2, wc1, rdz, rdc, cur.q1, cur.q2, row[pl], row[pu])
                                      : reg_bilinear_rcp(wz2, wz1, wc2, wc1, rdz, rdc, row[pl], row[pu], cur.q1, cur.q2);
                const double v = (cur.e * I_u + cur.aS) + cur.bS;
                // the new row goes, with its periodic ghost zones, straight into the carried row in
                // LDS (I_upper = I[idx, :] without a round trip through memory).  The plane in
                // memory is only read back after the last sweep (as the next plane's upwind plane
                // and by the ghost-column refresh), except that yz_up_ray refreshes its ghost
                // columns from columns 1 and nx - 2 inside the sweeps: earlier sweeps store only those.
                row_nxt[p] = v;
                if (p == n_par - 2) row_nxt[0] = v;
                if (p == 1) row_nxt[n_par - 1] = v;
                if (sweep == n_sweeps - 1 || (YZ && ghost_in_sweeps && (sc == 1 || sc == nx - 2))) {
                    AT(Ic, sc, p) = v;
                    if (p == n_par - 2) AT(Ic, sc, 0) = v;
                    if (p == 1) AT(Ic, sc, n_par - 1) = v;
                }
                cur = nxt;
                nxt = nxt2;
            }
            __syncthreads();
            { double *t_ = row; row = row_nxt; row_nxt = t_; }
        }
        if (YZ && ghost_in_sweeps) {                        // yz_up_ray only: inside the sweeps :480-482
            for (int iy = tid; iy < n_par; iy += T) {
                Ic[0 + nx * iy] = Ic[(nx - 2) + nx * iy];
                Ic[(nx - 1) + nx * iy] = Ic[1 + nx * iy];
            }
            __syncthreads();
        }
    }
#undef AT
}

// MAXT = 256: the usual case (rows of up to 256 points: one thread per point, one wave per SIMD, so
// the register allocator has the whole file and nothing spills); MAXT = 1024 for longer rows.
template <int MAXT>
__global__ void __launch_bounds__(MAXT)
k_regular_solve(RegArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) double rows[];  // carried row / column, double-buffered
    const int rlen = ra.nx > ra.ny ? ra.nx : ra.ny;
    double *row = rows, *row_nxt = rows + rlen;
    const int nz = ra.nz, nx = ra.nx, ny = ra.ny;
    const int tid = threadIdx.x, T = blockDim.x;
    const int solve = blockIdx.x;
    const double k0 = ra.k[3 * solve], k1 = ra.k[3 * solve + 1], k2 = ra.k[3 * solve + 2];
    const bool up = ra.up[solve] != 0;
    const int64_t plane = (int64_t)nx * ny;
    const int64_t field = ra.field_period > 0 ? solve % ra.field_period : solve;
    const double *S = ra.S + field * ra.S_stride;
    const double *Al = ra.alpha + field * ra.A_stride;
    double *I = ra.I + (int64_t)solve * plane * nz;
    double *coef = ra.coef + (int64_t)solve * 5 * plane;
    const double *x = ra.x, *y = ra.y, *z = ra.z;

    int sign_x, sign_y;                                           // xy_intersect, functions.jl:430-457
    if (k1 > 0 && k2 > 0) { sign_x = -1; sign_y = -1; }
    else if (k1 < 0 && k2 > 0) { sign_x = 1; sign_y = -1; }
    else if (k1 < 0 && k2 < 0) { sign_x = 1; sign_y = 1; }
    else if (k1 > 0 && k2 < 0) { sign_x = -1; sign_y = 1; }
    else { sign_x = 1; sign_y = 1; }
    const int hx = (sign_x + 1) / 2, hy = (sign_y + 1) / 2;
    const double r_x = fabs((x[1] - x[0]) / k1), r_y = fabs((y[1] - y[0]) / k2);

    // boundary plane: I[1,:,:] = I_0 (:61) / I[end,:,:] = I_0 (:146)
    {
        double *Ib = I + (int64_t)(up ? 0 : nz - 1) * plane;
        const double *I0 = ra.I0 + (int64_t)solve * plane;
        for (int t = tid; t < plane; t += T) Ib[t] = I0[t];      // same (ix + nx*iy) indexing
    }
    __syncthreads();

    for (int s = 1; s < nz; s++) {
        const int idz = up ? s : nz - 1 - s;
        const int idz_u = up ? idz - 1 : idz + 1;
        const double dzp = up ? z[idz] - z[idz - 1] : z[idz + 1] - z[idz];
        const double r_z = fabs(dzp / k0);
        int cut = 1;                                              // argmin([r_z, r_x, r_y]) (:72)
        double m = r_z;
        if (r_x < m) { m = r_x; cut = 2; }
        if (r_y < m) { m = r_y; cut = 3; }
        const double *Ip = I + (int64_t)idz_u * plane;            // upwind plane (final)
        double *Ic = I + (int64_t)idz * plane;                    // plane being solved
        const double *Sc = S + (int64_t)idz * plane, *Su = S + (int64_t)idz_u * plane;
        const double *Ac = Al + (int64_t)idz * plane, *Au = Al + (int64_t)idz_u * plane;

        if (cut == 1) {
            // ---- xy_up_ray :191-278 / xy_down_ray :288-372: every interior point independent ----
            const double r = fabs((z[idz_u] - z[idz]) / k0);
            const double x_inc = r * k1, y_inc = r * k2;
            const int mx = nx - 2, my = ny - 2;
            for (int t = tid; t < mx * my; t += T) {
                const int idx = 1 + t % mx, idy = 1 + t / mx;
                const int xl = idx - hx, xu = xl + 1, yl = idy - hy, yu = yl + 1;
                const double x_up = x[idx] + x_inc, y_up = y[idy] + y_inc;
                const double a_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Au, xl, yl),
                                                PL(Au, xl, yu), PL(Au, xu, yl), PL(Au, xu, yu));
                const double dtau = r * (PL(Ac, idx, idy) + a_u) / 2.0;
                const double S_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Su, xl, yl),
                                                PL(Su, xl, yu), PL(Su, xu, yl), PL(Su, xu, yu));
                double a, b, e;
                reg_linear_weights(dtau, a, b, e);
                const double I_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Ip, xl, yl),
                                                PL(Ip, xl, yu), PL(Ip, xu, yl), PL(Ip, xu, yu));
                PL(Ic, idx, idy) = (e * I_u + a * S_u) + b * PL(Sc, idx, idy);
            }
            __syncthreads();
            for (int idx = 1 + tid; idx <= nx - 2; idx += T) {    // y ghost zones :270-271
                PL(Ic, idx, 0) = PL(Ic, idx, ny - 2);
                PL(Ic, idx, ny - 1) = PL(Ic, idx, 1);
            }
            __syncthreads();
            for (int idy = tid; idy < ny; idy += T) {             // x ghost zones :274-275
                PL(Ic, 0, idy) = PL(Ic, nx - 2, idy);
                PL(Ic, nx - 1, idy) = PL(Ic, 1, idy);
            }
            __syncthreads();
            continue;
        }

        // the yz/xz kernels start from I = zero(I_0) (:396, :511, :627, :744)
        for (int t = tid; t < plane; t += T) Ic[t] = 0.0;
        const int nrow = cut == 2 ? ny : (nx > ny ? nx : ny);
        for (int t = tid; t < nrow; t += T) row[t] = 0.0;         // I_upper / I_lower = zeros
        __syncthreads();
        // z interval of the interpolation: up -> (z[idz-1], z[idz]); down -> (z[idz], z[idz+1])
        const double zb1 = up ? z[idz_u] : z[idz], zb2 = up ? z[idz] : z[idz_u];
        const double *A_lo = up ? Au : Ac, *A_hi = up ? Ac : Au;   // α_lower / α_upper planes
        const double *S_lo = up ? Su : Sc, *S_hi = up ? Sc : Su;

        if (cut == 2) {
            // ---- yz_up_ray :383-487 / yz_down_ray :497-604: serial in x, parallel in y ----------
            const double r = fabs((x[1] - x[0]) / k1);
            const double z_up = z[idz] + r * k0, y_inc = r * k2;
            const int sx0 = sign_x == 1 ? 1 : nx - 2;
            if (ny - 2 <= T) {
                row_march<true>(nx, nx - 2, ny, sx0, sign_x, hy, up, r, z_up, y_inc, zb1, zb2, y, A_lo, A_hi, Ac, S_lo,
                                S_hi, Sc, Ip, Ic, row, row_nxt, ra.n_sweeps, up, tid, T, coef);
            } else
            for (int sweep = 0; sweep < ra.n_sweeps; sweep++) {
                for (int q = 0; q < nx - 2; q++) {
                    const int idx = sx0 + q * sign_x, xu = idx + sign_x;
                    for (int idy = 1 + tid; idy <= ny - 2; idy += T) {
                        const int yl = idy - hy, yu = yl + 1;
                        const double y_up = y[idy] + y_inc;
                        const double a_u = reg_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu], PL(A_lo, xu, yl),
                                                        PL(A_lo, xu, yu), PL(A_hi, xu, yl), PL(A_hi, xu, yu));
                        const double dtau = r * (PL(Ac, idx, idy) + a_u) / 2.0;
                        const double S_u = reg_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu], PL(S_lo, xu, yl),
                                                        PL(S_lo, xu, yu), PL(S_hi, xu, yl), PL(S_hi, xu, yu));
                        double a, b, e;
                        reg_linear_weights(dtau, a, b, e);
                        const double I_u = up ? reg_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu], PL(Ip, xu, yl),
                                                             PL(Ip, xu, yu), row[yl], row[yu])
                                              : reg_bilinear(z_up, y_up, zb1, zb2, y[yl], y[yu], row[yl], row[yu],
                                                             PL(Ip, xu, yl), PL(Ip, xu, yu));
                        const double v = (e * I_u + a * S_u) + b * PL(Sc, idx, idy);
                        // the new row goes to the plane and, with its ghost zones (I[idx, 1] =
                        // I[idx, end-1], I[idx, end] = I[idx, 2]), straight into the carried row
                        // in LDS: I_upper = I[idx, :] without a round trip through memory
                        PL(Ic, idx, idy) = v;
                        row_nxt[idy] = v;
                        if (idy == ny - 2) { PL(Ic, idx, 0) = v; row_nxt[0] = v; }
                        if (idy == 1) { PL(Ic, idx, ny - 1) = v; row_nxt[ny - 1] = v; }
                    }
                    __syncthreads();
                    { double *t_ = row; row = row_nxt; row_nxt = t_; }
                }
                if (up) {                                         // yz_up_ray only: inside the sweeps :480-482
                    for (int idy = tid; idy < ny; idy += T) {
                        PL(Ic, 0, idy) = PL(Ic, nx - 2, idy);
                        PL(Ic, nx - 1, idy) = PL(Ic, 1, idy);
                    }
                    __syncthreads();
                }
            }
            if (!up) {                                            // yz_down_ray: after the sweeps :599-601
                for (int idy = tid; idy < ny; idy += T) {
                    PL(Ic, 0, idy) = PL(Ic, nx - 2, idy);
                    PL(Ic, nx - 1, idy) = PL(Ic, 1, idy);
                }
                __syncthreads();
            }
        } else {
            // ---- xz_up_ray :614-716 / xz_down_ray :726-835: serial in y, parallel in x ----------
            const double r = fabs((y[1] - y[0]) / k2);
            const double z_up = z[idz] + r * k0, x_inc = r * k1;
            const int sy0 = sign_y == 1 ? 1 : ny - 2;
            // centre values from α_upper / S_upper in BOTH variants (:672, :794): for the down
            // ray that is plane idz+1 (reference quirk)
            const double *A_cen = A_hi, *S_cen = S_hi;
            if (nx - 2 <= T) {
                row_march<false>(nx, ny - 2, nx, sy0, sign_y, hx, up, r, z_up, x_inc, zb1, zb2, x, A_lo, A_hi, A_cen,
                                 S_lo, S_hi, S_cen, Ip, Ic, row, row_nxt, ra.n_sweeps, false, tid, T, coef);
            } else
            for (int sweep = 0; sweep < ra.n_sweeps; sweep++) {
                for (int q = 0; q < ny - 2; q++) {
                    const int idy = sy0 + q * sign_y, yu = idy + sign_y;
                    for (int idx = 1 + tid; idx <= nx - 2; idx += T) {
                        const int xl = idx - hx, xu = xl + 1;
                        const double x_up = x[idx] + x_inc;
                        const double a_u = reg_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu], PL(A_lo, xl, yu),
                                                        PL(A_lo, xu, yu), PL(A_hi, xl, yu), PL(A_hi, xu, yu));
                        const double dtau = r * (PL(A_cen, idx, idy) + a_u) / 2.0;
                        const double S_u = reg_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu], PL(S_lo, xl, yu),
                                                        PL(S_lo, xu, yu), PL(S_hi, xl, yu), PL(S_hi, xu, yu));
                        double a, b, e;
                        reg_linear_weights(dtau, a, b, e);
                        const double I_u = up ? reg_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu], PL(Ip, xl, yu),
                                                             PL(Ip, xu, yu), row[xl], row[xu])
                                              : reg_bilinear(z_up, x_up, zb1, zb2, x[xl], x[xu], row[xl], row[xu],
                                                             PL(Ip, xl, yu), PL(Ip, xu, yu));
                        const double v = (e * I_u + a * S_u) + b * PL(S_cen, idx, idy);
                        PL(Ic, idx, idy) = v;                     // + ghost zones :704-705 / :822-823
                        row_nxt[idx] = v;
                        if (idx == nx - 2) { PL(Ic, 0, idy) = v; row_nxt[0] = v; }
                        if (idx == 1) { PL(Ic, nx - 1, idy) = v; row_nxt[nx - 1] = v; }
                    }
                    __syncthreads();
                    { double *t_ = row; row = row_nxt; row_nxt = t_; }
                }
            }
            for (int idx = tid; idx < nx; idx += T) {             // after the sweeps :713-714 / :831-832
                PL(Ic, idx, 0) = PL(Ic, idx, ny - 2);
                PL(Ic, idx, ny - 1) = PL(Ic, idx, 1);
            }
            __syncthreads();
        }
    }
}

// ---- batches whose planes are all of the xy kind (steep rays: every plane point-parallel) --------------------------
// The march through the planes is a dependent chain, but only through I: of a point's update
//     I_c = (e I_u + a S_u) + b S_c                      (xy_up_ray :263 / xy_down_ray :357)
// the interpolated upwind opacity and source function, Δτ, linear_weights and with them e, a S_u and b S_c depend
// on the fields alone.  k_reg_xy_coefs forms those three numbers for every point of every plane of every solve at
// once (grid = tiles x planes x solves: the whole chip), k_reg_xy_march then walks the planes of a solve with one
// workgroup whose step is four LDS reads of the upwind plane (kept in LDS with its ghost zones, double-buffered),
// the bilinear interpolation, two multiply-adds and ONE barrier; the coefficients of the next plane are already in
// flight (the barrier waits for LDS only).  Same expressions in the same order as the plane loop of
// k_regular_solve, so the results are identical to its bit for bit.
__global__ void __launch_bounds__(256)
k_reg_xy_coefs(RegArgs ra, double *__restrict__ xy, int64_t solve0)
{
    const int nz = ra.nz, nx = ra.nx, ny = ra.ny;
    const int64_t solve = solve0 + blockIdx.z;
    const int s = 1 + (int)blockIdx.y;
    const int mx = nx - 2, my = ny - 2;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= mx * my) return;
    const double k0 = ra.k[3 * solve], k1 = ra.k[3 * solve + 1], k2 = ra.k[3 * solve + 2];
    const bool up = ra.up[solve] != 0;
    const int64_t plane = (int64_t)nx * ny;
    const int64_t field = ra.field_period > 0 ? solve % ra.field_period : solve;
    const double *S = ra.S + field * ra.S_stride, *Al = ra.alpha + field * ra.A_stride;
    const double *x = ra.x, *y = ra.y, *z = ra.z;
    int sign_x, sign_y;                                           // xy_intersect, functions.jl:430-457
    if (k1 > 0 && k2 > 0) { sign_x = -1; sign_y = -1; }
    else if (k1 < 0 && k2 > 0) { sign_x = 1; sign_y = -1; }
    else if (k1 < 0 && k2 < 0) { sign_x = 1; sign_y = 1; }
    else if (k1 > 0 && k2 < 0) { sign_x = -1; sign_y = 1; }
    else { sign_x = 1; sign_y = 1; }
    const int hx = (sign_x + 1) / 2, hy = (sign_y + 1) / 2;
    const int idz = up ? s : nz - 1 - s, idz_u = up ? idz - 1 : idz + 1;
    const double *Sc = S + (int64_t)idz * plane, *Su = S + (int64_t)idz_u * plane;
    const double *Ac = Al + (int64_t)idz * plane, *Au = Al + (int64_t)idz_u * plane;
    const double r = fabs((z[idz_u] - z[idz]) / k0);
    const double x_inc = r * k1, y_inc = r * k2;
    const int idx = 1 + t % mx, idy = 1 + t / mx;
    const int xl = idx - hx, xu = xl + 1, yl = idy - hy, yu = yl + 1;
    const double x_up = x[idx] + x_inc, y_up = y[idy] + y_inc;
    const double a_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Au, xl, yl), PL(Au, xl, yu), PL(Au, xu, yl),
                                    PL(Au, xu, yu));
    const double dtau = r * (PL(Ac, idx, idy) + a_u) / 2.0;
    const double S_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Su, xl, yl), PL(Su, xl, yu), PL(Su, xu, yl),
                                    PL(Su, xu, yu));
    double a, b, e;
    reg_linear_weights(dtau, a, b, e);
    double *c = xy + ((int64_t)blockIdx.z * nz + idz) * 3 * plane + (idx + nx * idy);
    c[0] = e;
    c[plane] = a * S_u;
    c[2 * plane] = b * PL(Sc, idx, idy);
}

typedef unsigned int reg_u32x2 __attribute__((ext_vector_type(2)));

// The march with the upwind plane read back from memory: the fallback for planes of which two do not fit LDS.
__global__ void __launch_bounds__(1024)
k_reg_xy_march_mem(RegArgs ra, const double *__restrict__ xy, int64_t solve0)
{
    const int nz = ra.nz, nx = ra.nx, ny = ra.ny;
    const int tid = threadIdx.x, T = blockDim.x;
    const int64_t solve = solve0 + blockIdx.x;
    const int64_t plane = (int64_t)nx * ny;
    const double k0 = ra.k[3 * solve], k1 = ra.k[3 * solve + 1], k2 = ra.k[3 * solve + 2];
    const bool up = ra.up[solve] != 0;
    double *I = ra.I + solve * plane * nz;
    const double *cf = xy + (int64_t)blockIdx.x * nz * 3 * plane;
    const double *x = ra.x, *y = ra.y, *z = ra.z;
    int sign_x, sign_y;
    if (k1 > 0 && k2 > 0) { sign_x = -1; sign_y = -1; }
    else if (k1 < 0 && k2 > 0) { sign_x = 1; sign_y = -1; }
    else if (k1 < 0 && k2 < 0) { sign_x = 1; sign_y = 1; }
    else if (k1 > 0 && k2 < 0) { sign_x = -1; sign_y = 1; }
    else { sign_x = 1; sign_y = 1; }
    const int hx = (sign_x + 1) / 2, hy = (sign_y + 1) / 2;
    const int mx = nx - 2, my = ny - 2;
    {                                                             // boundary plane: I[1,:,:] = I_0 (:61) / I[end,:,:] = I_0 (:146)
        double *Ib = I + (int64_t)(up ? 0 : nz - 1) * plane;
        const double *I0 = ra.I0 + solve * plane;
        for (int64_t t = tid; t < plane; t += T) Ib[t] = I0[t];
    }
    __syncthreads();
    for (int s = 1; s < nz; s++) {
        const int idz = up ? s : nz - 1 - s, idz_u = up ? idz - 1 : idz + 1;
        const double r = fabs((z[idz_u] - z[idz]) / k0);
        const double x_inc = r * k1, y_inc = r * k2;
        double *Ic = I + (int64_t)idz * plane;
        const double *Ip = I + (int64_t)idz_u * plane;
        const double *cz = cf + (int64_t)idz * 3 * plane;
        for (int t = tid; t < mx * my; t += T) {
            const int iy = t / mx, idx = 1 + t - mx * iy, idy = 1 + iy;
            const int xl = idx - hx, xu = xl + 1, yl = idy - hy, yu = yl + 1;
            const double x_up = x[idx] + x_inc, y_up = y[idy] + y_inc;
            const double I_u = reg_bilinear(x_up, y_up, x[xl], x[xu], y[yl], y[yu], PL(Ip, xl, yl), PL(Ip, xl, yu),
                                            PL(Ip, xu, yl), PL(Ip, xu, yu));
            const int64_t o = idx + (int64_t)nx * idy;
            PL(Ic, idx, idy) = (cz[o] * I_u + cz[plane + o]) + cz[2 * plane + o];
        }
        __syncthreads();
        for (int idx = 1 + tid; idx <= nx - 2; idx += T) {        // y ghost zones :270-271
            PL(Ic, idx, 0) = PL(Ic, idx, ny - 2);
            PL(Ic, idx, ny - 1) = PL(Ic, idx, 1);
        }
        __syncthreads();
        for (int idy = tid; idy < ny; idy += T) {                 // x ghost zones :274-275
            PL(Ic, 0, idy) = PL(Ic, nx - 2, idy);
            PL(Ic, nx - 1, idy) = PL(Ic, 1, idy);
        }
        __syncthreads();
    }
}

// The march with the upwind plane in LDS.  One workgroup = one compute unit runs a solve, so the march is bound by
// the instructions it issues per point; everything that does not depend on the point is kept out of its way:
//  * of the bilinear interpolation (functions.jl:332-355) the differences (x2 - xm), (xm - x1) depend on (ix, plane)
//    only and dx on ix only (the same in y): 1-D tables in LDS, those of the next plane written during this one;
//    the three divisions are reg_div's three instructions;
//  * the planes in LDS hold no ghost zones: a read that falls on one goes to the interior point it mirrors
//    (:270-275; only one side per axis can, the upwind one) -- except in the first step, whose upwind plane is the
//    caller's I_0 with whatever its ghost zones hold (:61, :146);
//  * a finished plane goes to memory one step later, whole and coalesced, ghost zones filled on the way;
//  * the coefficients of the next plane are loaded (buffer loads: one offset register per point) BEFORE this plane's
//    stores are issued and the barrier waits for LDS only, so no wait in the loop ever waits for a store.
// NPT > 0: that many points per thread with their coefficients in registers; NPT = 0: any number, loaded where used.
// LDS (doubles): 2 planes | x, y | (dx, 1/dx) by ix_lower, (dy, 1/dy) | 2 x (x2-xm, xm-x1) by ix, 2 x the same in y | z
template <int NPT>
__global__ void __launch_bounds__(1024)
k_reg_xy_march_lds(RegArgs ra, const double *__restrict__ xy, int64_t solve0)
{
    extern __shared__ __attribute__((aligned(16))) double pl[];
    constexpr int NP = NPT > 0 ? NPT : 1;
    const int nz = ra.nz, nx = ra.nx, ny = ra.ny;
    const int tid = threadIdx.x, T = blockDim.x;
    const int64_t solve = solve0 + blockIdx.x;
    const int plane = nx * ny;
    int po = 0, co = plane;                                            // offsets of the upwind / the new plane in pl
    double *ax = pl + 2 * plane, *ay = ax + nx;
    const int lead = 2 * plane + nx + ny;
    double2 *dxr = (double2 *)(pl + lead + (lead & 1)), *dyr = dxr + nx;   // (padded to 16 bytes; the host's lds_m too)
    double2 *wxa = dyr + ny, *wya = wxa + 2 * nx;
    double *az = (double *)(wya + 2 * ny);
    const double k0 = ra.k[3 * solve], k1 = ra.k[3 * solve + 1], k2 = ra.k[3 * solve + 2];
    const bool up = ra.up[solve] != 0;
    double *I = ra.I + solve * plane * nz;
    const double *cf = xy + (int64_t)blockIdx.x * nz * 3 * plane;
    int sign_x, sign_y;
    if (k1 > 0 && k2 > 0) { sign_x = -1; sign_y = -1; }
    else if (k1 < 0 && k2 > 0) { sign_x = 1; sign_y = -1; }
    else if (k1 < 0 && k2 < 0) { sign_x = 1; sign_y = 1; }
    else if (k1 > 0 && k2 < 0) { sign_x = -1; sign_y = 1; }
    else { sign_x = 1; sign_y = 1; }
    const int hx = (sign_x + 1) / 2, hy = (sign_y + 1) / 2;
    const int mx = nx - 2, my = ny - 2;
    for (int t = tid; t < nx; t += T) {
        ax[t] = ra.x[t];
        if (t + 1 < nx) {
            const double d = ra.x[t + 1] - ra.x[t];
            dxr[t] = make_double2(d, 1.0 / d);
        }
    }
    for (int t = tid; t < ny; t += T) {
        ay[t] = ra.y[t];
        if (t + 1 < ny) {
            const double d = ra.y[t + 1] - ra.y[t];
            dyr[t] = make_double2(d, 1.0 / d);
        }
    }
    for (int t = tid; t < nz; t += T) az[t] = ra.z[t];
    {                                                             // boundary plane: I[1,:,:] = I_0 (:61) / I[end,:,:] = I_0 (:146)
        double *Ib = I + (int64_t)(up ? 0 : nz - 1) * plane;
        const double *I0 = ra.I0 + solve * plane;
        for (int t = tid; t < plane; t += T) {
            const double v = I0[t];
            Ib[t] = v;
            pl[po + t] = v;
        }
    }
    __syncthreads();
    // the interpolation weights of plane step s into table half (s & 1)
    auto weights = [&](int s) {
        const int idz = up ? s : nz - 1 - s, idz_u = up ? idz - 1 : idz + 1;
        const double r = fabs((az[idz_u] - az[idz]) / k0);
        const double x_inc = r * k1, y_inc = r * k2;
        for (int t = tid; t < mx + my; t += T) {
            if (t < mx) {
                const int idx = 1 + t, xl = idx - hx;
                const double x_up = ax[idx] + x_inc;
                wxa[(s & 1) * nx + idx] = make_double2(ax[xl + 1] - x_up, x_up - ax[xl]);
            } else {
                const int idy = 1 + t - mx, yl = idy - hy;
                const double y_up = ay[idy] + y_inc;
                wya[(s & 1) * ny + idy] = make_double2(ay[yl + 1] - y_up, y_up - ay[yl]);
            }
        }
    };
    int pt[NP];                                                   // ix | iy << 16; -1: none
    int cs[NP + 1];                                               // copy-out: LDS source of plane element tid + j T; -1: none
    if (NPT > 0) {
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int t = tid + j * T;
            pt[j] = t < mx * my ? (1 + t % mx) | ((1 + t / mx) << 16) : -1;          // (nx, ny <= 16384)
        }
#pragma unroll
        for (int j = 0; j < NP + 1; j++) {
            const int t = tid + j * T;
            const int iy = t / nx, ix = t - nx * iy;
            const int sx = ix == 0 ? nx - 2 : ix == nx - 1 ? 1 : ix, sy = iy == 0 ? ny - 2 : iy == ny - 1 ? 1 : iy;
            cs[j] = t < plane ? sx + nx * sy : -1;
        }
    }
    double ce[NP], caS[NP], cbS[NP], ne[NP], naS[NP], nbS[NP];
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(cf), 0, __builtin_amdgcn_readfirstlane((int)((unsigned)nz * 3u * (unsigned)plane * 8u)), 0x00020000);
    auto fetch = [&](int s, double (&e_)[NP], double (&aS_)[NP], double (&bS_)[NP]) {
        const int idz = up ? s : nz - 1 - s;
        const int so = __builtin_amdgcn_readfirstlane(idz * 3 * plane * 8), sp = __builtin_amdgcn_readfirstlane(plane * 8);
#pragma unroll
        for (int j = 0; j < NP; j++)
            if (pt[j] >= 0) {
                const int o = ((pt[j] & 0xFFFF) + nx * (pt[j] >> 16)) * 8;
                reg_u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(crs, o, so, 0);
                reg_u32x2 v1 = __builtin_amdgcn_raw_buffer_load_b64(crs, o, so + sp, 0);
                reg_u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(crs, o, so + 2 * sp, 0);
                __builtin_memcpy(&e_[j], &v0, 8);
                __builtin_memcpy(&aS_[j], &v1, 8);
                __builtin_memcpy(&bS_[j], &v2, 8);
            }
    };
    if (NPT > 0) fetch(1, ce, caS, cbS);
    weights(1);
    __syncthreads();
    for (int s = 1; s < nz; s++) {
        const int idz = up ? s : nz - 1 - s, idz_u = up ? idz - 1 : idz + 1;
        if (NPT > 0 && s + 1 < nz) fetch(s + 1, ne, naS, nbS);
        if (s > 1) {                                              // the plane of the previous step to memory, ghost zones filled
            double *Iu = I + (int64_t)idz_u * plane;
            if (NPT > 0) {
#pragma unroll
                for (int j = 0; j < NP + 1; j++)
                    if (cs[j] >= 0) Iu[tid + j * T] = pl[po + cs[j]];
            } else {
                for (int t = tid; t < plane; t += T) {
                    const int iy = t / nx, ix = t - nx * iy;
                    const int sx = ix == 0 ? nx - 2 : ix == nx - 1 ? 1 : ix, sy = iy == 0 ? ny - 2 : iy == ny - 1 ? 1 : iy;
                    Iu[t] = pl[po + sx + nx * sy];
                }
            }
        }
        if (s + 1 < nz) weights(s + 1);
        const bool wrap = s > 1;                                  // (step 1 reads the caller's I_0, ghost zones as given)
        const double2 *wxs = wxa + (s & 1) * nx, *wys = wya + (s & 1) * ny;
        const double *cz = cf + (int64_t)idz * 3 * plane;
        auto point = [&](int idx, int idy, double e_, double aS_, double bS_) {
            const int xl = idx - hx, yl = idy - hy;
            int xa = xl, xb = xl + 1, ya = yl, yb = yl + 1;       // where the four upwind values are read
            if (wrap) {
                if (hx) xa = xa == 0 ? nx - 2 : xa; else xb = xb == nx - 1 ? 1 : xb;
                if (hy) ya = ya == 0 ? ny - 2 : ya; else yb = yb == ny - 1 ? 1 : yb;
            }
            const double2 wx = wxs[idx], wy = wys[idy], dx = dxr[xl], dy = dyr[yl];
            const double *P = pl + po;
            const double Q11 = P[xa + nx * ya], Q12 = P[xa + nx * yb], Q21 = P[xb + nx * ya], Q22 = P[xb + nx * yb];
            const double f1 = reg_div(wx.x * Q11 + wx.y * Q21, dx.x, dx.y);          // reg_bilinear's expressions
            const double f2 = reg_div(wx.x * Q12 + wx.y * Q22, dx.x, dx.y);
            const double I_u = reg_div(wy.x * f1 + wy.y * f2, dy.x, dy.y);
            pl[co + idx + nx * idy] = (e_ * I_u + aS_) + bS_;
        };
        if (NPT > 0) {
#pragma unroll
            for (int j = 0; j < NP; j++) {
                int pj = pt[j];
                asm volatile("" : "+v"(pj));                      // (indices and LDS offsets re-derived per plane, not kept per point)
                if (pj >= 0) point(pj & 0xFFFF, pj >> 16, ce[j], caS[j], cbS[j]);
            }
#pragma unroll
            for (int j = 0; j < NP; j++) {
                ce[j] = ne[j]; caS[j] = naS[j]; cbS[j] = nbS[j];
            }
        } else {
#pragma unroll 1
            for (int t = tid; t < mx * my; t += T) {
                const int iy = t / mx, idx = 1 + t - mx * iy, p = idx + nx * (1 + iy);
                point(idx, 1 + iy, cz[p], cz[plane + p], cz[2 * plane + p]);
            }
        }
        // the planes talk through LDS only: wait for the LDS writes, not for the loads and stores in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int t_ = po; po = co; co = t_;
    }
    if (nz > 1) {                                                 // the last plane
        double *Il = I + (int64_t)(up ? nz - 1 : 0) * plane;
        for (int t = tid; t < plane; t += T) {
            const int iy = t / nx, ix = t - nx * iy;
            const int sx = ix == 0 ? nx - 2 : ix == nx - 1 ? 1 : ix, sy = iy == 0 ? ny - 2 : iy == ny - 1 ? 1 : iy;
            Il[t] = pl[po + sx + nx * sy];
        }
    }
}

#undef PL

}  // namespace vrt

using namespace vrt;

// ---- device-resident form: a handle owns the grid axes and the (grow-only) workspaces -----------
struct vrt_regular {
    int device = 0;
    int64_t nz = 0, nx = 0, ny = 0;
    double *d_g = nullptr;                 // z | x | y
    std::vector<double> h_g;               // the same on the host (launch geometry)
    double *d_S = nullptr, *d_A = nullptr, *d_I = nullptr, *d_k = nullptr, *d_coef = nullptr, *d_xy = nullptr;
    int *d_up = nullptr;
    int64_t cap_S = 0, cap_A = 0, cap_I = 0, cap_k = 0, cap_coef = 0, cap_xy = 0;      // in solves
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool timed = false;
    int force_threads = 0;                 // VRT_REG_THREADS, read once at creation (tests: forces the launch shape)
    int xy_split = 1;                      // VRT_REG_XY (creation): 0 = all-xy batches through k_regular_solve too;
                                           //   2 = split, upwind plane read from memory instead of LDS (tests)
};

static void regular_free(vrt_regular *r)
{
    if (!r) return;
    for (void *p : {(void *)r->d_g, (void *)r->d_S, (void *)r->d_A, (void *)r->d_I, (void *)r->d_k, (void *)r->d_up,
                    (void *)r->d_coef, (void *)r->d_xy})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : r->ev)
        if (e) (void)hipEventDestroy(e);
    delete r;
}

extern "C" int vrt_regular_create(int64_t nz, int64_t nx, int64_t ny, const double *z, const double *x,
                                  const double *y, int device, vrt_regular **out)
{
    DeviceScope scope;
    if (!z || !x || !y || !out) return fail(VRT_EINVAL, "NULL argument");
    if (nz < 2 || nx < 3 || ny < 3) return fail(VRT_EINVAL, "bad sizes");
    if (nx > 16384 || ny > 16384) return fail(VRT_EINVAL, "nx, ny must be at most 16384");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(VRT_ENODEVICE, "no HIP device available (libvrt_hip has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(VRT_EINVAL, "device ordinal out of range");
    VRT_HIP_TRY(hipSetDevice(device));
    vrt_regular *r = new vrt_regular;
    if (const char *e = std::getenv("VRT_REG_THREADS")) r->force_threads = std::max(64, std::min(1024, std::atoi(e) / 64 * 64));
    if (const char *e = std::getenv("VRT_REG_XY")) r->xy_split = std::max(0, std::min(2, std::atoi(e)));
    r->device = device;
    r->nz = nz; r->nx = nx; r->ny = ny;
    r->h_g.assign(z, z + nz);
    r->h_g.insert(r->h_g.end(), x, x + nx);
    r->h_g.insert(r->h_g.end(), y, y + ny);
    hipError_t e = hipMalloc((void **)&r->d_g, sizeof(double) * (size_t)(nz + nx + ny));
    if (e == hipSuccess) e = hipMemcpy(r->d_g, z, sizeof(double) * nz, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(r->d_g + nz, x, sizeof(double) * nx, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(r->d_g + nz + nx, y, sizeof(double) * ny, hipMemcpyHostToDevice);
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreate(&r->ev[i]);
    if (e != hipSuccess) {
        regular_free(r);
        return fail(VRT_ENODEVICE, std::string("vrt_regular_create: ") + hipGetErrorString(e));
    }
    *out = r;
    return VRT_OK;
}

extern "C" void vrt_regular_destroy(vrt_regular *r) { regular_free(r); }

static int regular_grow(double *&buf, int64_t &cap, int64_t need, size_t per)
{
    if (buf && need <= cap) return VRT_OK;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&buf, sizeof(double) * per * (size_t)need);
    if (e != hipSuccess) {
        buf = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    cap = need;
    return VRT_OK;
}

// dS, dalpha, dI0, dI_out: device pointers in the caller's (Julia) layouts; k, up: host
extern "C" int vrt_regular_execute_dev(vrt_regular *r, int64_t n_solve, const double *k, const int *up,
                                       const double *dS, int64_t S_stride, const double *dalpha,
                                       int64_t alpha_stride, int64_t field_period, const double *dI0,
                                       int n_sweeps, double *dI_out, void *stream)
{
    DeviceScope scope;
    if (field_period < 0 || field_period > n_solve) return fail(VRT_EINVAL, "field_period must be in [0, n_solve]");
    if (!r || !k || !up || !dS || !dalpha || !dI0 || !dI_out) return fail(VRT_EINVAL, "NULL argument");
    if (n_solve < 1 || n_sweeps < 1) return fail(VRT_EINVAL, "bad sizes");
    const int64_t nz = r->nz, nx = r->nx, ny = r->ny, vol = nz * nx * ny;
    if ((S_stride != 0 && S_stride != vol) || (alpha_stride != 0 && alpha_stride != vol))
        return fail(VRT_EINVAL, "S_stride / alpha_stride must be 0 (shared) or nz*nx*ny");
    for (int64_t s = 0; s < n_solve; s++) {
        const double *ks = k + 3 * s;
        const double nrm = std::sqrt(ks[0] * ks[0] + ks[1] * ks[1] + ks[2] * ks[2]);
        if (!(std::fabs(nrm - 1.0) < 1e-6))            // functions.jl:432 @assert norm(k) ≈ 1
            return fail(VRT_EINVAL, "direction " + std::to_string(s + 1) + " is not a unit vector");
        if (ks[0] == 0.0) return fail(VRT_EINVAL, "horizontal ray (k_z = 0) has no upwind plane");
    }
    VRT_HIP_TRY(hipSetDevice(r->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nfield = field_period > 0 ? field_period : n_solve;
    const int64_t nS = S_stride ? nfield : 1, nA = alpha_stride ? nfield : 1;
    int rc;
    if ((rc = regular_grow(r->d_S, r->cap_S, nS, (size_t)vol))) return rc;
    if ((rc = regular_grow(r->d_A, r->cap_A, nA, (size_t)vol))) return rc;
    if ((rc = regular_grow(r->d_I, r->cap_I, n_solve, (size_t)vol))) return rc;
    if ((rc = regular_grow(r->d_coef, r->cap_coef, n_solve, (size_t)(5 * nx * ny)))) return rc;
    if (n_solve > r->cap_k) {
        if (r->d_k) (void)hipFree(r->d_k);
        if (r->d_up) (void)hipFree(r->d_up);
        r->d_k = nullptr; r->d_up = nullptr; r->cap_k = 0;
        VRT_HIP_TRY(hipMalloc((void **)&r->d_k, sizeof(double) * 3 * (size_t)n_solve));
        VRT_HIP_TRY(hipMalloc((void **)&r->d_up, sizeof(int) * (size_t)n_solve));
        r->cap_k = n_solve;
    }
    VRT_HIP_TRY(hipMemcpyAsync(r->d_k, k, sizeof(double) * 3 * (size_t)n_solve, hipMemcpyHostToDevice, st));
    VRT_HIP_TRY(hipMemcpyAsync(r->d_up, up, sizeof(int) * (size_t)n_solve, hipMemcpyHostToDevice, st));
    VRT_HIP_TRY(hipEventRecord(r->ev[0], st));
    const unsigned tb = (unsigned)((vol + 255) / 256);
    for (int64_t s = 0; s < nS; s++)
        hipLaunchKernelGGL(k_reg_to_planes, dim3(tb), dim3(256), 0, st, (int)nz, (int)nx, (int)ny, dS + s * vol, r->d_S + s * vol);
    for (int64_t s = 0; s < nA; s++)
        hipLaunchKernelGGL(k_reg_to_planes, dim3(tb), dim3(256), 0, st, (int)nz, (int)nx, (int)ny, dalpha + s * vol, r->d_A + s * vol);
    RegArgs ra;
    ra.nz = (int)nz; ra.nx = (int)nx; ra.ny = (int)ny; ra.n_sweeps = n_sweeps;
    ra.z = r->d_g; ra.x = r->d_g + nz; ra.y = r->d_g + nz + nx;
    ra.k = r->d_k; ra.up = r->d_up;
    ra.S = r->d_S; ra.alpha = r->d_A; ra.S_stride = S_stride; ra.A_stride = alpha_stride;
    ra.field_period = field_period;
    ra.I0 = dI0; ra.I = r->d_I;
    ra.coef = r->d_coef;
    const size_t lds = 2 * sizeof(double) * (size_t)std::max(nx, ny);
    VRT_HIP_TRY(hipEventRecord(r->ev[1], st));
    // one thread per point of a row (the yz/xz planes march row by row), at least two waves; many
    // such workgroups share a CU, which is where the batch's throughput comes from
    int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(128, (std::max(nx, ny) - 2 + 63) / 64 * 64));
    // A batch too small to fill the chip whose rays are all steep enough to cut the horizontal plane
    // first (every plane point-parallel: the searchlight / emergent-intensity case, θ = 180°) spends
    // its threads on the plane loops instead: up to 1024 per solve (60³: 1.46 -> 0.73 ms per solve)
    bool all_xy = true;
    {
        const double *hz = r->h_g.data(), *hx = hz + nz, *hy = hx + nx;
        for (int64_t s = 0; s < n_solve && all_xy; s++) {
            const double *ks = k + 3 * s;
            const double r_x = std::fabs((hx[1] - hx[0]) / ks[1]), r_y = std::fabs((hy[1] - hy[0]) / ks[2]);
            for (int64_t iz = 1; iz < nz && all_xy; iz++) {
                const double r_z = std::fabs((hz[iz] - hz[iz - 1]) / ks[0]);
                if (r_x < r_z || r_y < r_z) all_xy = false;          // the kernel's argmin (:72)
            }
        }
        if (all_xy)
            while (threads * 2 <= 1024 && n_solve * threads * 2 <= 256 * 1024 && (int64_t)threads * 2 <= (nx - 2) * (ny - 2))
                threads *= 2;
    }
    if (r->force_threads) threads = r->force_threads;
    // Such a batch takes the split form (k_reg_xy_coefs over the whole chip + k_reg_xy_march, above)
    const int64_t interior = (nx - 2) * (ny - 2);
    int mt = (int)std::min<int64_t>(1024, (interior + 63) / 64 * 64);
    if (r->force_threads) mt = r->force_threads;
    const int npt = (int)((interior + mt - 1) / mt);
    if (all_xy && r->xy_split && 24 * vol < ((int64_t)1 << 31) && nz <= 65536) {   // (a solve's coefficients: 32-bit byte offsets; planes = grid.y)
        // coefficients: 3 doubles per point, plane and solve, in chunks of solves of at most 2 GiB
        const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({n_solve, 65535, ((int64_t)1 << 31) / (24 * vol)}));
        if ((rc = regular_grow(r->d_xy, r->cap_xy, chunk, (size_t)(3 * vol)))) return rc;
        // LDS of the march: two planes, the axes and the weight tables (layout: k_reg_xy_march_lds); the planes
        // padded to an even number of doubles so that the 16-byte tables behind them stay aligned
        const size_t planes_d = 2 * (size_t)(nx * ny), lead_d = planes_d + (size_t)(nx + ny);
        const size_t lds_m = sizeof(double) * (lead_d + (lead_d & 1) + (size_t)(6 * (nx + ny) + nz));
        const bool in_lds = r->xy_split == 1 && lds_m <= 150 * 1024;
#define VRT_XY_MARCH(N)                                                                                                 \
    do {                                                                                                                \
        if (lds_m > 48 * 1024)                                                                                          \
            VRT_HIP_TRY(hipFuncSetAttribute((const void *)k_reg_xy_march_lds<N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)lds_m));                                                               \
        hipLaunchKernelGGL((k_reg_xy_march_lds<N>), dim3((unsigned)cnt), dim3((unsigned)mt), lds_m, st, ra, r->d_xy, s0); \
    } while (0)
        // (registers hold NPT points and NPT + 1 elements of the copy to memory per thread)
        const bool fits = nx * ny <= (int64_t)(npt <= 1 ? 2 : npt <= 2 ? 3 : 5) * mt;
        for (int64_t s0 = 0; s0 < n_solve; s0 += chunk) {
            const int64_t cnt = std::min(chunk, n_solve - s0);
            hipLaunchKernelGGL(k_reg_xy_coefs, dim3((unsigned)((interior + 255) / 256), (unsigned)(nz - 1), (unsigned)cnt),
                               dim3(256), 0, st, ra, r->d_xy, s0);
            if (!in_lds) hipLaunchKernelGGL(k_reg_xy_march_mem, dim3((unsigned)cnt), dim3((unsigned)mt), 0, st, ra, r->d_xy, s0);
            else if (npt <= 1 && fits) VRT_XY_MARCH(1);
            else if (npt <= 2 && fits) VRT_XY_MARCH(2);
            else if (npt <= 4 && fits) VRT_XY_MARCH(4);
            else VRT_XY_MARCH(0);
        }
#undef VRT_XY_MARCH
    } else if (threads <= 256)
        hipLaunchKernelGGL(k_regular_solve<256>, dim3((unsigned)n_solve), dim3((unsigned)threads), lds, st, ra);
    else
        hipLaunchKernelGGL(k_regular_solve<1024>, dim3((unsigned)n_solve), dim3((unsigned)threads), lds, st, ra);
    VRT_HIP_TRY(hipEventRecord(r->ev[2], st));
    for (int64_t s = 0; s < n_solve; s++)
        hipLaunchKernelGGL(k_reg_from_planes, dim3(tb), dim3(256), 0, st, (int)nz, (int)nx, (int)ny, r->d_I + s * vol, dI_out + s * vol);
    VRT_HIP_TRY(hipGetLastError());
    r->timed = true;
    return VRT_OK;
}

// milliseconds of the last execute's solve kernel alone (HIP events on its stream); the stream
// must have been synchronised
extern "C" int vrt_regular_last_solve_ms(const vrt_regular *r, double *ms)
{
    if (!r || !ms || !r->timed) return fail(VRT_EINVAL, "no timed execute yet");
    float f = 0;
    VRT_HIP_TRY(hipEventElapsedTime(&f, r->ev[1], r->ev[2]));
    *ms = f;
    return VRT_OK;
}

// Host-pointer form: stages the arrays through the device around vrt_regular_execute_dev.
extern "C" int vrt_short_characteristics(int64_t nz, int64_t nx, int64_t ny, const double *z,
                                         const double *x, const double *y, int64_t n_solve,
                                         const double *k, const int *up, const double *S,
                                         int64_t S_stride, const double *alpha, int64_t alpha_stride,
                                         const double *I0, int n_sweeps, int device, double *I_out)
{
    DeviceScope scope;
    if (!z || !x || !y || !k || !up || !S || !alpha || !I0 || !I_out) return fail(VRT_EINVAL, "NULL argument");
    if (nz < 2 || nx < 3 || ny < 3 || n_solve < 1 || n_sweeps < 1) return fail(VRT_EINVAL, "bad sizes");
    const int64_t vol = nz * nx * ny, plane = nx * ny;
    if ((S_stride != 0 && S_stride != vol) || (alpha_stride != 0 && alpha_stride != vol))
        return fail(VRT_EINVAL, "S_stride / alpha_stride must be 0 (shared) or nz*nx*ny");
    vrt_regular *r = nullptr;
    int rc = vrt_regular_create(nz, nx, ny, z, x, y, device, &r);
    if (rc) return rc;
    const int64_t nS = S_stride ? n_solve : 1, nA = alpha_stride ? n_solve : 1;
    double *d_S = nullptr, *d_A = nullptr, *d_I0 = nullptr, *d_out = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_S, (void *)d_A, (void *)d_I0, (void *)d_out})
            if (p) (void)hipFree(p);
        regular_free(r);
    };
#define REG_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            cleanup();                                                                     \
            return fail(_e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE,            \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                \
        }                                                                                  \
    } while (0)
    REG_TRY(hipMalloc((void **)&d_S, sizeof(double) * vol * nS));
    REG_TRY(hipMalloc((void **)&d_A, sizeof(double) * vol * nA));
    REG_TRY(hipMalloc((void **)&d_I0, sizeof(double) * plane * n_solve));
    REG_TRY(hipMalloc((void **)&d_out, sizeof(double) * vol * n_solve));
    REG_TRY(hipMemcpy(d_S, S, sizeof(double) * vol * nS, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_A, alpha, sizeof(double) * vol * nA, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_I0, I0, sizeof(double) * plane * n_solve, hipMemcpyHostToDevice));
    rc = vrt_regular_execute_dev(r, n_solve, k, up, d_S, S_stride, d_A, alpha_stride, 0, d_I0, n_sweeps, d_out, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    REG_TRY(hipDeviceSynchronize());
    REG_TRY(hipMemcpy(I_out, d_out, sizeof(double) * vol * n_solve, hipMemcpyDeviceToHost));
#undef REG_TRY
    cleanup();
    return VRT_OK;
}
