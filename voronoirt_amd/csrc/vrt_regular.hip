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

struct RegArgs {
    int nz, nx, ny, n_sweeps;
    const double *z, *x, *y;
    const double *k;          // (3, n_solve)
    const int *up;            // 1 = up, 0 = down
    const double *S, *alpha;  // plane-major, per solve stride below (0 = shared)
    int64_t S_stride, A_stride;
    const double *I0;         // (nx, ny) Julia order per solve: I0[ix + nx*iy]
    double *I;                // plane-major [solve][iz][iy][ix]
};

#define PL(p, ix, iy) (p)[(ix) + nx * (iy)]

__global__ void __launch_bounds__(1024)
k_regular_solve(RegArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) double row[];   // carried row / column
    const int nz = ra.nz, nx = ra.nx, ny = ra.ny;
    const int tid = threadIdx.x, T = blockDim.x;
    const int solve = blockIdx.x;
    const double k0 = ra.k[3 * solve], k1 = ra.k[3 * solve + 1], k2 = ra.k[3 * solve + 2];
    const bool up = ra.up[solve] != 0;
    const int64_t plane = (int64_t)nx * ny;
    const double *S = ra.S + (int64_t)solve * ra.S_stride;
    const double *Al = ra.alpha + (int64_t)solve * ra.A_stride;
    double *I = ra.I + (int64_t)solve * plane * nz;
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
                        PL(Ic, idx, idy) = (e * I_u + a * S_u) + b * PL(Sc, idx, idy);
                    }
                    __syncthreads();
                    if (tid == 0) {                               // ghost zones of the row
                        PL(Ic, idx, 0) = PL(Ic, idx, ny - 2);
                        PL(Ic, idx, ny - 1) = PL(Ic, idx, 1);
                    }
                    __syncthreads();
                    for (int j = tid; j < ny; j += T) row[j] = PL(Ic, idx, j);   // I_upper = I[idx, :]
                    __syncthreads();
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
                        PL(Ic, idx, idy) = (e * I_u + a * S_u) + b * PL(S_cen, idx, idy);
                    }
                    __syncthreads();
                    if (tid == 0) {                               // :704-705 / :822-823
                        PL(Ic, 0, idy) = PL(Ic, nx - 2, idy);
                        PL(Ic, nx - 1, idy) = PL(Ic, 1, idy);
                    }
                    __syncthreads();
                    for (int i = tid; i < nx; i += T) row[i] = PL(Ic, i, idy);   // I_upper = I[:, idy]
                    __syncthreads();
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

#undef PL

}  // namespace vrt

using namespace vrt;

extern "C" int vrt_short_characteristics(int64_t nz, int64_t nx, int64_t ny, const double *z,
                                         const double *x, const double *y, int64_t n_solve,
                                         const double *k, const int *up, const double *S,
                                         int64_t S_stride, const double *alpha, int64_t alpha_stride,
                                         const double *I0, int n_sweeps, int device, double *I_out)
{
    if (!z || !x || !y || !k || !up || !S || !alpha || !I0 || !I_out) return fail(VRT_EINVAL, "NULL argument");
    if (nz < 2 || nx < 3 || ny < 3 || n_solve < 1 || n_sweeps < 1) return fail(VRT_EINVAL, "bad sizes");
    if (nx > 16384 || ny > 16384) return fail(VRT_EINVAL, "nx, ny must be at most 16384");
    const int64_t vol = nz * nx * ny, plane = nx * ny;
    if ((S_stride != 0 && S_stride != vol) || (alpha_stride != 0 && alpha_stride != vol))
        return fail(VRT_EINVAL, "S_stride / alpha_stride must be 0 (shared) or nz*nx*ny");
    for (int64_t s = 0; s < n_solve; s++) {
        const double *ks = k + 3 * s;
        const double nrm = std::sqrt(ks[0] * ks[0] + ks[1] * ks[1] + ks[2] * ks[2]);
        if (!(std::fabs(nrm - 1.0) < 1e-6))            // functions.jl:432 @assert norm(k) ≈ 1
            return fail(VRT_EINVAL, "direction " + std::to_string(s + 1) + " is not a unit vector");
        if (ks[0] == 0.0) return fail(VRT_EINVAL, "horizontal ray (k_z = 0) has no upwind plane");
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(VRT_ENODEVICE, "no HIP device available (libvrt_hip has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(VRT_EINVAL, "device ordinal out of range");
    VRT_HIP_TRY(hipSetDevice(device));
    const int64_t nS = S_stride ? n_solve : 1, nA = alpha_stride ? n_solve : 1;
    double *d_in = nullptr, *d_S = nullptr, *d_A = nullptr, *d_I = nullptr, *d_I0 = nullptr, *d_g = nullptr,
           *d_k = nullptr;
    int *d_up = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_in, (void *)d_S, (void *)d_A, (void *)d_I, (void *)d_I0, (void *)d_g,
                        (void *)d_k, (void *)d_up})
            if (p) (void)hipFree(p);
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
    REG_TRY(hipMalloc((void **)&d_in, sizeof(double) * vol));
    REG_TRY(hipMalloc((void **)&d_S, sizeof(double) * vol * nS));
    REG_TRY(hipMalloc((void **)&d_A, sizeof(double) * vol * nA));
    REG_TRY(hipMalloc((void **)&d_I, sizeof(double) * vol * n_solve));
    REG_TRY(hipMalloc((void **)&d_I0, sizeof(double) * plane * n_solve));
    REG_TRY(hipMalloc((void **)&d_g, sizeof(double) * (nz + nx + ny)));
    REG_TRY(hipMalloc((void **)&d_k, sizeof(double) * 3 * n_solve));
    REG_TRY(hipMalloc((void **)&d_up, sizeof(int) * n_solve));
    const unsigned tb = (unsigned)((vol + 255) / 256);
    for (int64_t s = 0; s < nS; s++) {
        REG_TRY(hipMemcpy(d_in, S + s * vol, sizeof(double) * vol, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_reg_to_planes, dim3(tb), dim3(256), 0, 0, (int)nz, (int)nx, (int)ny, d_in, d_S + s * vol);
        REG_TRY(hipDeviceSynchronize());
    }
    for (int64_t s = 0; s < nA; s++) {
        REG_TRY(hipMemcpy(d_in, alpha + s * vol, sizeof(double) * vol, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_reg_to_planes, dim3(tb), dim3(256), 0, 0, (int)nz, (int)nx, (int)ny, d_in, d_A + s * vol);
        REG_TRY(hipDeviceSynchronize());
    }
    REG_TRY(hipMemcpy(d_I0, I0, sizeof(double) * plane * n_solve, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_g, z, sizeof(double) * nz, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_g + nz, x, sizeof(double) * nx, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_g + nz + nx, y, sizeof(double) * ny, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_k, k, sizeof(double) * 3 * n_solve, hipMemcpyHostToDevice));
    REG_TRY(hipMemcpy(d_up, up, sizeof(int) * n_solve, hipMemcpyHostToDevice));
    RegArgs ra;
    ra.nz = (int)nz; ra.nx = (int)nx; ra.ny = (int)ny; ra.n_sweeps = n_sweeps;
    ra.z = d_g; ra.x = d_g + nz; ra.y = d_g + nz + nx;
    ra.k = d_k; ra.up = d_up;
    ra.S = d_S; ra.alpha = d_A; ra.S_stride = S_stride; ra.A_stride = alpha_stride;
    ra.I0 = d_I0; ra.I = d_I;
    const size_t lds = sizeof(double) * (size_t)std::max(nx, ny);
    hipLaunchKernelGGL(k_regular_solve, dim3((unsigned)n_solve), dim3(1024), lds, 0, ra);
    REG_TRY(hipGetLastError());
    REG_TRY(hipDeviceSynchronize());
    for (int64_t s = 0; s < n_solve; s++) {
        hipLaunchKernelGGL(k_reg_from_planes, dim3(tb), dim3(256), 0, 0, (int)nz, (int)nx, (int)ny, d_I + s * vol, d_in);
        REG_TRY(hipMemcpy(I_out + s * vol, d_in, sizeof(double) * vol, hipMemcpyDeviceToHost));
    }
#undef REG_TRY
    cleanup();
    return VRT_OK;
}
