// Multi-device object of the C ABI: ONE host process (the reference's Julia driver is one)
// uses several GPUs of a node.  The object owns a grid + plan per device and, when the devices are distinct,
// an in-process RCCL communicator (ncclCommInitAll; the library is dlopen'ed on first use, libvrt_hip.so does
// not link it).  The angle x wavelength loop of J_λ_voronoi (src/lambda_iteration.jl:84-111) is sharded the
// way voronoirt_amd/distributed.py shards it across processes:
//   "lambda"  nλ >= devices: contiguous wavelength blocks (51 over 8 -> 7,7,7,6,6,6,6,6), every device solves
//             all angles of its block and owns whole rows J[l, :]: no exchange, the blocks go home by strided copies
//   "angle"   nλ < devices (or forced): the angles are dealt to the devices (ups and downs separately), the
//             partial J's are summed by ONE RCCL all-reduce -- the north star's scheme
// Two handles on the SAME device (rehearsal on a one-GPU box; RCCL refuses that) sum through a kernel instead.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "vrt_internal.h"

using namespace vrt;

namespace {

// The few RCCL entry points this file calls, declared here: librccl is dlopen'ed on first use, and the library builds
// (and serves one device) on a ROCm installation without the RCCL development headers.  The values are NCCL's ABI.
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclDouble = 8, ncclSum = 0;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllReduce && Reduce && GroupStart && GroupEnd && GetErrorString;
    }
};

struct Member {
    int device = 0;
    vrt_grid *grid = nullptr;
    vrt_plan *plan_all = nullptr;          // every angle (lambda mode)
    vrt_plan *plan_part = nullptr;         // this device's angles (angle mode), built on first use
    std::vector<int> my_angles;
    hipStream_t stream = nullptr;
    double *dS = nullptr, *dA = nullptr, *dU = nullptr, *dD = nullptr, *dJ = nullptr, *dV = nullptr;
    size_t cS = 0, cA = 0, cU = 0, cD = 0, cJ = 0, cV = 0;
    int rc = VRT_OK;
    std::string err;
};

int ensure(double *&buf, size_t &cap, size_t count)
{
    if (buf && count <= cap) return VRT_OK;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&buf, std::max<size_t>(count, 1) * sizeof(double));
    if (e != hipSuccess) {
        buf = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    cap = count;
    return VRT_OK;
}

}  // namespace

struct vrt_multi {
    std::vector<Member> m;
    int64_t n = 0, n_angles = 0;
    int n_sweeps = 3;
    std::vector<double> k;
    std::vector<int> dirs;
    bool distinct = true;                   // all devices different: RCCL; otherwise the same-device rehearsal
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    int shard = 0;                          // 0 auto, 1 lambda, 2 angle
    int last_shard = 0;
    std::mutex mu;
};

static void multi_free(vrt_multi *mm)
{
    if (!mm) return;
    for (size_t d = 0; d < mm->m.size(); d++) {
        Member &me = mm->m[d];
        if (!me.grid && !me.stream) continue;                // never got a device (creation failed there)
        (void)hipSetDevice(me.device);
        if (d < mm->comms.size() && mm->comms[d] && mm->rccl.CommDestroy) (void)mm->rccl.CommDestroy(mm->comms[d]);
        for (double *q : {me.dS, me.dA, me.dU, me.dD, me.dJ, me.dV})
            if (q) (void)hipFree(q);
        if (me.stream) (void)hipStreamDestroy(me.stream);
        if (me.plan_part) vrt_plan_destroy(me.plan_part);
        if (me.plan_all) vrt_plan_destroy(me.plan_all);
        if (me.grid) vrt_grid_destroy(me.grid);
    }
    delete mm;
}

// contiguous block partition: the first n_units % world ranks get one extra unit (distributed.partition)
static void block_of(int64_t n_units, int world, int rank, int64_t &start, int64_t &stop)
{
    const int64_t base = n_units / world, extra = n_units % world;
    start = rank * base + std::min<int64_t>(rank, extra);
    stop = start + base + (rank < extra ? 1 : 0);
}

extern "C" {

int vrt_multi_create(int n_devices, const int *devices, int64_t n, const double *pos_zxy, const int64_t *nbr, int64_t D1,
                     const double bounds[6], int64_t n_angles, const double *k, const int *dirs, int n_sweeps,
                     vrt_multi **out)
{
    DeviceScope scope;
    if (!out) return fail(VRT_EINVAL, "out is NULL");
    *out = nullptr;
    if (!devices || !pos_zxy || !nbr || !bounds || !k) return fail(VRT_EINVAL, "NULL argument");
    if (n_devices < 1 || n_devices > 64) return fail(VRT_EINVAL, "need 1 <= n_devices <= 64");
    if (n_angles < 1) return fail(VRT_EINVAL, "n_angles must be >= 1");
    try {
        vrt_multi *mm = new (std::nothrow) vrt_multi();
        if (!mm) return fail(VRT_ENOMEM, "out of host memory");
        mm->n = n;
        mm->n_angles = n_angles;
        mm->n_sweeps = n_sweeps;
        mm->k.assign(k, k + 3 * n_angles);
        mm->dirs.resize((size_t)n_angles);
        for (int64_t a = 0; a < n_angles; a++)
            mm->dirs[(size_t)a] = dirs ? (dirs[a] > 0 ? 1 : (dirs[a] < 0 ? -1 : 0))
                                       : (std::fabs(k[3 * a]) < 1e-12 ? 0 : (k[3 * a] < 0 ? 1 : -1));
        mm->m.resize((size_t)n_devices);
        for (int d = 0; d < n_devices; d++) {
            mm->m[(size_t)d].device = devices[d];
            for (int e = 0; e < d; e++)
                if (devices[e] == devices[d]) mm->distinct = false;
        }
        // one grid + all-angle plan per device, built concurrently (plan creation is host-side schedule work)
        if (!run_workers(n_devices, [&](int d) {
                Member &me = mm->m[(size_t)d];
                me.rc = vrt_grid_create(n, pos_zxy, nbr, D1, bounds, me.device, &me.grid);
                if (!me.rc) me.rc = vrt_plan_create_ex(me.grid, n_angles, k, mm->dirs.data(), n_sweeps, &me.plan_all);
                if (!me.rc && (hipSetDevice(me.device) != hipSuccess ||
                               hipStreamCreateWithFlags(&me.stream, hipStreamNonBlocking) != hipSuccess))
                    me.rc = fail(VRT_ENODEVICE, "cannot create a stream");
                if (me.rc) me.err = vrt_last_error();
            })) {
            multi_free(mm);
            return fail(VRT_ENOMEM, "out of host memory while creating the per-device plans");
        }
        for (const Member &me : mm->m)
            if (me.rc) {
                const int rc = me.rc;
                const std::string msg = "device " + std::to_string(me.device) + ": " + me.err;
                multi_free(mm);
                return fail(rc, msg);
            }
        // angles of the "angle" mode: ups and downs dealt round-robin separately (distributed.angle_assignment)
        {
            int ju = 0, jd = 0;
            for (int64_t a = 0; a < n_angles; a++) {
                if (mm->dirs[(size_t)a] > 0) mm->m[(size_t)(ju++ % n_devices)].my_angles.push_back((int)a);
                else if (mm->dirs[(size_t)a] < 0) mm->m[(size_t)((n_devices - 1 - (jd++ % n_devices)))].my_angles.push_back((int)a);
            }
            for (Member &me : mm->m) std::sort(me.my_angles.begin(), me.my_angles.end());
        }
        // VRT_MULTI_FORCE_RCCL=1: a communicator also for ONE device (a one-rank ncclCommInitAll), so that the RCCL legs of this
        // file -- ncclReduce of the angle shards, ncclAllReduce of the rate-integral shares, the group calls -- execute on a
        // one-GPU box too (tests/test_physics.py); read here, once per object
        const char *force = std::getenv("VRT_MULTI_FORCE_RCCL");
        const bool force_rccl = force && force[0] == '1';
        if (mm->distinct && (n_devices > 1 || force_rccl)) {
            if (!mm->rccl.load()) {
                multi_free(mm);
                return fail(VRT_ENODEVICE, "cannot load librccl.so (needed for more than one device)");
            }
            mm->comms.assign((size_t)n_devices, nullptr);
            const ncclResult_t r = mm->rccl.CommInitAll(mm->comms.data(), n_devices, devices);
            if (r != ncclSuccess) {
                const std::string msg = std::string("ncclCommInitAll: ") + mm->rccl.GetErrorString(r);
                mm->comms.clear();
                multi_free(mm);
                return fail(VRT_ENODEVICE, msg);
            }
        }
        *out = mm;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_multi_set_shard(vrt_multi *mm, const char *mode)
{
    if (!mm || !mode) return fail(VRT_EINVAL, "NULL argument");
    const std::string s(mode);
    std::lock_guard<std::mutex> lock(mm->mu);
    if (s == "auto") mm->shard = 0;
    else if (s == "lambda") mm->shard = 1;
    else if (s == "angle") mm->shard = 2;
    else return fail(VRT_EINVAL, "shard must be auto, lambda or angle");
    return VRT_OK;
}

int vrt_multi_last_shard(const vrt_multi *mm) { return mm ? mm->last_shard : 0; }
int vrt_multi_uses_rccl(const vrt_multi *mm) { return mm && !mm->comms.empty() ? 1 : 0; }

int vrt_multi_execute(vrt_multi *mm, int64_t nlam, int64_t ld, const double *S, const double *alpha, int alpha_mode,
                      const double *I0_up, const double *I0_down, const double *weights, double *J)
{
    DeviceScope scope;
    if (!mm || !S || !alpha || !weights || !J) return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (alpha_mode < 0 || alpha_mode > 2) return fail(VRT_EINVAL, "bad alpha_mode (host arrays: 0, 1 or 2)");
    try {
        std::lock_guard<std::mutex> lock(mm->mu);
        const int W = (int)mm->m.size();
        const int64_t n = mm->n, A = mm->n_angles;
        int shard = mm->shard;
        if (shard == 0) shard = nlam >= W ? 1 : 2;
        if (shard == 2 && W > A) return fail(VRT_EINVAL, "more devices than angles");
        mm->last_shard = shard;
        const int64_t n1u = mm->m[0].grid->up.n1, n1d = mm->m[0].grid->down.n1;
        // angle mode: the devices' partial plans
        if (shard == 2)
            for (Member &me : mm->m)
                if (!me.plan_part && !me.my_angles.empty()) {
                    std::vector<double> kk;
                    std::vector<int> dd;
                    for (int a : me.my_angles) {
                        kk.insert(kk.end(), mm->k.begin() + 3 * a, mm->k.begin() + 3 * a + 3);
                        dd.push_back(mm->dirs[(size_t)a]);
                    }
                    int rc = vrt_plan_create_ex(me.grid, (int64_t)me.my_angles.size(), kk.data(), dd.data(), mm->n_sweeps, &me.plan_part);
                    if (rc) return rc;
                }
        auto work = [&](int d) {
            Member &me = mm->m[(size_t)d];
            me.rc = VRT_OK;
            auto chk = [&](hipError_t e, const char *what) {
                if (e != hipSuccess && !me.rc) {
                    me.rc = VRT_ENODEVICE;
                    me.err = std::string(what) + ": " + hipGetErrorString(e);
                }
            };
            chk(hipSetDevice(me.device), "hipSetDevice");
            if (me.rc) return;
            hipStream_t st = me.stream;
            int64_t l0 = 0, l1 = nlam;
            if (shard == 1) block_of(nlam, W, d, l0, l1);
            const int64_t nb = l1 - l0;
            const size_t w8 = sizeof(double);
            if (nb <= 0 || (shard == 2 && me.my_angles.empty())) {       // nothing to do here: contributes zeros in angle mode
                if (shard == 2) {
                    if ((me.rc = ensure(me.dJ, me.cJ, (size_t)n * (size_t)nlam))) { me.err = vrt_last_error(); return; }
                    chk(hipMemsetAsync(me.dJ, 0, w8 * (size_t)n * (size_t)nlam, st), "hipMemsetAsync");
                    chk(hipStreamSynchronize(st), "hipStreamSynchronize");      // another member's stream reads these zeros
                }
                return;
            }
            vrt_plan *plan = shard == 1 ? me.plan_all : me.plan_part;
            const int64_t nA = shard == 1 ? A : (int64_t)me.my_angles.size();
            // S block (nb, n) dense on the device
            if ((me.rc = ensure(me.dS, me.cS, (size_t)n * (size_t)nb)) || (me.rc = ensure(me.dJ, me.cJ, (size_t)n * (size_t)nb))) {
                me.err = vrt_last_error();
                return;
            }
            chk(hipMemcpy2DAsync(me.dS, w8 * (size_t)nb, S + l0, w8 * (size_t)ld, w8 * (size_t)nb, (size_t)n, hipMemcpyHostToDevice, st), "upload S");
            // alpha
            const double *dA = nullptr;
            if (alpha_mode == VRT_ALPHA_SITE) {
                if ((me.rc = ensure(me.dA, me.cA, (size_t)n))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpyAsync(me.dA, alpha, w8 * (size_t)n, hipMemcpyHostToDevice, st), "upload alpha");
            } else if (alpha_mode == VRT_ALPHA_SITE_LAM) {
                if ((me.rc = ensure(me.dA, me.cA, (size_t)n * (size_t)nb))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpy2DAsync(me.dA, w8 * (size_t)nb, alpha + l0, w8 * (size_t)ld, w8 * (size_t)nb, (size_t)n, hipMemcpyHostToDevice, st), "upload alpha");
            } else {
                if ((me.rc = ensure(me.dA, me.cA, (size_t)nA * (size_t)n * (size_t)nb))) { me.err = vrt_last_error(); return; }
                for (int64_t j = 0; j < nA; j++) {
                    const int64_t a = shard == 1 ? j : me.my_angles[(size_t)j];
                    chk(hipMemcpy2DAsync(me.dA + (size_t)j * (size_t)n * (size_t)nb, w8 * (size_t)nb,
                                         alpha + (size_t)a * (size_t)n * (size_t)ld + l0, w8 * (size_t)ld, w8 * (size_t)nb, (size_t)n,
                                         hipMemcpyHostToDevice, st), "upload alpha");
                }
            }
            dA = me.dA;
            double *dU = nullptr, *dD = nullptr;
            if (I0_up && n1u) {
                if ((me.rc = ensure(me.dU, me.cU, (size_t)n1u * (size_t)nb))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpy2DAsync(me.dU, w8 * (size_t)nb, I0_up + l0, w8 * (size_t)nlam, w8 * (size_t)nb, (size_t)n1u, hipMemcpyHostToDevice, st), "upload I0");
                dU = me.dU;
            }
            if (I0_down && n1d) {
                if ((me.rc = ensure(me.dD, me.cD, (size_t)n1d * (size_t)nb))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpy2DAsync(me.dD, w8 * (size_t)nb, I0_down + l0, w8 * (size_t)nlam, w8 * (size_t)nb, (size_t)n1d, hipMemcpyHostToDevice, st), "upload I0");
                dD = me.dD;
            }
            if (me.rc) return;
            std::vector<double> wv;
            if (shard == 1) wv.assign(weights, weights + A);
            else
                for (int a : me.my_angles) wv.push_back(weights[a]);
            me.rc = vrt_plan_execute_dev(plan, nb, nb, me.dS, dA, alpha_mode, dU, dD, wv.data(), me.dJ, nullptr, st);
            if (me.rc) { me.err = vrt_last_error(); return; }
            if (shard == 1)     // the device owns rows l0..l1 of J: straight home
                chk(hipMemcpy2DAsync(J + l0, w8 * (size_t)ld, me.dJ, w8 * (size_t)nb, w8 * (size_t)nb, (size_t)n, hipMemcpyDeviceToHost, st), "download J");
            chk(hipStreamSynchronize(st), "hipStreamSynchronize");
            // (a chained sweep that gave up waiting: reported by THIS call, whose J it spoiled)
            if (!me.rc && (me.rc = patch_chain_check(plan))) me.err = vrt_last_error();
        };
        if (!run_workers(W, work)) return fail(VRT_ENOMEM, "out of host memory in a device worker");
        for (const Member &me : mm->m)
            if (me.rc) return fail(me.rc, "device " + std::to_string(me.device) + ": " + me.err);
        if (shard == 2) {
            // J = Σ over the devices' partial sums, wanted on ONE device only (the host array is filled from device 0):
            // one RCCL reduce to rank 0 over xGMI -- half the bytes of an all-reduce -- or, on a shared device, adds
            const size_t cnt = (size_t)n * (size_t)nlam;
            if (!mm->comms.empty()) {
                // (nothing may return between GroupStart and GroupEnd: an open group poisons every later call)
                ncclResult_t r = mm->rccl.GroupStart();
                hipError_t he = hipSuccess;
                for (int d = 0; d < W && r == ncclSuccess && he == hipSuccess; d++) {
                    Member &me = mm->m[(size_t)d];
                    he = hipSetDevice(me.device);
                    if (he == hipSuccess)
                        r = mm->rccl.Reduce(me.dJ, me.dJ, cnt, ncclDouble, ncclSum, 0, mm->comms[(size_t)d], me.stream);
                }
                const ncclResult_t r2 = mm->rccl.GroupEnd();
                if (he != hipSuccess) return fail(VRT_ENODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he));
                if (r != ncclSuccess || r2 != ncclSuccess)
                    return fail(VRT_ENODEVICE, std::string("ncclReduce: ") + mm->rccl.GetErrorString(r != ncclSuccess ? r : r2));
                for (Member &me : mm->m) {
                    VRT_HIP_TRY(hipSetDevice(me.device));
                    VRT_HIP_TRY(hipStreamSynchronize(me.stream));
                }
            } else {
                Member &m0 = mm->m[0];
                VRT_HIP_TRY(hipSetDevice(m0.device));
                (void)hipGetLastError();
                for (int d = 1; d < W; d++)
                    if (int rc = launch_axpy(cnt, mm->m[(size_t)d].dJ, m0.dJ, m0.stream)) return rc;     // same device: plain adds
            }
            Member &m0 = mm->m[0];
            VRT_HIP_TRY(hipSetDevice(m0.device));
            VRT_HIP_TRY(hipMemcpy2DAsync(J, sizeof(double) * (size_t)ld, m0.dJ, sizeof(double) * (size_t)nlam,
                                         sizeof(double) * (size_t)nlam, (size_t)n, hipMemcpyDeviceToHost, m0.stream));
            VRT_HIP_TRY(hipStreamSynchronize(m0.stream));
        }
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

// ---- the line case across several devices ---------------------------------------------------------------------------
// vrt_multi_execute_line: J_λ_voronoi of the line case (src/lambda_iteration.jl:72-111) from host arrays, the wavelengths
// in contiguous blocks over the devices: every device makes the per-angle α_tot of ITS wavelengths itself
// (vrt_line_opacity's kernel), so the (nλ, n, n_angles) array neither exists on the host nor crosses PCIe, and owns
// whole rows J[l, :] -- no exchange.
int vrt_multi_execute_line(vrt_multi *mm, int64_t nlam, int64_t ld, const double *lambda, double lambda0, double c0,
                           const double *velocity, const double *doppler_width, const double *gamma,
                           const double *line_strength, const double *alpha_cont, const double *S, const double *I0_up,
                           const double *I0_down, const double *weights, double *J)
{
    DeviceScope scope;
    if (!mm || !lambda || !velocity || !doppler_width || !gamma || !line_strength || !alpha_cont || !S || !weights || !J)
        return fail(VRT_EINVAL, "NULL argument");
    if (nlam < 1 || ld < nlam) return fail(VRT_EINVAL, "need nlam >= 1 and ld >= nlam");
    if (!(lambda0 > 0) || !(c0 > 0)) return fail(VRT_EINVAL, "lambda0 and c0 must be positive");
    try {
        std::lock_guard<std::mutex> lock(mm->mu);
        const int W = (int)mm->m.size();
        const int64_t n = mm->n;
        const int64_t n1u = mm->m[0].grid->up.n1, n1d = mm->m[0].grid->down.n1;
        mm->last_shard = 1;
        auto work = [&](int d) {
            Member &me = mm->m[(size_t)d];
            me.rc = VRT_OK;
            auto chk = [&](hipError_t e, const char *what) {
                if (e != hipSuccess && !me.rc) {
                    me.rc = VRT_ENODEVICE;
                    me.err = std::string(what) + ": " + hipGetErrorString(e);
                }
            };
            int64_t l0, l1;
            block_of(nlam, W, d, l0, l1);
            const int64_t nb = l1 - l0;
            if (nb <= 0) return;
            chk(hipSetDevice(me.device), "hipSetDevice");
            if (me.rc) return;
            hipStream_t st = me.stream;
            vrt_plan *plan = me.plan_all;
            std::lock_guard<std::mutex> plock(plan->mu);
            const size_t w8 = sizeof(double), sn = (size_t)n;
            const size_t nnat = (size_t)vrt_plan_native_alpha_count(plan, nb);
            // S | seven per-site vectors + the block's wavelengths | native alpha | J
            if ((me.rc = ensure(me.dS, me.cS, sn * (size_t)nb)) || (me.rc = ensure(me.dJ, me.cJ, sn * (size_t)nb)) ||
                (me.rc = ensure(me.dA, me.cA, nnat)) || (me.rc = ensure(me.dV, me.cV, 7 * sn + (size_t)nb))) {
                me.err = vrt_last_error();
                return;
            }
            double *d_vel = me.dV, *d_dop = me.dV + 3 * sn, *d_gam = me.dV + 4 * sn, *d_str = me.dV + 5 * sn, *d_ac = me.dV + 6 * sn,
                   *d_lam = me.dV + 7 * sn;
            chk(hipMemcpy2DAsync(me.dS, w8 * (size_t)nb, S + l0, w8 * (size_t)ld, w8 * (size_t)nb, sn, hipMemcpyHostToDevice, st), "upload S");
            chk(hipMemcpyAsync(d_vel, velocity, w8 * 3 * sn, hipMemcpyHostToDevice, st), "upload velocity");
            chk(hipMemcpyAsync(d_dop, doppler_width, w8 * sn, hipMemcpyHostToDevice, st), "upload doppler");
            chk(hipMemcpyAsync(d_gam, gamma, w8 * sn, hipMemcpyHostToDevice, st), "upload gamma");
            chk(hipMemcpyAsync(d_str, line_strength, w8 * sn, hipMemcpyHostToDevice, st), "upload strength");
            chk(hipMemcpyAsync(d_ac, alpha_cont, w8 * sn, hipMemcpyHostToDevice, st), "upload alpha_cont");
            chk(hipMemcpyAsync(d_lam, lambda + l0, w8 * (size_t)nb, hipMemcpyHostToDevice, st), "upload lambda");
            double *dU = nullptr, *dD = nullptr;
            if (I0_up && n1u) {
                if ((me.rc = ensure(me.dU, me.cU, (size_t)n1u * (size_t)nb))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpy2DAsync(me.dU, w8 * (size_t)nb, I0_up + l0, w8 * (size_t)nlam, w8 * (size_t)nb, (size_t)n1u, hipMemcpyHostToDevice, st), "upload I0");
                dU = me.dU;
            }
            if (I0_down && n1d) {
                if ((me.rc = ensure(me.dD, me.cD, (size_t)n1d * (size_t)nb))) { me.err = vrt_last_error(); return; }
                chk(hipMemcpy2DAsync(me.dD, w8 * (size_t)nb, I0_down + l0, w8 * (size_t)nlam, w8 * (size_t)nb, (size_t)n1d, hipMemcpyHostToDevice, st), "upload I0");
                dD = me.dD;
            }
            if (me.rc) return;
            if ((me.rc = launch_line_opacity(plan, nb, d_lam, lambda0, c0, d_vel, d_dop, d_gam, d_str, d_ac, me.dA, st)) ||
                (me.rc = execute_dev_locked(plan, nb, nb, me.dS, me.dA, VRT_ALPHA_ANGLE_NATIVE, dU, dD, weights, me.dJ, nullptr, st))) {
                me.err = vrt_last_error();
                return;
            }
            chk(hipMemcpy2DAsync(J + l0, w8 * (size_t)ld, me.dJ, w8 * (size_t)nb, w8 * (size_t)nb, sn, hipMemcpyDeviceToHost, st), "download J");
            chk(hipStreamSynchronize(st), "hipStreamSynchronize");
            if (!me.rc && (me.rc = patch_chain_check(plan))) me.err = vrt_last_error();
        };
        if (!run_workers(W, work)) return fail(VRT_ENOMEM, "out of host memory in a device worker");
        for (const Member &me : mm->m)
            if (me.rc) return fail(me.rc, "device " + std::to_string(me.device) + ": " + me.err);
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

}  // extern "C"

// ---- Λ-iteration session across several devices (BASELINE configs[3]; src/lambda_iteration.jl:205-300) ---------------
// Every device owns a contiguous block of the wavelengths (91 over 8 -> 12,12,12,11,11,11,11,11) and the whole per-site
// state.  Per iteration it runs, for ITS wavelengths: the line terms of the current populations, α_tot of every
// angle, the sweep, S_new with its share of the convergence maximum, and its share of the six λ-integrals of the
// radiative rates (rates.jl:154-201).  The shares are summed by ONE all-reduce of 6 n doubles over xGMI (SURVEY 8e:
// "only R and one scalar are reduced"); J never travels.  Every device then solves the statistical equilibrium of
// every site itself (populations.jl:191-221), so the next iteration's opacity needs no further exchange.
struct LambdaMember {
    int device = 0;
    int64_t l0 = 0, l1 = 0;
    double *d_small = nullptr;          // lambda | planck2 | sigma_bf1 | sigma_bf2 of ALL wavelengths
    double *d_velocity = nullptr, *d_doppler = nullptr, *d_gamma_static = nullptr, *d_gamma_unsold = nullptr,
           *d_alpha_cont = nullptr, *d_eps = nullptr, *d_temperature = nullptr, *d_atom = nullptr, *d_lte = nullptr,
           *d_C = nullptr;
    double *d_B0 = nullptr, *d_S_old = nullptr, *d_S_new = nullptr, *d_J = nullptr, *d_I0 = nullptr, *d_native = nullptr;   // this block's columns
    double *d_gamma = nullptr, *d_strength = nullptr, *d_pops = nullptr, *d_R = nullptr, *d_shares = nullptr;
    unsigned long long *d_scalars = nullptr;
    hipEvent_t ev = nullptr;
    // the block's S and J in sweep order between the steps (as the one-device session keeps them: vrt_lambda.cpp)
    double *d_S_nat[2] = {nullptr, nullptr}, *d_J_nat[2] = {nullptr, nullptr}, *d_B_up = nullptr;
};

struct vrt_multi_lambda {
    vrt_multi *mm = nullptr;
    int64_t n = 0, nlam = 0;
    int64_t blocks[6] = {0, 0, 0, 0, 0, 0};
    double lambda0 = 0, c0 = 0, strength_const = 0, Bij = 0, Bji = 0, sigma_bb_const = 0, hc_over_kB = 0, pref_ij = 0,
           pref_ji = 0;
    std::vector<double> weights;
    std::vector<LambdaMember> lm;
    int iterations = 0;
    bool native = false;                // every member keeps its block in sweep order (VRT_LAMBDA_NATIVE, all plans fit)
};

static void multi_lambda_free(vrt_multi_lambda *s)
{
    if (!s) return;
    for (size_t d = 0; d < s->lm.size(); d++) {
        LambdaMember &l = s->lm[d];
        (void)hipSetDevice(l.device);                        // (its own copy: the vrt_multi may be gone already)
        for (double *q : {l.d_small, l.d_velocity, l.d_doppler, l.d_gamma_static, l.d_gamma_unsold, l.d_alpha_cont, l.d_eps,
                          l.d_temperature, l.d_atom, l.d_lte, l.d_C, l.d_B0, l.d_S_old, l.d_S_new, l.d_J, l.d_I0, l.d_native,
                          l.d_gamma, l.d_strength, l.d_pops, l.d_R, l.d_shares, l.d_S_nat[0], l.d_S_nat[1], l.d_J_nat[0], l.d_J_nat[1], l.d_B_up})
            if (q) (void)hipFree(q);
        if (l.d_scalars) (void)hipFree(l.d_scalars);
        if (l.ev) (void)hipEventDestroy(l.ev);
    }
    delete s;
}

namespace {

int dmalloc(double **p, size_t count)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(double));
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(e == hipErrorOutOfMemory ? VRT_ENOMEM : VRT_ENODEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    return VRT_OK;
}
int dupload(double **d, const double *h, size_t count, hipStream_t st)
{
    int rc = dmalloc(d, count);
    if (rc) return rc;
    VRT_HIP_TRY(hipMemcpyAsync(*d, h, sizeof(double) * count, hipMemcpyHostToDevice, st));
    return VRT_OK;
}
// columns [l0, l1) of a host array (rows, nlam) into a dense device block (rows, l1 - l0)
int dupload_cols(double **d, const double *h, size_t rows, int64_t nlam, int64_t l0, int64_t l1, hipStream_t st)
{
    const size_t nb = (size_t)(l1 - l0);
    int rc = dmalloc(d, rows * nb);
    if (rc || nb == 0) return rc;
    VRT_HIP_TRY(hipMemcpy2DAsync(*d, sizeof(double) * nb, h + l0, sizeof(double) * (size_t)nlam, sizeof(double) * nb, rows, hipMemcpyHostToDevice, st));
    return VRT_OK;
}

}  // namespace

extern "C" {

int vrt_multi_lambda_create(vrt_multi *mm, const vrt_line_case *lc, const double *weights, vrt_multi_lambda **out)
{
    DeviceScope scope;
    if (!out) return fail(VRT_EINVAL, "out is NULL");
    *out = nullptr;
    if (!mm || !lc || !weights) return fail(VRT_EINVAL, "NULL argument");
    const int64_t nlam = lc->nlam;
    if (nlam < 2) return fail(VRT_EINVAL, "nlam must be >= 2");
    if (!lc->lambda || !lc->velocity || !lc->doppler_width || !lc->gamma_static || !lc->gamma_unsold || !lc->alpha_cont ||
        !lc->eps || !lc->temperature || !lc->atom_density || !lc->B0 || !lc->lte_populations || !lc->C || !lc->planck2 ||
        !lc->sigma_bf1 || !lc->sigma_bf2)
        return fail(VRT_EINVAL, "NULL array in the line case");
    for (int b = 0; b < 3; b++)
        if (lc->blocks[2 * b] < 0 || lc->blocks[2 * b + 1] > nlam || lc->blocks[2 * b + 1] - lc->blocks[2 * b] < 2)
            return fail(VRT_EINVAL, "each wavelength block needs at least two wavelengths inside [0, nlam)");
    if (!(lc->lambda0 > 0) || !(lc->c0 > 0)) return fail(VRT_EINVAL, "lambda0 and c0 must be positive");
    try {
        std::lock_guard<std::mutex> lock(mm->mu);
        const int W = (int)mm->m.size();
        for (const Member &me : mm->m) {
            const vrt_plan *p = me.plan_all;
            if (!p->patch_ok && (!p->tile_ok || p->tile_max_layer_size > steps_max_layer(false)))
                return fail(VRT_EINVAL, "the line session needs a layer path (at most 4 visits per site and 255 levels per layer)");
            if (p->A != (int)p->n_angles_user)
                return fail(VRT_EINVAL, "per-angle alpha needs every angle active (no θ = 90 direction)");
        }
        vrt_multi_lambda *s = new (std::nothrow) vrt_multi_lambda();
        if (!s) return fail(VRT_ENOMEM, "out of host memory");
        s->mm = mm;
        s->n = mm->n;
        s->nlam = nlam;
        for (int q = 0; q < 6; q++) s->blocks[q] = lc->blocks[q];
        s->lambda0 = lc->lambda0; s->c0 = lc->c0; s->strength_const = lc->strength_const; s->Bij = lc->Bij; s->Bji = lc->Bji;
        s->sigma_bb_const = lc->sigma_bb_const; s->hc_over_kB = lc->hc_over_kB; s->pref_ij = lc->pref_ij; s->pref_ji = lc->pref_ji;
        s->weights.assign(weights, weights + mm->n_angles);
        s->lm.resize((size_t)W);
        s->native = true;
        for (const Member &me : mm->m)
            s->native = s->native && me.plan_all->tune.lambda_native != 0 && native_planes_ok(me.plan_all) == VRT_OK &&
                        me.plan_all->tune.path != 1 && me.plan_all->tune.path != 2;
        const size_t n = (size_t)mm->n, nl = (size_t)nlam;
        const size_t nb1 = (size_t)(lc->blocks[3] - lc->blocks[2]), nb2 = (size_t)(lc->blocks[5] - lc->blocks[4]);
        std::vector<double> small;
        small.insert(small.end(), lc->lambda, lc->lambda + nl);
        small.insert(small.end(), lc->planck2, lc->planck2 + nl);
        small.insert(small.end(), lc->sigma_bf1, lc->sigma_bf1 + nb1);
        small.insert(small.end(), lc->sigma_bf2, lc->sigma_bf2 + nb2);
        std::vector<int> rcs((size_t)W, VRT_OK);
        std::vector<std::string> errs((size_t)W);
        auto work = [&](int d) {
            Member &me = mm->m[(size_t)d];
            LambdaMember &l = s->lm[(size_t)d];
            l.device = me.device;
            block_of(nlam, W, d, l.l0, l.l1);
            const size_t nb = (size_t)(l.l1 - l.l0);
            int rc = VRT_OK;
            if (hipSetDevice(me.device) != hipSuccess) { rcs[(size_t)d] = VRT_ENODEVICE; errs[(size_t)d] = "hipSetDevice"; return; }
            hipStream_t st = me.stream;
            vrt_grid *g = me.grid;
#define VRT_S(expr) do { if (!rc) rc = (expr); } while (0)
            VRT_S(dupload(&l.d_small, small.data(), small.size(), st));
            VRT_S(dupload(&l.d_velocity, lc->velocity, 3 * n, st));
            VRT_S(dupload(&l.d_doppler, lc->doppler_width, n, st));
            VRT_S(dupload(&l.d_gamma_static, lc->gamma_static, n, st));
            VRT_S(dupload(&l.d_gamma_unsold, lc->gamma_unsold, n, st));
            VRT_S(dupload(&l.d_alpha_cont, lc->alpha_cont, n, st));
            VRT_S(dupload(&l.d_eps, lc->eps, n, st));
            VRT_S(dupload(&l.d_temperature, lc->temperature, n, st));
            VRT_S(dupload(&l.d_atom, lc->atom_density, n, st));
            VRT_S(dupload(&l.d_lte, lc->lte_populations, 3 * n, st));
            VRT_S(dupload(&l.d_C, lc->C, 9 * n, st));
            VRT_S(dupload(&l.d_pops, lc->lte_populations, 3 * n, st));               // populations = copy(LTE_pops), :232
            VRT_S(dupload_cols(&l.d_B0, lc->B0, n, nlam, l.l0, l.l1, st));
            VRT_S(dupload_cols(&l.d_S_new, lc->B0, n, nlam, l.l0, l.l1, st));        // S_new = B_0, :236-239
            VRT_S(dmalloc(&l.d_S_old, n * nb));
            VRT_S(dmalloc(&l.d_J, n * nb));
            VRT_S(dmalloc(&l.d_gamma, n));
            VRT_S(dmalloc(&l.d_strength, n));
            VRT_S(dmalloc(&l.d_R, 9 * n));
            VRT_S(dmalloc(&l.d_shares, 6 * n));
            VRT_S(dmalloc(&l.d_I0, (size_t)g->up.n1 * nb));
            VRT_S(dmalloc(&l.d_native, nb ? (size_t)vrt_plan_native_alpha_count(me.plan_all, (int64_t)nb) : 1));
            if (!rc && hipMalloc((void **)&l.d_scalars, 2 * sizeof(unsigned long long)) != hipSuccess) rc = fail(VRT_ENOMEM, "hipMalloc");
            if (!rc && hipEventCreateWithFlags(&l.ev, hipEventDisableTiming) != hipSuccess) rc = fail(VRT_ENODEVICE, "hipEventCreate");
            if (!rc && nb && (hipMemsetAsync(l.d_S_old, 0, sizeof(double) * n * nb, st) != hipSuccess ||
                              hipMemsetAsync(l.d_J, 0, sizeof(double) * n * nb, st) != hipSuccess))
                rc = fail(VRT_ENODEVICE, "hipMemsetAsync failed");
            if (!rc && nb) VRT_S(launch_gather_rows(g->up.n1, (int64_t)nb, (int64_t)nb, g->up.d_order, l.d_B0, l.d_I0, st));   // I_0 = B_λ of the bottom layer, :99-101
            if (s->native && nb) {
                const size_t np = (size_t)vrt_plan_native_plane_count(me.plan_all, (int64_t)nb);
                for (int dd = 0; dd < 2; dd++) {
                    VRT_S(dmalloc(&l.d_S_nat[dd], np));
                    VRT_S(dmalloc(&l.d_J_nat[dd], np));
                    if (!rc && hipMemsetAsync(l.d_J_nat[dd], 0, sizeof(double) * np, st) != hipSuccess) rc = fail(VRT_ENODEVICE, "hipMemsetAsync failed");
                }
                VRT_S(dmalloc(&l.d_B_up, np));
                VRT_S(planes_to_native(me.plan_all, (int64_t)nb, (int64_t)nb, l.d_B0, l.d_S_nat[0], l.d_S_nat[1], st));     // S_new = B_0
                VRT_S(planes_to_native(me.plan_all, (int64_t)nb, (int64_t)nb, l.d_B0, l.d_B_up, nullptr, st));
            }
#undef VRT_S
            if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(VRT_ENODEVICE, "uploading the line case failed");
            if (rc) { rcs[(size_t)d] = rc; errs[(size_t)d] = vrt_last_error(); }
        };
        const bool ok = run_workers(W, work);
        for (int d = 0; d < W; d++)
            if (!ok || rcs[(size_t)d]) {
                const int rc = ok ? rcs[(size_t)d] : VRT_ENOMEM;
                const std::string msg = ok ? "device " + std::to_string(mm->m[(size_t)d].device) + ": " + errs[(size_t)d] : "out of host memory";
                multi_lambda_free(s);
                return fail(rc, msg);
            }
        *out = s;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_multi_lambda_iterate(vrt_multi_lambda *s, double *max_rel_change)
{
    DeviceScope scope;
    if (!s || !max_rel_change) return fail(VRT_EINVAL, "NULL argument");
    try {
        vrt_multi *mm = s->mm;
        std::lock_guard<std::mutex> lock(mm->mu);
        const int W = (int)mm->m.size();
        const int64_t n = s->n, nlam = s->nlam;
        // ---- every device: its wavelengths' share of the iteration, enqueued on its stream ----------------------------
        auto work = [&](int d) {
            Member &me = mm->m[(size_t)d];
            LambdaMember &l = s->lm[(size_t)d];
            me.rc = VRT_OK;
            const int64_t nb = l.l1 - l.l0;
            if (hipSetDevice(me.device) != hipSuccess) { me.rc = VRT_ENODEVICE; me.err = "hipSetDevice"; return; }
            hipStream_t st = me.stream;
            vrt_plan *p = me.plan_all;
            vrt_grid *g = me.grid;
            std::lock_guard<std::mutex> plock(p->mu);
            int rc = VRT_OK;
            if (nb > 0 && s->native) {
                // the block's S and J stay in sweep order: no layout change, no copy of S (the update reads the old S where it writes the new)
                rc = launch_line_terms(n, l.d_gamma_static, l.d_gamma_unsold, l.d_pops, s->strength_const, s->Bij, s->Bji, l.d_gamma, l.d_strength, st);
                if (!rc) rc = launch_line_opacity(p, nb, l.d_small + l.l0, s->lambda0, s->c0, l.d_velocity, l.d_doppler, l.d_gamma, l.d_strength, l.d_alpha_cont, l.d_native, st);
                if (!rc) rc = execute_native_locked(p, nb, l.d_S_nat[0], l.d_S_nat[1], l.d_native, VRT_ALPHA_ANGLE_NATIVE, l.d_I0, nullptr, s->weights.data(),
                                                    l.d_J_nat[0], l.d_J_nat[1], st);
                if (!rc) rc = launch_lambda_update_native(g, nb, l.d_J_nat[0], l.d_J_nat[1], l.d_B_up, l.d_eps, l.d_S_nat[0], l.d_S_nat[1], l.d_scalars, st);
            } else if (nb > 0) {
                if (hipMemcpyAsync(l.d_S_old, l.d_S_new, sizeof(double) * (size_t)n * (size_t)nb, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    rc = fail(VRT_ENODEVICE, "hipMemcpyAsync");                                                  // S_old = copy(S_new), :258
                // γ and the line strength of the current populations (:72-75, line.jl:219-225), α_tot of every angle (:89-96)
                if (!rc) rc = launch_line_terms(n, l.d_gamma_static, l.d_gamma_unsold, l.d_pops, s->strength_const, s->Bij, s->Bji, l.d_gamma, l.d_strength, st);
                if (!rc) rc = launch_line_opacity(p, nb, l.d_small + l.l0, s->lambda0, s->c0, l.d_velocity, l.d_doppler, l.d_gamma, l.d_strength, l.d_alpha_cont, l.d_native, st);
                // J_λ of this block (:84-111)
                if (!rc) rc = execute_dev_locked(p, nb, nb, l.d_S_old, l.d_native, VRT_ALPHA_ANGLE_NATIVE, l.d_I0, nullptr, s->weights.data(), l.d_J, nullptr, st);
                // S_new = (1 - ε) J + ε B_0 and this block's share of the criterion (:261-263, :325-349)
                if (!rc) rc = launch_lambda_update(n, nb, nb, l.d_J, l.d_B0, l.d_eps, l.d_S_old, l.d_S_new, l.d_scalars, st);
            } else {
                if (hipMemsetAsync(l.d_scalars, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) rc = fail(VRT_ENODEVICE, "hipMemsetAsync");
                if (!rc) rc = launch_line_terms(n, l.d_gamma_static, l.d_gamma_unsold, l.d_pops, s->strength_const, s->Bij, s->Bji, l.d_gamma, l.d_strength, st);
            }
            // this block's share of the six rate integrals (rates.jl:154-201)
            const bool natJ = s->native && nb > 0;
            if (!rc) rc = launch_rates_partial(g, nlam, l.l0, l.l1, std::max<int64_t>(nb, 1), s->blocks, l.d_small, l.d_J, s->lambda0, s->c0, l.d_doppler,
                                               l.d_gamma, s->sigma_bb_const, l.d_temperature, l.d_lte, s->hc_over_kB, s->pref_ij, s->pref_ji, l.d_shares, st,
                                               natJ ? l.d_J_nat[0] : nullptr, natJ ? l.d_J_nat[1] : nullptr);
            if (rc) { me.rc = rc; me.err = vrt_last_error(); }
        };
        if (!run_workers(W, work)) return fail(VRT_ENOMEM, "out of host memory in a device worker");
        for (const Member &me : mm->m)
            if (me.rc) return fail(me.rc, "device " + std::to_string(me.device) + ": " + me.err);
        // ---- the shares summed over the devices: ONE all-reduce of 6 n doubles (in place), stream-ordered ---------------
        const size_t cnt = 6 * (size_t)n;
        if (!mm->comms.empty()) {
            ncclResult_t r = mm->rccl.GroupStart();
            hipError_t he = hipSuccess;
            for (int d = 0; d < W && r == ncclSuccess && he == hipSuccess; d++) {
                he = hipSetDevice(mm->m[(size_t)d].device);
                if (he == hipSuccess)
                    r = mm->rccl.AllReduce(s->lm[(size_t)d].d_shares, s->lm[(size_t)d].d_shares, cnt, ncclDouble, ncclSum, mm->comms[(size_t)d], mm->m[(size_t)d].stream);
            }
            const ncclResult_t r2 = mm->rccl.GroupEnd();
            if (he != hipSuccess) return fail(VRT_ENODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he));
            if (r != ncclSuccess || r2 != ncclSuccess)
                return fail(VRT_ENODEVICE, std::string("ncclAllReduce: ") + mm->rccl.GetErrorString(r != ncclSuccess ? r : r2));
        } else if (W > 1) {
            // handles on ONE device (rehearsal): member 0 adds the others' shares behind their streams, the others copy the sum
            Member &m0 = mm->m[0];
            VRT_HIP_TRY(hipSetDevice(m0.device));
            for (int d = 1; d < W; d++) {
                VRT_HIP_TRY(hipEventRecord(s->lm[(size_t)d].ev, mm->m[(size_t)d].stream));
                VRT_HIP_TRY(hipStreamWaitEvent(m0.stream, s->lm[(size_t)d].ev, 0));
                if (int rc = launch_axpy(cnt, s->lm[(size_t)d].d_shares, s->lm[0].d_shares, m0.stream)) return rc;
            }
            VRT_HIP_TRY(hipEventRecord(s->lm[0].ev, m0.stream));
            for (int d = 1; d < W; d++) {
                VRT_HIP_TRY(hipStreamWaitEvent(mm->m[(size_t)d].stream, s->lm[0].ev, 0));
                VRT_HIP_TRY(hipMemcpyAsync(s->lm[(size_t)d].d_shares, s->lm[0].d_shares, sizeof(double) * cnt, hipMemcpyDeviceToDevice, mm->m[(size_t)d].stream));
            }
        }
        // ---- every device: R and the populations of every site; the criterion's scalars come home -----------------------
        std::vector<unsigned long long> h((size_t)(2 * W), 0);
        for (int d = 0; d < W; d++) {
            Member &me = mm->m[(size_t)d];
            LambdaMember &l = s->lm[(size_t)d];
            VRT_HIP_TRY(hipSetDevice(me.device));
            (void)hipGetLastError();
            if (int rc = launch_populations_from_shares(me.grid, l.d_shares, l.d_C, l.d_atom, l.d_R, l.d_pops, me.stream)) return rc;
            VRT_HIP_TRY(hipMemcpyAsync(h.data() + 2 * d, l.d_scalars, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, me.stream));
        }
        double worst = 0.0;
        bool is_nan = false;
        for (int d = 0; d < W; d++) {
            VRT_HIP_TRY(hipSetDevice(mm->m[(size_t)d].device));
            VRT_HIP_TRY(hipStreamSynchronize(mm->m[(size_t)d].stream));
            if (int rcc = patch_chain_check(mm->m[(size_t)d].plan_all)) return rcc;     // this iteration's sweep gave up on that device
            double v;
            std::memcpy(&v, &h[(size_t)(2 * d)], sizeof(double));
            worst = std::max(worst, v);
            is_nan = is_nan || h[(size_t)(2 * d + 1)] != 0;
        }
        *max_rel_change = is_nan ? std::nan("") : worst;
        s->iterations++;
        return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

int vrt_multi_lambda_get(vrt_multi_lambda *s, double *J, double *S, double *populations, double *R, double *gamma)
{
    DeviceScope scope;
    if (!s) return fail(VRT_EINVAL, "NULL session");
    try {
    vrt_multi *mm = s->mm;
    std::lock_guard<std::mutex> lock(mm->mu);
    const size_t n = (size_t)s->n, nl = (size_t)s->nlam, w8 = sizeof(double);
    for (size_t d = 0; d < mm->m.size(); d++) {
        const LambdaMember &l = s->lm[d];
        const size_t nb = (size_t)(l.l1 - l.l0);
        VRT_HIP_TRY(hipSetDevice(mm->m[d].device));
        if (s->native && nb && (J || S)) {
            // the caller's layout is formed here, on request (d_J / d_S_old of the block serve as scratch)
            vrt_plan *p = mm->m[d].plan_all;
            hipStream_t st = mm->m[d].stream;
            std::lock_guard<std::mutex> plock(p->mu);
            if (J) {
                if (int rc = J_from_native(p, (int64_t)nb, (int64_t)nb, l.d_J_nat[0], l.d_J_nat[1], l.d_J, st)) return rc;
                VRT_HIP_TRY(hipMemcpy2DAsync(J + l.l0, w8 * nl, l.d_J, w8 * nb, w8 * nb, n, hipMemcpyDeviceToHost, st));
            }
            if (S) {
                if (int rc = plane_from_native(p, 0, (int64_t)nb, (int64_t)nb, l.d_S_nat[0], l.d_S_old, st)) return rc;
                VRT_HIP_TRY(hipMemcpy2DAsync(S + l.l0, w8 * nl, l.d_S_old, w8 * nb, w8 * nb, n, hipMemcpyDeviceToHost, st));
            }
            VRT_HIP_TRY(hipStreamSynchronize(st));
        } else {
        if (nb && J) VRT_HIP_TRY(hipMemcpy2D(J + l.l0, w8 * nl, l.d_J, w8 * nb, w8 * nb, n, hipMemcpyDeviceToHost));
        if (nb && S) VRT_HIP_TRY(hipMemcpy2D(S + l.l0, w8 * nl, l.d_S_new, w8 * nb, w8 * nb, n, hipMemcpyDeviceToHost));
        }
        if (d == 0) {
            if (populations) VRT_HIP_TRY(hipMemcpy(populations, l.d_pops, w8 * 3 * n, hipMemcpyDeviceToHost));
            if (R) VRT_HIP_TRY(hipMemcpy(R, l.d_R, w8 * 9 * n, hipMemcpyDeviceToHost));
            if (gamma) VRT_HIP_TRY(hipMemcpy(gamma, l.d_gamma, w8 * n, hipMemcpyDeviceToHost));
        }
    }
    return VRT_OK;
    } catch (const std::bad_alloc &) {
        return fail(VRT_ENOMEM, "out of host memory");
    } catch (...) {
        return fail(VRT_EINVAL, "unexpected exception");
    }
}

void vrt_multi_lambda_destroy(vrt_multi_lambda *s)
{
    DeviceScope scope;
    multi_lambda_free(s);
}

void vrt_multi_destroy(vrt_multi *mm)
{
    DeviceScope scope;
    multi_free(mm);
}

}  // extern "C"
