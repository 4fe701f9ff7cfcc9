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
    double *dS = nullptr, *dA = nullptr, *dU = nullptr, *dD = nullptr, *dJ = nullptr;
    size_t cS = 0, cA = 0, cU = 0, cD = 0, cJ = 0;
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
        for (double *q : {me.dS, me.dA, me.dU, me.dD, me.dJ})
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
        if (mm->distinct && n_devices > 1) {
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

void vrt_multi_destroy(vrt_multi *mm)
{
    DeviceScope scope;
    multi_free(mm);
}

}  // extern "C"
