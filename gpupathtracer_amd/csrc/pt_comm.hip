// Device plumbing, the RCCL frame exchange and the libm diagnostics of the C ABI (include/ptcore.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <string>

#include "pt_device.h"
#include "pt_host.h"
#include "pt_libm.h"

namespace {

int hip_err(hipError_t e, const char* what)
{
    std::string msg = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    pt::set_error(msg.c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return PT_ERR_HIP;
}
int nccl_err(ncclResult_t r, const char* what)
{
    std::string msg = std::string(what) + ": " + ncclGetErrorString(r);
    pt::set_error(msg.c_str());
    return PT_ERR_HIP;
}

__global__ void k_libm_eval(int fn, const float* __restrict__ x, size_t n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 7) { out[i] = ptm::atan2f_glibc(x[2 * i], x[2 * i + 1]); return; }
    if (fn == 10) { out[i] = ptd::quot(x[2 * i], ptd::rcp_refined(x[2 * i + 1])); return; }     // the kernels' shortened forms ...
    if (fn == 11) { out[i] = x[2 * i] / x[2 * i + 1]; return; }                                // ... beside the compiler's full ones
    const float v = x[i];
    float r;
    switch (fn) {
    case 0: r = ptm::sinf_glibc(v); break;
    case 1: r = ptm::cosf_glibc(v); break;
    case 2: r = ptm::logf_glibc(v); break;
    case 3: r = ptm::pow5f_glibc(v); break;
    case 6: r = ptm::asinf_glibc(v); break;
    case 8: r = ptd::sqrt_tame(v); break;
    case 9: r = sqrtf(v); break;
    case 12: case 13: { float sn, cs; ptm::sincosf_glibc<true>(v, &sn, &cs); r = fn == 12 ? sn : cs; } break;   // the form for 0 <= v < 120
    default: { float sn, cs; ptm::sincosf_glibc(v, &sn, &cs); r = fn == 4 ? sn : cs; } break;
    }
    out[i] = r;
}

}  // namespace

struct PtComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* d_scalar = nullptr;      // one device double for the barrier and the scalar reductions
};

extern "C" int pt_set_device(int device)
{
    hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? PT_OK : hip_err(e, "hipSetDevice");
}

extern "C" int pt_device_synchronize(void)
{
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? PT_OK : hip_err(e, "hipDeviceSynchronize");
}

extern "C" int pt_device_malloc(size_t bytes, void** out)
{
    if (!out) { pt::set_error("pt_device_malloc: null argument"); return PT_ERR_INVALID; }
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes ? bytes : 4);
    if (e != hipSuccess) { hip_err(e, "hipMalloc"); return PT_ERR_OOM; }
    return PT_OK;
}

extern "C" void pt_device_free(void* p)
{
    if (p) (void)hipFree(p);
}

extern "C" int pt_memcpy_to_host(void* dst, const void* src, size_t bytes)
{
    if (!dst || !src) { pt::set_error("pt_memcpy_to_host: null argument"); return PT_ERR_INVALID; }
    hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? PT_OK : hip_err(e, "hipMemcpy");
}

extern "C" int pt_memcpy_to_device(void* dst, const void* src, size_t bytes)
{
    if (!dst || !src) { pt::set_error("pt_memcpy_to_device: null argument"); return PT_ERR_INVALID; }
    hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? PT_OK : hip_err(e, "hipMemcpy");
}

extern "C" int pt_comm_unique_id(void* id128)
{
    if (!id128) { pt::set_error("pt_comm_unique_id: null argument"); return PT_ERR_INVALID; }
    static_assert(sizeof(ncclUniqueId) == 128, "unique id is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return nccl_err(r, "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof id);
    return PT_OK;
}

extern "C" int pt_comm_init(const void* id128, int rank, int world, PtComm** out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { pt::set_error("pt_comm_init: bad arguments"); return PT_ERR_INVALID; }
    *out = nullptr;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    PtComm* c = new PtComm();
    c->rank = rank; c->world = world;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return nccl_err(r, "ncclCommInitRank"); }
    hipError_t e = hipMalloc((void**)&c->d_scalar, 64);
    if (e != hipSuccess) { (void)ncclCommDestroy(c->comm); delete c; hip_err(e, "hipMalloc(comm scalar)"); return PT_ERR_OOM; }
    *out = c;
    return PT_OK;
}

extern "C" int pt_comm_reduce_frame(PtComm* c, void* d_frame, size_t n_floats, int root, void* hip_stream)
{
    if (!c || !d_frame || root < 0 || root >= c->world) { pt::set_error("pt_comm_reduce_frame: bad arguments"); return PT_ERR_INVALID; }
    ncclResult_t r = ncclReduce(d_frame, d_frame, n_floats, ncclFloat, ncclSum, root, c->comm, static_cast<hipStream_t>(hip_stream));
    if (r != ncclSuccess) return nccl_err(r, "ncclReduce");
    return PT_OK;
}

// max over the ranks of one double, on every rank; blocks until the result is back on the host.
extern "C" int pt_comm_allreduce_max_f64(PtComm* c, double* inout, void* hip_stream)
{
    if (!c || !inout) { pt::set_error("pt_comm_allreduce_max_f64: bad arguments"); return PT_ERR_INVALID; }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipError_t e = hipMemcpyAsync(c->d_scalar, inout, sizeof(double), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return hip_err(e, "hipMemcpyAsync(scalar)");
    ncclResult_t r = ncclAllReduce(c->d_scalar, c->d_scalar, 1, ncclDouble, ncclMax, c->comm, st);
    if (r != ncclSuccess) return nccl_err(r, "ncclAllReduce(max)");
    e = hipMemcpyAsync(inout, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return hip_err(e, "hipMemcpyAsync(scalar)");
    e = hipStreamSynchronize(st);
    return e == hipSuccess ? PT_OK : hip_err(e, "hipStreamSynchronize");
}

// Returns on every rank only after every rank has called it (and after the work queued before it
// on hip_stream has finished): an all-reduce of one value followed by a stream synchronize.
extern "C" int pt_comm_barrier(PtComm* c, void* hip_stream)
{
    double one = 1.0;
    return pt_comm_allreduce_max_f64(c, &one, hip_stream);
}

extern "C" void pt_comm_destroy(PtComm* c)
{
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    delete c;
}

extern "C" int pt_libm_eval(int fn, const float* x, size_t n, float* out, int on_device)
{
    if (fn < 0 || fn > 13 || (n && (!x || !out))) { pt::set_error("pt_libm_eval: bad arguments"); return PT_ERR_INVALID; }
    if (n == 0) return PT_OK;
    if (!on_device) {
        for (size_t i = 0; i < n; i++) {
            switch (fn) {
            case 0: out[i] = ptm::sinf_glibc(x[i]); break;
            case 1: out[i] = ptm::cosf_glibc(x[i]); break;
            case 2: out[i] = ptm::logf_glibc(x[i]); break;
            case 3: out[i] = ptm::pow5f_glibc(x[i]); break;
            case 6: out[i] = ptm::asinf_glibc(x[i]); break;
            case 7: out[i] = ptm::atan2f_glibc(x[2 * i], x[2 * i + 1]); break;
            case 8: case 9: out[i] = std::sqrt(x[i]); break;
            case 10: case 11: out[i] = x[2 * i] / x[2 * i + 1]; break;
            case 12: case 13: { float sn, cs; ptm::sincosf_glibc<true>(x[i], &sn, &cs); out[i] = fn == 12 ? sn : cs; } break;
            default: { float sn, cs; ptm::sincosf_glibc(x[i], &sn, &cs); out[i] = fn == 4 ? sn : cs; } break;
            }
        }
        return PT_OK;
    }
    float *dx = nullptr, *dy = nullptr;
    hipError_t e;
    const size_t n_in = (fn == 7 || fn == 10 || fn == 11) ? 2 * n : n;
    if ((e = hipMalloc((void**)&dx, n_in * 4)) != hipSuccess) return hip_err(e, "hipMalloc");
    if ((e = hipMalloc((void**)&dy, n * 4)) != hipSuccess) { (void)hipFree(dx); return hip_err(e, "hipMalloc"); }
    int rc = PT_OK;
    if ((e = hipMemcpy(dx, x, n_in * 4, hipMemcpyHostToDevice)) != hipSuccess) rc = hip_err(e, "hipMemcpy");
    if (rc == PT_OK) {
        hipLaunchKernelGGL(k_libm_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, fn, dx, n, dy);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) rc = hip_err(e, "libm kernel");
    }
    if (rc == PT_OK && (e = hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_err(e, "hipMemcpy");
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc;
}
