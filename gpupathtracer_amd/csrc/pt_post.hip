// Post filter of the finished frame: the reference's medianFilter3x3 / medianFilter2x2
// (src/ptMain.cu:700-788), which its -m flag applies on the host before writeImage (:684-693).
//
// The reference sorts each pixel's neighbourhood of Vector3f by squared length with qsort and
// takes element (count-1)/2.  glibc's qsort is a stable merge sort for arrays this small, so the
// element taken is the one of rank (count-1)/2 under the order (squared length, position in the
// neighbourhood array); the kernel finds that rank by counting, one lane per pixel.  The filter
// moves 24 bytes per pixel (12 read, 12 written; neighbours come from cache) and is HBM-bound.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ptcore.h"
#include "pt_host.h"

namespace {

template <int N>
__global__ __launch_bounds__(256) void k_median(const float* __restrict__ in, float* __restrict__ out, int nx, int ny)
{
    const int s = blockIdx.x * 64 + (threadIdx.x & 63);
    const int l = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (s >= nx || l >= ny) return;
    // neighbourhood in the reference's order: column by column, each top to bottom
    const int cols[3] = { max(0, s - 1), s, min(nx - 1, s + 1) };
    const int rows[3] = { max(0, l - 1), l, min(ny - 1, l + 1) };
    float r[N * N], g[N * N], b[N * N], key[N * N];
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float* px = in + 3 * ((size_t)rows[q] * nx + cols[c]);
            const int i = c * N + q;
            r[i] = px[0]; g[i] = px[1]; b[i] = px[2];
            key[i] = r[i] * r[i] + g[i] * g[i] + b[i] * b[i];          // squared_length()
        }
    }
    constexpr int target = (N * N - 1) / 2;
    float mr = r[target], mg = g[target], mb = b[target];
#pragma unroll
    for (int i = 0; i < N * N; i++) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < N * N; j++) rank += (key[j] < key[i] || (key[j] == key[i] && j < i)) ? 1 : 0;
        if (rank == target) { mr = r[i]; mg = g[i]; mb = b[i]; }
    }
    float* o = out + 3 * ((size_t)l * nx + s);
    o[0] = mr; o[1] = mg; o[2] = mb;
}

bool hip_ok(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    pt::set_error((std::string(what) + ": " + hipGetErrorString(e)).c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return false;
}

}  // namespace

extern "C" int pt_median_filter_device(const void* d_in_rgb, void* d_out_rgb, int32_t nx, int32_t ny, int32_t size, void* hip_stream)
{
    if (!d_in_rgb || !d_out_rgb) { pt::set_error("pt_median_filter_device: null argument"); return PT_ERR_INVALID; }
    if (d_in_rgb == d_out_rgb) { pt::set_error("pt_median_filter_device: the filter cannot run in place"); return PT_ERR_INVALID; }
    if (nx <= 0 || ny <= 0 || (int64_t)nx * ny > ((int64_t)1 << 30)) { pt::set_error("pt_median_filter_device: bad image size"); return PT_ERR_INVALID; }
    if (size != 2 && size != 3) { pt::set_error("pt_median_filter_device: size must be 3 (3x3) or 2 (2x2)"); return PT_ERR_INVALID; }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const dim3 block(256), grid((nx + 63) / 64, (ny + 3) / 4);
    const float* in = static_cast<const float*>(d_in_rgb);
    float* out = static_cast<float*>(d_out_rgb);
    if (size == 3) hipLaunchKernelGGL(k_median<3>, grid, block, 0, st, in, out, nx, ny);
    else hipLaunchKernelGGL(k_median<2>, grid, block, 0, st, in, out, nx, ny);
    if (!hip_ok(hipGetLastError(), "median filter launch")) return PT_ERR_HIP;
    return PT_OK;
}

extern "C" int pt_median_filter(const float* in_rgb, float* out_rgb, int32_t nx, int32_t ny, int32_t size)
{
    if (!in_rgb || !out_rgb) { pt::set_error("pt_median_filter: null argument"); return PT_ERR_INVALID; }
    if (nx <= 0 || ny <= 0 || (int64_t)nx * ny > ((int64_t)1 << 30)) { pt::set_error("pt_median_filter: bad image size"); return PT_ERR_INVALID; }
    const size_t bytes = (size_t)3 * nx * ny * sizeof(float);
    void *d_in = nullptr, *d_out = nullptr;
    if (!hip_ok(hipMalloc(&d_in, bytes), "hipMalloc(filter input)")) return PT_ERR_OOM;
    if (!hip_ok(hipMalloc(&d_out, bytes), "hipMalloc(filter output)")) { (void)hipFree(d_in); return PT_ERR_OOM; }
    int rc = PT_OK;
    if (!hip_ok(hipMemcpy(d_in, in_rgb, bytes, hipMemcpyHostToDevice), "hipMemcpy(filter input)")) rc = PT_ERR_HIP;
    if (rc == PT_OK) rc = pt_median_filter_device(d_in, d_out, nx, ny, size, nullptr);
    if (rc == PT_OK && !hip_ok(hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost), "hipMemcpy(filter output)")) rc = PT_ERR_HIP;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}
