// Attainable HBM bandwidth beside the 8 TB/s specification (SURVEY.md 8d "measure the attainable figure
// with a device copy microbenchmark in the same run"): a grid-stride float4 copy, HIP-event timed.
#include "../../include/peapods_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>

namespace {

__global__ __launch_bounds__(256) void k_copy4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    // eight independent 16-byte loads in flight per thread before the first store (measured on MI355X: 5.3 TB/s
    // with 65536 workgroups; four loads or 1024 workgroups: 4.6 TB/s; hipMemcpy device-to-device: 4.9-5.1 TB/s)
    for (; i + 7 * stride < n; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

} // namespace

extern "C" int pa_device_copy_bandwidth(int32_t device, uint64_t bytes, int32_t iterations, double *gbytes_per_s_out,
                                        char *err, size_t err_len) {
    auto fail = [&](const std::string &msg) {
        if (err && err_len) std::snprintf(err, err_len, "%s", msg.c_str());
        return PA_ERR_DEVICE;
    };
    int n_devices = 0;
    if (hipGetDeviceCount(&n_devices) != hipSuccess || device < 0 || device >= n_devices)
        return fail("HIP device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    const size_t n = std::max<uint64_t>(bytes / sizeof(float4), 1024);
    float4 *src = nullptr, *dst = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&src), n * sizeof(float4)) != hipSuccess) return fail("hipMalloc failed");
    if (hipMalloc(reinterpret_cast<void **>(&dst), n * sizeof(float4)) != hipSuccess) {
        (void)hipFree(src);
        return fail("hipMalloc failed");
    }
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMemset(src, 0x3c, n * sizeof(float4));
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    float best_ms = 1e30f;
    const unsigned blocks = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 256u * std::max(prop.multiProcessorCount, 1)));
    for (int it = 0; e == hipSuccess && it < std::max(iterations, 1) + 1; ++it) { // the first launch warms up
        e = hipEventRecord(a, nullptr);
        hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, nullptr, src, dst, n);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(b, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms > 0.f) best_ms = std::min(best_ms, ms);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (e != hipSuccess) return fail(std::string("copy bandwidth probe: ") + hipGetErrorString(e));
    *gbytes_per_s_out = 2.0 * static_cast<double>(n * sizeof(float4)) / (best_ms * 1e-3) / 1e9;
    return PA_OK;
}
