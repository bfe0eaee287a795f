// Probe: what a cross-stream dependency costs on this stack.
//   (a) chain of tiny kernels on ONE stream                          -> us per dependent launch
//   (b) the same chain alternating between TWO streams, every hop an hipEventRecord + hipStreamWaitEvent
//   (c) two-stage software pipeline with distance-1 hand-offs: X: A(1) A(2) ...; Y: B(1) B(2) ...; B(s) waits A(s),
//       A(s + 1) waits B(s - 1); kernels of ~20 us (A) and ~20 us (B): the per-step time of the overlapped pipeline
//       against the 40 us of running A and B back to back on one stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o build_probe/cross_stream tools/probe/cross_stream.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_tiny(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_busy(unsigned *p, int iters) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeef) p[1] = x;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    unsigned *p; CK(hipMalloc(&p, 64)); CK(hipMemset(p, 0, 64));
    hipStream_t x, y; CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&y, hipStreamNonBlocking));
    const int N = 1000;
    std::vector<hipEvent_t> ev(2 * N + 2);
    for (auto &evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    // (a)
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, x, p);
    CK(hipStreamSynchronize(x));
    double t0 = now_us();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, x, p);
    CK(hipStreamSynchronize(x));
    printf("{\"one_stream_tiny_us_per_launch\": %.2f,\n", (now_us() - t0) / N);
    // (b)
    t0 = now_us();
    for (int i = 0; i < N; ++i) {
        hipStream_t s = (i & 1) ? y : x, o = (i & 1) ? x : y;
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, p);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(o, ev[i], 0));
    }
    CK(hipStreamSynchronize(x)); CK(hipStreamSynchronize(y));
    printf(" \"ping_pong_tiny_us_per_hop\": %.2f,\n", (now_us() - t0) / N);
    // calibrate a ~20 us kernel
    int iters = 2000;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, x, p, iters); CK(hipStreamSynchronize(x));
        t0 = now_us();
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, x, p, iters);
        CK(hipStreamSynchronize(x));
        const double per = (now_us() - t0) / 50;
        iters = static_cast<int>(iters * 20.0 / per);
    }
    t0 = now_us();
    for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, x, p, iters);
    CK(hipStreamSynchronize(x));
    printf(" \"serial_busy_us_per_pair\": %.2f,\n", (now_us() - t0) / N);
    // (c) pipeline: A on x (half the machine: 512 blocks), B on y (512 blocks): overlapped they fit side by side
    for (int blocks : {1024, 512}) {
        t0 = now_us();
        for (int s = 0; s < N; ++s) {
            if (s >= 2) CK(hipStreamWaitEvent(x, ev[2 * (s - 2) + 1], 0)); // A(s) waits B(s - 2)
            hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, x, p, iters);
            CK(hipEventRecord(ev[2 * s], x));
            CK(hipStreamWaitEvent(y, ev[2 * s], 0));                        // B(s) waits A(s)
            hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, y, p, iters);
            CK(hipEventRecord(ev[2 * s + 1], y));
        }
        CK(hipStreamSynchronize(x)); CK(hipStreamSynchronize(y));
        printf(" \"pipeline_us_per_step_%d_blocks\": %.2f%s\n", blocks, (now_us() - t0) / N, blocks == 512 ? "}" : ",");
    }
    return 0;
}
