// Probe: cost of a dependent f32 addition chain fed from LDS (the consumer wave of k_energy_f).
// Build: hipcc --offload-arch=gfx950 -O3 -o addchain tools/probe/addchain.hip ; run on a GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chain(float *out, int n_quads, int reps) {
    __shared__ float4 buf[96][32 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 96 * 64; i += 64 * WAVES) reinterpret_cast<float4 *>(buf)[i] = make_float4(1e-3f * i, 1.f, -1.f, 2e-3f);
    __syncthreads();
    float total = 0.f;
    if (tid < 64) {
        for (int r = 0; r < reps; ++r) {
            for (int q = 0; q < n_quads; q += 8) {
                float4 v[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = buf[(q + m) % 96][lane];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    total = __fadd_rn(total, v[m].x);
                    total = __fadd_rn(total, v[m].y);
                    total = __fadd_rn(total, v[m].z);
                    total = __fadd_rn(total, v[m].w);
                }
            }
        }
    } else { // other waves spin on cheap VALU work, as producers sharing the SIMDs would
        float x = tid;
        for (int r = 0; r < reps * n_quads / 2; ++r) x = x * 1.0001f + 0.5f;
        total = x;
    }
    out[blockIdx.x * 64 * WAVES + tid] = total;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int n_quads = 48, reps = 64; // 64 chunks of 192 additions
    for (int waves : {1, 8}) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(a);
            if (waves == 1)
                hipLaunchKernelGGL(chain<1>, dim3(256), dim3(64), 0, 0, out, n_quads, reps);
            else
                hipLaunchKernelGGL(chain<8>, dim3(256), dim3(512), 0, 0, out, n_quads, reps);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("waves %d: %.1f us for %d additions -> %.2f ns per addition\n", waves, ms * 1e3, n_quads * 4 * reps,
                   ms * 1e6 / (n_quads * 4 * reps));
        }
    }
    return 0;
}
