// Probe: do scalar stores (s_store_dwordx4 + s_dcache_wb) work on this GPU, and are the results
// visible to a following kernel?  Build: hipcc --offload-arch=gfx950 -O3 sstore.hip -o sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_write(uint4 *out, unsigned n_iter) {
    const unsigned wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    for (unsigned i = 0; i < n_iter; ++i) {
        const unsigned long long m0 = __ballot((threadIdx.x + i) & 1);
        const unsigned long long m1 = __ballot((threadIdx.x * 7 + i) & 4);
        const unsigned long long addr = (unsigned long long)(out + (size_t)wave * n_iter + i);
        const unsigned long long dst = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(addr >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((unsigned)addr);
        const unsigned a = (unsigned)m0, b = (unsigned)(m0 >> 32), c = (unsigned)m1, d = (unsigned)(m1 >> 32);
        asm volatile("s_store_dword %1, %0, 0x0\n\ts_store_dword %2, %0, 0x4\n\t"
                     "s_store_dword %3, %0, 0x8\n\ts_store_dword %4, %0, 0xc"
                     :: "s"(dst), "s"(a), "s"(b), "s"(c), "s"(d) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ void k_check(const uint4 *in, unsigned n_waves, unsigned n_iter, unsigned *bad) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_waves * n_iter) return;
    const unsigned i = idx % n_iter;
    unsigned long long e0 = 0, e1 = 0;
    for (unsigned l = 0; l < 64; ++l) {
        // every wave in k_write starts at a multiple of 64 within its block, so lane == threadIdx & 63
        if ((l + i) & 1) e0 |= 1ull << l;
        if ((l * 7 + i) & 4) e1 |= 1ull << l;
    }
    const uint4 v = in[idx];
    if (v.x != (unsigned)e0 || v.y != (unsigned)(e0 >> 32) || v.z != (unsigned)e1 || v.w != (unsigned)(e1 >> 32))
        atomicAdd(bad, 1u);
}

int main() {
    const unsigned n_blocks = 512, n_iter = 256, waves_per_block = 4, n_waves = n_blocks * waves_per_block;
    uint4 *buf;
    unsigned *bad;
    hipMalloc(&buf, sizeof(uint4) * n_waves * n_iter);
    hipMalloc(&bad, 4);
    hipMemset(buf, 0xFF, sizeof(uint4) * n_waves * n_iter);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_write, dim3(n_blocks), dim3(256), 0, 0, buf, n_iter);
    hipLaunchKernelGGL(k_check, dim3((n_waves * n_iter + 255) / 256), dim3(256), 0, 0, buf, n_waves, n_iter, bad);
    unsigned h = 12345;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    hipError_t e = hipDeviceSynchronize();
    printf("scalar-store probe: mismatches=%u status=%s\n", h, hipGetErrorString(e));
    return h == 0 && e == hipSuccess ? 0 : 1;
}
