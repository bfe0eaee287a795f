// Probe: whole-step variants of the random-stream inner loop (kernels.hip rng_step) on gfx950.
//
// The per-opcode probe (valu_issue.hip) prices instructions in isolation, and its own rows show that the prices do
// not add up in a mix (v_cmp + v_writelane together cost little more than either alone).  This probe therefore times
// the thing itself: one launch shaped like the random-stream blocks of k_rng_relay (1536 workgroups x 4 waves, every
// wave 128 xoshiro256** steps with three threshold compares per draw and the two level planes of 2 x 32 slots
// filed per site), in several formulations that produce the SAME bytes (checked against variant 0 on the device):
//
//   park:  0  v_writelane_b32 x 4 per step, one 8-byte store per lane per 32 steps          (the round-2 kernel)
//          1  EXEC window of one lane (s_lshl_b64 exec, 1, J), v_mov_b32 x 4 from the ballot SGPRs,
//             two 8-byte stores per lane per 64 steps
//          2  the same window with v_mov_b64 x 2
//   lazy:  the third compare (draw < c2, the rarest level) only when some lane passed the second one (scalar branch)
//   t64:   s1 << 17 as one v_lshlrev_b64 instead of v_lshlrev_b32 + v_alignbit_b32
//
// Output: one JSON row per (variant, LDS reservation): us per launch (best of 5), wave-steps per second.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_probe/step_variants tools/probe/step_variants.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

template <int SH> __device__ __forceinline__ uint64_t lshl_add_u64(uint64_t x, uint64_t y) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SH), "v"(y));
    return r;
}

template <bool T64> struct Rng32 {
    uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
    __device__ __forceinline__ uint32_t next_top() {
        const uint64_t s1 = (static_cast<uint64_t>(b1) << 32) | b0;
        const uint64_t pp = lshl_add_u64<2>(s1, s1);
        const uint32_t plo = static_cast<uint32_t>(pp), phi = static_cast<uint32_t>(pp >> 32);
        const uint32_t rhi = __builtin_amdgcn_alignbit(phi, plo, 25);
        const uint32_t rlo = __builtin_amdgcn_alignbit(plo, phi, 25);
        const uint64_t rr = (static_cast<uint64_t>(rhi) << 32) | rlo;
        const uint32_t top = static_cast<uint32_t>(lshl_add_u64<3>(rr, rr) >> 32);
        uint32_t t0, t1;
        if (T64) {
            uint64_t t;
            asm("v_lshlrev_b64 %0, 17, %1" : "=v"(t) : "v"(s1));
            t0 = static_cast<uint32_t>(t);
            t1 = static_cast<uint32_t>(t >> 32);
        } else {
            t0 = b0 << 17;
            t1 = __builtin_amdgcn_alignbit(b1, b0, 15);
        }
        const uint32_t e0 = d0 ^ b0, e1 = d1 ^ b1;
        const uint32_t nb0 = __builtin_amdgcn_bitop3_b32(b0, c0, a0, 0x96);
        const uint32_t nb1 = __builtin_amdgcn_bitop3_b32(b1, c1, a1, 0x96);
        const uint32_t nc0 = __builtin_amdgcn_bitop3_b32(c0, a0, t0, 0x96);
        const uint32_t nc1 = __builtin_amdgcn_bitop3_b32(c1, a1, t1, 0x96);
        a0 ^= e0; a1 ^= e1;
        b0 = nb0; b1 = nb1;
        c0 = nc0; c1 = nc1;
        d0 = __builtin_amdgcn_alignbit(e1, e0, 19);
        d1 = __builtin_amdgcn_alignbit(e0, e1, 19);
        return top;
    }
};

template <int LANE> __device__ __forceinline__ uint32_t park(uint32_t vec, uint32_t value) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(value), "n"(LANE));
    return vec;
}

// level ballots of one draw: m0 = bit 0 of the level, l1 = bit 1 (thresholds fall: c0 >= c1 >= c2)
template <bool LAZY>
__device__ __forceinline__ void level_ballots(uint32_t draw, uint32_t c0, uint32_t c1, uint32_t c2,
                                              unsigned long long &m0, unsigned long long &l1) {
    const unsigned long long l0 = __ballot(draw < c0);
    l1 = __ballot(draw < c1);
    m0 = l0 ^ l1;
    if (LAZY) {
        if (l1 != 0) m0 ^= __ballot(draw < c2); // draw < c2 implies draw < c1
    } else {
        m0 ^= __ballot(draw < c2);
    }
}

// PARK 0: chunk of 32 steps, writelane parking
template <int J, bool LAZY, bool T64>
__device__ __forceinline__ void step_writelane(Rng32<T64> &r, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t &k0,
                                               uint32_t &k1) {
    const uint32_t draw = r.next_top();
    unsigned long long m0, l1;
    level_ballots<LAZY>(draw, c0, c1, c2, m0, l1);
    k0 = park<J>(k0, static_cast<uint32_t>(m0));
    k0 = park<J + 32>(k0, static_cast<uint32_t>(m0 >> 32));
    k1 = park<J>(k1, static_cast<uint32_t>(l1));
    k1 = park<J + 32>(k1, static_cast<uint32_t>(l1 >> 32));
}

// PARK 1 / 2: lane J of four registers receives the four ballot words inside a one-lane EXEC window
template <int J, int PARK, bool LAZY, bool T64>
__device__ __forceinline__ void step_window(Rng32<T64> &r, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t &kA,
                                            uint32_t &kB, uint32_t &kC, uint32_t &kD) {
    const uint32_t draw = r.next_top();
    unsigned long long m0, l1;
    level_ballots<LAZY>(draw, c0, c1, c2, m0, l1);
    if (PARK == 1) {
        asm volatile("s_lshl_b64 exec, 1, %8\n\t"
                     "v_mov_b32 %0, %4\n\t"
                     "v_mov_b32 %1, %5\n\t"
                     "v_mov_b32 %2, %6\n\t"
                     "v_mov_b32 %3, %7\n\t"
                     "s_mov_b64 exec, -1"
                     : "+v"(kA), "+v"(kB), "+v"(kC), "+v"(kD)
                     : "s"(static_cast<uint32_t>(m0)), "s"(static_cast<uint32_t>(m0 >> 32)),
                       "s"(static_cast<uint32_t>(l1)), "s"(static_cast<uint32_t>(l1 >> 32)), "n"(J)
                     : "scc");
    } else {
        uint64_t ab = (static_cast<uint64_t>(kB) << 32) | kA, cd = (static_cast<uint64_t>(kD) << 32) | kC;
        asm volatile("s_lshl_b64 exec, 1, %4\n\t"
                     "v_mov_b64 %0, %2\n\t"
                     "v_mov_b64 %1, %3\n\t"
                     "s_mov_b64 exec, -1"
                     : "+v"(ab), "+v"(cd)
                     : "s"(m0), "s"(l1), "n"(J)
                     : "scc");
        kA = static_cast<uint32_t>(ab); kB = static_cast<uint32_t>(ab >> 32);
        kC = static_cast<uint32_t>(cd); kD = static_cast<uint32_t>(cd >> 32);
    }
}

constexpr int kSteps = 128; // per wave and launch, like a 128-site block

template <int PARK, bool LAZY, bool T64>
__global__ __launch_bounds__(256) void k_steps(uint2 *out, uint32_t n_waves) {
    extern __shared__ uint32_t lds_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = blockIdx.x * 4 + (tid >> 6);
    if (tid == 0xFFFFFFFFu) lds_dyn[0] = 0; // (never true: keeps the reservation referenced)
    // state: SplitMix-like hash of the global lane; thresholds of temperature slot (lane & 31) of linspace(0.8, 1.4, 24)
    Rng32<T64> r;
    {
        uint64_t z = (static_cast<uint64_t>(wave) * 64 + lane + 1) * 0x9E3779B97F4A7C15ull;
        uint32_t *st = &r.a0;
        for (int j = 0; j < 8; ++j) {
            z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
            st[j] = static_cast<uint32_t>(z >> 16) | 1u;
            z += 0x9E3779B97F4A7C15ull;
        }
    }
    const float temp = 0.8f + 0.6f * static_cast<float>((lane & 31u) % 24u) / 23.0f;
    // thresholds on the high word (PRESHIFTED form): draw24 < c  <=>  top < c << 8
    const uint32_t c0 = static_cast<uint32_t>(ceilf(expf(-4.0f / temp) * 16777216.0f)) << 8;
    const uint32_t c1 = static_cast<uint32_t>(ceilf(expf(-8.0f / temp) * 16777216.0f)) << 8;
    const uint32_t c2 = static_cast<uint32_t>(ceilf(expf(-12.0f / temp) * 16777216.0f)) << 8;
    uint2 *w0 = out + (static_cast<size_t>(wave) * 2 + 0) * kSteps;
    uint2 *w1 = out + (static_cast<size_t>(wave) * 2 + 1) * kSteps;
    if (PARK == 0) {
        const uint32_t bit = lane & 31u, half = lane >> 5;
        uint2 *dst = half ? w1 : w0;
        for (int j0 = 0; j0 < kSteps; j0 += 32) {
            uint32_t k0 = 0, k1 = 0;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (step_writelane<J, LAZY, T64>(r, c0, c1, c2, k0, k1), ...);
            }(std::make_integer_sequence<int, 32>{});
            dst[j0 + bit] = make_uint2(k0, k1);
        }
    } else {
        for (int j0 = 0; j0 < kSteps; j0 += 64) {
            uint32_t kA = 0, kB = 0, kC = 0, kD = 0;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (step_window<J, PARK, LAZY, T64>(r, c0, c1, c2, kA, kB, kC, kD), ...);
            }(std::make_integer_sequence<int, 64>{});
            w0[j0 + lane] = make_uint2(kA, kC);
            w1[j0 + lane] = make_uint2(kB, kD);
        }
    }
    if (wave == n_waves + 1) out[0] = make_uint2(r.a0 ^ r.b1, r.c0 ^ r.d1); // keep the state live
}

struct Row {
    const char *name;
    float us;
    bool same;
};

template <int PARK, bool LAZY, bool T64>
static int run(const char *name, size_t lds, uint2 *out, const std::vector<uint2> &ref, std::vector<uint2> &host,
               uint32_t n_blocks, bool &first) {
    const uint32_t n_waves = n_blocks * 4;
    const size_t n = static_cast<size_t>(n_waves) * 2 * kSteps;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_steps<PARK, LAZY, T64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipMemset(out, 0, n * sizeof(uint2)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_steps<PARK, LAZY, T64>), dim3(n_blocks), dim3(256), lds, 0, out, n_waves);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0) best = std::min(best, ms);
    }
    CHECK(hipMemcpy(host.data(), out, n * sizeof(uint2), hipMemcpyDeviceToHost));
    bool same = true;
    if (!ref.empty())
        for (size_t i = 0; i < n && same; ++i) same = host[i].x == ref[i].x && host[i].y == ref[i].y;
    printf("%s\n  {\"variant\": \"%s\", \"lds_bytes\": %zu, \"blocks\": %u, \"us_per_launch\": %.2f, "
           "\"wave_steps_per_s\": %.4g, \"same_bytes_as_variant0\": %s}",
           first ? "" : ",", name, lds, n_blocks, best * 1e3, static_cast<double>(n_waves) * kSteps / (best * 1e-3),
           ref.empty() ? "null" : (same ? "true" : "false"));
    first = false;
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t n_blocks = 1536; // the random-stream workgroups of the bench ensemble
    const size_t n = static_cast<size_t>(n_blocks) * 4 * 2 * kSteps;
    uint2 *out;
    CHECK(hipMalloc(&out, n * sizeof(uint2)));
    std::vector<uint2> ref, host(n), none;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"steps_per_wave\": %d,\n \"rows\": [", prop.name,
           prop.gcnArchName, prop.multiProcessorCount, kSteps);
    bool first = true;
    // reference bytes
    if (run<0, false, false>("writelane", 0, out, none, host, n_blocks, first)) return 1;
    ref = host;
    for (size_t lds : {static_cast<size_t>(0), static_cast<size_t>(33 * 1024)}) {
        if (run<0, false, false>("writelane", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<0, true, false>("writelane+lazy", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<0, false, true>("writelane+t64", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<1, false, false>("window_mov32", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<1, true, false>("window_mov32+lazy", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<1, true, true>("window_mov32+lazy+t64", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<2, false, false>("window_mov64", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<2, true, false>("window_mov64+lazy", lds, out, ref, host, n_blocks, first)) return 1;
        if (run<2, true, true>("window_mov64+lazy+t64", lds, out, ref, host, n_blocks, first)) return 1;
    }
    printf("\n]}\n");
    CHECK(hipFree(out));
    return 0;
}
