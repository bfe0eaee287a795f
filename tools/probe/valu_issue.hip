// Probe: issue cost of the VALU / lane / ballot instructions of the random-stream kernel (k_rng) on gfx950.
//
// For every opcode of rng_step (device_util.hpp Rng32::next_top, kernels.hip rng_step) a kernel runs a long
// stream of INDEPENDENT instances of that instruction (16 rotating destinations) at 1, 2, 4 and 8 waves per
// SIMD, every SIMD of every CU loaded.  Reported per (op, waves/SIMD):
//   cyc_per_inst_wave  = shader cycles (s_memtime) one wave needs per instruction
//   cyc_per_inst_simd  = the same divided by the waves sharing the SIMD = issue cost of the SIMD per
//                        wave-instruction (this is what the limiter of bench.py is priced with)
//   wall_inst_per_s    = wave-instructions per second of the whole chip from hipEvent time
// plus the in-kernel clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
// The last rows run the real generator step + three compares + ballot parking ("rng_step").
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_issue tools/probe/valu_issue.hip
// Run:   ./valu_issue > profiles/r02_valu_issue.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

enum Op {
    OP_XOR,        // v_xor_b32
    OP_BITOP3,     // v_bitop3_b32 (three-input xor)
    OP_ALIGNBIT,   // v_alignbit_b32
    OP_LSHL,       // v_lshlrev_b32
    OP_ADD,        // v_add_u32
    OP_LSHL_ADD64, // v_lshl_add_u64
    OP_CMP_SGPR,   // v_cmp_lt_u32 -> SGPR pair
    OP_CMP_XOR,    // v_cmp_lt_u32 x3 + 2 s_xor_b64 (the level ballots of one draw)
    OP_WRITELANE,  // v_writelane_b32 (SGPR value, constant lane)
    OP_CMP_WRITELANE, // v_cmp -> SGPR, then v_writelane of that SGPR (the hazard the verdict suspects)
    OP_READLANE,   // v_readlane_b32
    OP_CNDMASK,    // v_cndmask_b32 with an SGPR-pair mask
    OP_ADDC,       // v_addc_co_u32 x, x, x, sgpr-pair carry in (shift a compare bit in)
    OP_FMA32,      // v_fma_f32 (calibration against the guide's constants table)
    OP_PKFMA32,    // v_pk_fma_f32
    OP_MOV_DPP,    // v_mov_b32 row_ror
    OP_SNOP,       // s_nop 0
    OP_SXOR,       // s_xor_b64
    OP_AND,        // v_and_b32
    OP_LSHR,       // v_lshrrev_b32
    OP_BFE,        // v_bfe_u32
    OP_BFI,        // v_bfi_b32
    OP_PERM,       // v_perm_b32
    OP_LSHL_OR,    // v_lshl_or_b32
    OP_LSHL_ADD32, // v_lshl_add_u32
    OP_AND_OR,     // v_and_or_b32
    OP_MUL_U24,    // v_mul_u32_u24
    OP_MAD_U24,    // v_mad_u32_u24
    OP_MUL_LO,     // v_mul_lo_u32
    OP_MUL_HI,     // v_mul_hi_u32
    OP_SUB,        // v_sub_u32
    OP_ADD3,       // v_add3_u32
    OP_XAD,        // v_xad_u32
    OP_MOV,        // v_mov_b32
    OP_MOV_SGPR,   // v_mov_b32 from an SGPR
    OP_CMP_VCC,    // v_cmp_lt_u32 -> vcc
    OP_ADD_CO,     // v_add_co_u32 (carry out to an SGPR pair)
    OP_SDWA_XOR,   // v_xor_b32_sdwa (word select = free 16-bit shift)
    OP_MIX_1_1,    // alternating v_xor / v_alignbit
    OP_MIX_4_4,    // 4 v_xor then 4 v_alignbit
    OP_MIX_8_8,    // 8 v_xor then 8 v_alignbit
    OP_MIX_3_1,    // 3 v_xor, 1 v_alignbit
    OP_MIX_FULL2,  // alternating v_xor / v_bitop3 (both in the fast class)
    OP_RNG_STEP,   // the real thing: one xoshiro step + 3 compares + 4 parkings
    OP_RNG_ONLY,   // xoshiro step only (next_top)
    OP_COUNT
};

static const char *kNames[OP_COUNT] = {
    "v_xor_b32",      "v_bitop3_b32",  "v_alignbit_b32", "v_lshlrev_b32",    "v_add_u32",
    "v_lshl_add_u64", "v_cmp_lt_u32_sgpr", "cmp3_sxor2",  "v_writelane_b32", "cmp_then_writelane",
    "v_readlane_b32", "v_cndmask_b32_sgpr", "v_addc_co_u32_sgpr", "v_fma_f32", "v_pk_fma_f32",
    "v_mov_b32_dpp",  "s_nop_0",       "s_xor_b64",
    "v_and_b32", "v_lshrrev_b32", "v_bfe_u32", "v_bfi_b32", "v_perm_b32", "v_lshl_or_b32", "v_lshl_add_u32",
    "v_and_or_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_sub_u32", "v_add3_u32",
    "v_xad_u32", "v_mov_b32", "v_mov_b32_from_sgpr", "v_cmp_lt_u32_vcc", "v_add_co_u32_sgpr", "v_xor_b32_sdwa",
    "mix_xor1_alignbit1", "mix_xor4_alignbit4", "mix_xor8_alignbit8", "mix_xor3_alignbit1", "mix_xor1_bitop3_1",
    "rng_step_full",    "xoshiro_next_top"};
// VALU (or scalar) instructions per unit of the unrolled body, for the per-instruction figures
static const int kInstPerUnit[OP_COUNT] = {1, 1, 1, 1, 1, 1, 1, 5, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1,
                                           1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                           1, 1, 1, 1, 1, 23, 16};

struct Rng32p {
    uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
    __device__ __forceinline__ uint32_t next_top() {
        uint64_t s1 = (static_cast<uint64_t>(b1) << 32) | b0, pp, rr2;
        asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(pp) : "v"(s1));
        const uint32_t plo = static_cast<uint32_t>(pp), phi = static_cast<uint32_t>(pp >> 32);
        const uint32_t rhi = __builtin_amdgcn_alignbit(phi, plo, 25);
        const uint32_t rlo = __builtin_amdgcn_alignbit(plo, phi, 25);
        const uint64_t rr = (static_cast<uint64_t>(rhi) << 32) | rlo;
        asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(rr2) : "v"(rr));
        const uint32_t top = static_cast<uint32_t>(rr2 >> 32);
        const uint32_t t0 = b0 << 17;
        const uint32_t t1 = __builtin_amdgcn_alignbit(b1, b0, 15);
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

template <int J> __device__ __forceinline__ void full_step(Rng32p &r, uint32_t c0, uint32_t c1, uint32_t c2,
                                                           uint32_t &k0, uint32_t &k1) {
    const uint32_t draw = r.next_top();
    const unsigned long long l0 = __ballot(draw < c0), l1 = __ballot(draw < c1), l2 = __ballot(draw < c2);
    const unsigned long long m0 = l0 ^ l1 ^ l2;
    k0 = park<J>(k0, static_cast<uint32_t>(m0));
    k0 = park<J + 32>(k0, static_cast<uint32_t>(m0 >> 32));
    k1 = park<J>(k1, static_cast<uint32_t>(l1));
    k1 = park<J + 32>(k1, static_cast<uint32_t>(l1 >> 32));
}

// UNITS units per loop iteration; a unit is one instruction (or the small group named above).
constexpr int UNITS = 64;

template <int OP>
__global__ __launch_bounds__(1024) void k_probe(uint32_t *out, unsigned long long *cycles, unsigned long long *real,
                                                int iters, uint32_t seed) {
    const uint32_t tid = threadIdx.x;
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed * (tid + 1) + i * 0x9E3779B9u;
    uint32_t k = seed ^ tid;
    unsigned long long s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0;
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (static_cast<uint64_t>(x[2 * i]) << 32) | x[2 * i + 1];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 1.0f + i;
    Rng32p rng{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
    uint32_t sval = __builtin_amdgcn_readfirstlane(seed);
    uint32_t k0 = 0, k1 = 0;

    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == OP_RNG_STEP) {
            // 32 real steps = 32 x 23 VALU (+ 2 SALU each)
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (full_step<J>(rng, 0x40000000u, 0x20000000u, 0x10000000u, k0, k1), ...);
            }(std::make_integer_sequence<int, 32>{});
            x[0] ^= k0 ^ k1;
        } else if constexpr (OP == OP_RNG_ONLY) {
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u & 15] ^= rng.next_top();
        } else {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                const int i = u & 15;
                if constexpr (OP == OP_XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_BITOP3)
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[i]));
                if constexpr (OP == OP_ADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_LSHL_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(w[i & 7]));
                if constexpr (OP == OP_CMP_SGPR) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(s[i]) : "v"(x[i]), "v"(k));
                if constexpr (OP == OP_CMP_XOR) {
                    unsigned long long a, b, c;
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(a) : "v"(x[i]), "v"(k));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(b) : "v"(x[(i + 1) & 15]), "v"(k));
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(c) : "v"(x[(i + 2) & 15]), "v"(k));
                    asm volatile("s_xor_b64 %0, %1, %2\n\ts_xor_b64 %0, %0, %3" : "=&s"(s[i]) : "s"(a), "s"(b), "s"(c) : "scc");
                }
                if constexpr (OP == OP_WRITELANE) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(x[i]) : "s"(sval));
                if constexpr (OP == OP_CMP_WRITELANE) {
                    unsigned long long a;
                    asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(a) : "v"(x[(i + 8) & 15]), "v"(k));
                    asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(x[i]) : "s"(static_cast<uint32_t>(a)));
                }
                if constexpr (OP == OP_READLANE) {
                    uint32_t r;
                    asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(r) : "v"(x[i]));
                    s[i] = r;
                }
                if constexpr (OP == OP_CNDMASK)
                    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "s"(s[0]));
                if constexpr (OP == OP_ADDC)
                    asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(x[i]) : "s"(s[0]) : "vcc");
                if constexpr (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 8) & 15]));
                if constexpr (OP == OP_PKFMA32) {
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i & 7]));
                }
                if constexpr (OP == OP_MOV_DPP)
                    asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_SNOP) asm volatile("s_nop 0");
                if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x[i]));
                if constexpr (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(x[i]));
                if constexpr (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 8) & 15]), "v"(k));
                if constexpr (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 13, %1" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_LSHL_ADD32) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_SUB) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 8) & 15]));
                if constexpr (OP == OP_MOV_SGPR) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "s"(sval));
                if constexpr (OP == OP_CMP_VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(k) : "vcc");
                if constexpr (OP == OP_ADD_CO) asm volatile("v_add_co_u32 %0, %1, %0, %2" : "+v"(x[i]), "=s"(s[i]) : "v"(k));
                if constexpr (OP == OP_SDWA_XOR)
                    asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x[i]) : "v"(k));
                if constexpr (OP == OP_MIX_1_1) {
                    if (u & 1) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(x[i]) : "v"(k));
                    else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                }
                if constexpr (OP == OP_MIX_4_4) {
                    if (u & 4) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(x[i]) : "v"(k));
                    else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                }
                if constexpr (OP == OP_MIX_8_8) {
                    if (u & 8) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(x[i]) : "v"(k));
                    else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                }
                if constexpr (OP == OP_MIX_3_1) {
                    if ((u & 3) == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(x[i]) : "v"(k));
                    else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                }
                if constexpr (OP == OP_MIX_FULL2) {
                    if (u & 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(k), "v"(x[(i + 8) & 15]));
                    else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(k));
                }
                if constexpr (OP == OP_SXOR) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(s[i]) : "s"(s[(i + 8) & 15]) : "scc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = k0 ^ k1 ^ rng.a0 ^ rng.b1 ^ rng.c0 ^ rng.d1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= x[i] ^ static_cast<uint32_t>(s[i]) ^ static_cast<uint32_t>(s[i] >> 32) ^ __float_as_uint(f[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= static_cast<uint32_t>(w[i]) ^ static_cast<uint32_t>(w[i] >> 32);
    out[blockIdx.x * blockDim.x + tid] = acc;
    if ((tid & 63u) == 0) {
        const uint32_t wv = (blockIdx.x * blockDim.x + tid) >> 6;
        cycles[wv] = t1 - t0;
        real[wv] = r1 - r0;
    }
}

struct Result {
    double cyc_wave, wall_us, mhz;
};

template <int OP> static int run_one(int waves_per_simd, int n_cu, int iters, Result &res) {
    // 4 SIMDs per CU; a workgroup of 256 k threads puts k waves on every SIMD (k <= 4); 8 = two such workgroups
    const int per_block = std::min(waves_per_simd, 4);
    const int blocks = n_cu * (waves_per_simd / per_block);
    const int threads = 256 * per_block;
    const size_t n_threads = static_cast<size_t>(blocks) * threads, n_waves = n_threads / 64;
    uint32_t *out;
    unsigned long long *cyc, *real;
    CHECK(hipMalloc(&out, n_threads * 4));
    CHECK(hipMalloc(&cyc, n_waves * 8));
    CHECK(hipMalloc(&real, n_waves * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, real, iters, 12345u + rep);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> hc(n_waves), hr(n_waves);
    CHECK(hipMemcpy(hc.data(), cyc, n_waves * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), real, n_waves * 8, hipMemcpyDeviceToHost));
    std::vector<double> c(hc.begin(), hc.end()), r(hr.begin(), hr.end());
    std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
    std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
    res.cyc_wave = c[c.size() / 2];
    res.wall_us = best * 1e3;
    res.mhz = r[r.size() / 2] > 0 ? c[c.size() / 2] / r[r.size() / 2] * 100.0 : 0.0;
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
    CHECK(hipFree(real));
    return 0;
}

template <int OP> static int run_op(int n_cu, bool &first) {
    const int iters = (OP == OP_RNG_STEP || OP == OP_RNG_ONLY) ? 400 : 2000;
    const double units = (OP == OP_RNG_STEP || OP == OP_RNG_ONLY) ? 32.0 : UNITS;
    const double inst_per_wave = units * kInstPerUnit[OP] * iters;
    for (int wps : {1, 2, 4, 8}) {
        Result r;
        if (run_one<OP>(wps, n_cu, iters, r)) return 1;
        const double total_inst = inst_per_wave * wps * 4.0 * n_cu;
        printf("%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"inst_per_unit\": %d, \"cyc_per_inst_wave\": %.3f, "
               "\"cyc_per_inst_simd\": %.3f, \"wall_us\": %.1f, \"wall_inst_per_s\": %.4g, \"clock_mhz\": %.0f}",
               first ? "" : ",", kNames[OP], wps, kInstPerUnit[OP], r.cyc_wave / inst_per_wave,
               r.cyc_wave / inst_per_wave / wps, r.wall_us, total_inst / (r.wall_us * 1e-6), r.mhz);
        first = false;
    }
    return 0;
}

template <int... OPS> static int run_all(int n_cu, std::integer_sequence<int, OPS...>) {
    bool first = true;
    int rc = 0;
    ((rc = rc ? rc : run_op<OPS>(n_cu, first)), ...);
    return rc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz_reported\": %d,\n \"note\": \"independent "
           "instruction streams, every SIMD of every CU loaded; cyc_per_inst_simd = shader cycles per wave-instruction "
           "per SIMD\",\n \"rows\": [",
           prop.name, prop.gcnArchName, n_cu, prop.clockRate);
    const int rc = run_all(n_cu, std::make_integer_sequence<int, OP_COUNT>{});
    printf("\n]}\n");
    return rc;
}
