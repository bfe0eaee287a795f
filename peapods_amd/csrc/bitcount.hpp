// Bit-sliced counting shared by the on-chip kernels (kernels.hip) and the general-coupling ones
// (kernels_general.hip): majority / full adders on 32-slot words, byte-sliced vertical counters, the DPP wave sum
// and the workgroup reduction of the counters.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pa {

__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) {
    return (a & b) | (c & (a ^ b));
}

// Byte-sliced vertical counters: c[j] holds four byte counters for bits j, j+8, j+16, j+24.
__device__ __forceinline__ void byte_count(uint32_t c[8], uint32_t x) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] += (x >> j) & 0x01010101u;
}

// Per-bit count of 16 words at once.  A carry-save adder tree (11 full + 4 half adders, two
// v_bitop3 each) compresses the 16 one-bit inputs of every bit position into a 5-bit number held
// in five planes; the planes are then spread into the byte-sliced counters.  150 instructions per
// 16 words instead of 16 x 24 for word-by-word byte counting.
__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum,
                                         uint32_t &carry) {
    sum = a ^ b ^ c;
    carry = maj3(a, b, c);
}

__device__ __forceinline__ void count16(const uint32_t w[16], uint32_t c[8]) {
    uint32_t s0, s1, s2, s3, s4, c0, c1, c2, c3, c4;
    full_add(w[0], w[1], w[2], s0, c0);
    full_add(w[3], w[4], w[5], s1, c1);
    full_add(w[6], w[7], w[8], s2, c2);
    full_add(w[9], w[10], w[11], s3, c3);
    full_add(w[12], w[13], w[14], s4, c4);
    uint32_t t0, t1, d0, d1;
    full_add(s0, s1, s2, t0, d0);
    full_add(s3, s4, w[15], t1, d1);
    const uint32_t p0 = t0 ^ t1, d2 = t0 & t1;                  // weight 1
    uint32_t u0, u1, u2, e0, e1, e2;
    full_add(c0, c1, c2, u0, e0);
    full_add(c3, c4, d0, u1, e1);
    full_add(d1, d2, u0, u2, e2);
    const uint32_t p1 = u1 ^ u2, e3 = u1 & u2;                  // weight 2
    uint32_t v0, f0;
    full_add(e0, e1, e2, v0, f0);
    const uint32_t p2 = v0 ^ e3, f1 = v0 & e3;                  // weight 4
    const uint32_t p3 = f0 ^ f1, p4 = f0 & f1;                  // weights 8, 16
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t acc = c[j] + ((p0 >> j) & 0x01010101u);
        acc += ((p1 >> j) & 0x01010101u) << 1;
        acc += ((p2 >> j) & 0x01010101u) << 2;
        acc += ((p3 >> j) & 0x01010101u) << 3;
        acc += ((p4 >> j) & 0x01010101u) << 4;
        c[j] = acc;
    }
}

// Wave-wide sum by DPP only (plain VALU instructions; unlike __shfl_xor / ds_bpermute they do not
// touch the LDS pipe, which the 48-register reductions below would otherwise saturate):
// row_ror 8,4,2,1 leave every lane with its 16-lane row sum, row_bcast15 adds row r into row r+1
// for rows 1 and 3, row_bcast31 adds lane 31 into rows 2-3: lane 63 ends with the wave total.
__device__ __forceinline__ uint32_t wave_sum_lane63(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xF, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xF, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x122, 0xF, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x121, 0xF, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xA, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xC, 0xF, false));
    return v;
}

// Reduce NSET sets of byte counters over the workgroup (blockDim.x / 64 waves).  Thread b < 32
// receives in out[set] the total for bit b.  scratch: [n_waves + 1][NSET*16] dwords of LDS.
template <int NSET, uint32_t N_WAVES = 4>
__device__ __forceinline__ void block_bit_totals(const uint32_t (*c)[8], uint32_t *scratch,
                                                 uint32_t out[NSET]) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t n_waves = N_WAVES; // a 256-thread role (possibly the first four waves of a wider block), or one wave
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // widen the four byte counters to 16 bits (two registers) before summing lanes
            const uint32_t lo = wave_sum_lane63(c[s][j] & 0x00FF00FFu);        // bits j, j+16
            const uint32_t hi = wave_sum_lane63((c[s][j] >> 8) & 0x00FF00FFu); // bits j+8, j+24
            if (lane == 63) {
                scratch[wave * (NSET * 16) + s * 16 + j * 2 + 0] = lo;
                scratch[wave * (NSET * 16) + s * 16 + j * 2 + 1] = hi;
            }
        }
    }
    __syncthreads();
    uint32_t *total = scratch + n_waves * (NSET * 16);
    if (tid < NSET * 16) { // 16-bit fields cannot overflow: a total is at most the site count
        uint32_t sum = 0;
        for (uint32_t wv = 0; wv < n_waves; ++wv) sum += scratch[wv * (NSET * 16) + tid];
        total[tid] = sum;
    }
    __syncthreads();
    if (tid < 32) {
        const uint32_t j = tid & 7u, byte = tid >> 3;
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            const uint32_t v = total[s * 16 + j * 2 + (byte & 1u)];
            out[s] = (byte >> 1) ? (v >> 16) : (v & 0xFFFFu);
        }
    }
    __syncthreads();
}

} // namespace pa
