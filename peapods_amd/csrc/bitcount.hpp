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

// ---------------------------------------------------------------------------
// The same counts with the spreading shared by the four lanes of a quad.  Spreading a lane's planes into byte counters
// (above) and summing the counters over the wave cost three times what the adder tree costs, and every lane pays
// them for its own few sites.  Here the quad first adds its four bit-sliced numbers -- two DPP exchanges, a full
// adder per plane -- so that its lanes hold the same planes, and each lane spreads only the two bit-position classes
// j = 2 r, 2 r + 1 (r = lane & 3) of the eight; the sum over the wave then runs over lanes of equal r, two counters
// per lane instead of eight (planes16 + quad_add_planes + spread_pair: ~115 instructions per 16 words against 150, and
// the reduction behind it 12 against 128).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quad_swap1(uint32_t v) { // the value of lane ^ 1
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
}
__device__ __forceinline__ uint32_t quad_swap2(uint32_t v) { // the value of lane ^ 2
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
}

// the adder tree of count16 alone: 16 one-bit words -> the five planes of their per-bit count
__device__ __forceinline__ void planes16(const uint32_t w[16], uint32_t (&p)[5]) {
    uint32_t s0, s1, s2, s3, s4, c0, c1, c2, c3, c4;
    full_add(w[0], w[1], w[2], s0, c0);
    full_add(w[3], w[4], w[5], s1, c1);
    full_add(w[6], w[7], w[8], s2, c2);
    full_add(w[9], w[10], w[11], s3, c3);
    full_add(w[12], w[13], w[14], s4, c4);
    uint32_t t0, t1, d0, d1;
    full_add(s0, s1, s2, t0, d0);
    full_add(s3, s4, w[15], t1, d1);
    p[0] = t0 ^ t1;
    const uint32_t d2 = t0 & t1;
    uint32_t u0, u1, u2, e0, e1, e2;
    full_add(c0, c1, c2, u0, e0);
    full_add(c3, c4, d0, u1, e1);
    full_add(d1, d2, u0, u2, e2);
    p[1] = u1 ^ u2;
    const uint32_t e3 = u1 & u2;
    uint32_t v0, f0;
    full_add(e0, e1, e2, v0, f0);
    p[2] = v0 ^ e3;
    const uint32_t f1 = v0 & e3;
    p[3] = f0 ^ f1;
    p[4] = f0 & f1;
}

// ... of 8 words: four planes
__device__ __forceinline__ void planes8(const uint32_t w[8], uint32_t (&p)[4]) {
    uint32_t s0, s1, c0, c1, t0, d0;
    full_add(w[0], w[1], w[2], s0, c0);
    full_add(w[3], w[4], w[5], s1, c1);
    full_add(s0, s1, w[6], t0, d0);
    p[0] = t0 ^ w[7];
    const uint32_t d1 = t0 & w[7];
    uint32_t u0, e0;
    full_add(c0, c1, d0, u0, e0);
    p[1] = u0 ^ d1;
    const uint32_t e1 = u0 & d1;
    p[2] = e0 ^ e1;
    p[3] = e0 & e1;
}

// ... of 4 words: three planes
__device__ __forceinline__ void planes4(const uint32_t w[4], uint32_t (&p)[3]) {
    uint32_t s, c;
    full_add(w[0], w[1], w[2], s, c);
    p[0] = s ^ w[3];
    const uint32_t d = s & w[3];
    p[1] = c ^ d;
    p[2] = c & d;
}

// bit-sliced sum of the four lanes of a quad: NP planes in, NP + 2 out, alike in the four lanes
template <int NP>
__device__ __forceinline__ void quad_add_planes(const uint32_t (&p)[NP], uint32_t (&o)[NP + 2]) {
    uint32_t t[NP + 1], c;
    {
        const uint32_t b = quad_swap1(p[0]);
        t[0] = p[0] ^ b;
        c = p[0] & b;
    }
#pragma unroll
    for (int k = 1; k < NP; ++k) {
        const uint32_t b = quad_swap1(p[k]);
        t[k] = p[k] ^ b ^ c;
        c = maj3(p[k], b, c);
    }
    t[NP] = c;
    {
        const uint32_t b = quad_swap2(t[0]);
        o[0] = t[0] ^ b;
        c = t[0] & b;
    }
#pragma unroll
    for (int k = 1; k <= NP; ++k) {
        const uint32_t b = quad_swap2(t[k]);
        o[k] = t[k] ^ b ^ c;
        c = maj3(t[k], b, c);
    }
    o[NP + 1] = c;
}

// byte counters of the bit positions 2 r + jj + 8 m (byte m of out[jj]) from planes every lane of the quad holds
template <int NP>
__device__ __forceinline__ void spread_pair(const uint32_t (&P)[NP], uint32_t r, uint32_t (&out)[2]) {
#pragma unroll
    for (uint32_t jj = 0; jj < 2; ++jj) {
        const uint32_t sh = 2 * r + jj;
        uint32_t acc = (P[0] >> sh) & 0x01010101u;
#pragma unroll
        for (int k = 1; k < NP; ++k) acc += ((P[k] >> sh) & 0x01010101u) << k;
        out[jj] = acc;
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

// The reduction behind spread_pair for a 256-thread role: c2[set][jj] = a lane's two byte counters (<= 64 per byte:
// 16 words of each of the quad's lanes).  Lanes of equal r = lane & 3 are summed inside their row of 16 (one step on
// the bytes, <= 128; widened to 16-bit fields; one more step), the sixteen rows of the workgroup file their sums, and
// 16 NSET threads add them up.  Thread b < 32 receives in out[set] the total for bit b, as from block_bit_totals.
// scratch: [17][NSET * 16] dwords of LDS.
template <int NSET>
__device__ __forceinline__ void block_bit_totals_quad(const uint32_t (*c2)[2], uint32_t *scratch, uint32_t out[NSET]) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, row = lane >> 4, r = lane & 3u;
    const bool files = (lane & 12u) == 12u;
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
#pragma unroll
        for (uint32_t jj = 0; jj < 2; ++jj) {
            uint32_t v = c2[s][jj];
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xF, 0xF, false)); // row_ror:8
            uint32_t lo = v & 0x00FF00FFu, hi = (v >> 8) & 0x00FF00FFu; // bits j, j+16 | j+8, j+24
            lo += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), 0x124, 0xF, 0xF, false)); // row_ror:4
            hi += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), 0x124, 0xF, 0xF, false));
            if (files) {
                const uint32_t at = ((wave * 4 + row) * NSET + s) * 16 + (2 * r + jj) * 2;
                scratch[at] = lo;
                scratch[at + 1] = hi;
            }
        }
    }
    __syncthreads();
    uint32_t *total = scratch + 16 * (NSET * 16);
    if (tid < NSET * 16) { // 16-bit fields cannot overflow: a total is at most the site count
        uint32_t sum = 0;
#pragma unroll 4
        for (uint32_t g = 0; g < 16; ++g) sum += scratch[g * (NSET * 16) + tid];
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
