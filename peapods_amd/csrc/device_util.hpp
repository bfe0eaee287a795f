// Device-side helpers shared by the kernel translation units (kernels.hip: unit-coupling
// bit-sliced path; kernels_general.hip: f32-coupling path).
#pragma once

#include "engine.hpp"

#include <cstdint>
#include <hip/hip_runtime.h>

namespace pa {

// ---------------------------------------------------------------------------
// xoshiro256** on the device.  State is kept as four u64; hipcc lowers the
// 64-bit shifts/rotates to v_alignbit / v_lshl_add pairs.
// ---------------------------------------------------------------------------
struct DevRng {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ void advance() {
        const uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 45) | (s3 >> 19);
    }
    // (plain C: wave-uniform generators -- k_wolff_wave -- stay on the scalar unit)
    __device__ __forceinline__ uint64_t next() {
        const uint64_t m5 = s1 * 5;
        const uint64_t result = ((m5 << 7) | (m5 >> 57)) * 9;
        advance();
        return result;
    }
    // The same step for per-lane generators: s1 * 5 and * 9 as one 64-bit shift-add each (v_lshl_add_u64); hipcc
    // lowers the two multiplications to six v_mad_u64_u32 otherwise (the event replay of the random-stream kernel
    // steps the ladder generators in every workgroup).
    __device__ __forceinline__ uint64_t next_lane() {
        uint64_t m5, result;
        asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(m5) : "v"(s1));
        const uint64_t rot = (m5 << 7) | (m5 >> 57);
        asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(result) : "v"(rot));
        advance();
        return result;
    }
};

// floor(n / d) through the reciprocal m = ceil(2^32 / d) the host precomputed: exact whenever n d < 2^32 (the error
// n (m d - 2^32) / (d 2^32) then stays below 1 / d) -- tile words by W <= 16, slots by n_temps, tasks by n_replicas.
// A runtime u32 division costs ~25 VALU instructions per use on this target.
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t recip) { return recip ? __umulhi(n, recip) : n; } // (recip 0: d = 1)

__device__ __forceinline__ void load_cursor(const uint32_t *cursor, uint32_t real, uint32_t SP,
                                            uint32_t slot, uint32_t st[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = cursor[(static_cast<size_t>(real) * 8 + j) * SP + slot];
}

__device__ __forceinline__ void store_cursor(uint32_t *cursor, uint32_t real, uint32_t SP,
                                             uint32_t slot, const uint32_t st[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cursor[(static_cast<size_t>(real) * 8 + j) * SP + slot] = st[j];
}

// ---------------------------------------------------------------------------
// xoshiro256** in 32-bit halves: every operation is a full-rate VALU op (no
// 64-bit multiply); only the top 24 bits of the output are formed, which is
// all the reference uses (next_u32 >> 8 of next_u64 >> 32, sweep.rs:181).
// ---------------------------------------------------------------------------
// (x << SH) + y in one instruction.  hipcc canonicalises (x << 2) + x to a multiplication and
// lowers it to three v_mad_u64_u32, so the instruction is named explicitly.
template <int SH> __device__ __forceinline__ uint64_t lshl_add_u64(uint64_t x, uint64_t y) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SH), "v"(y));
    return r;
}

struct Rng32 {
    uint32_t a0, a1, b0, b1, c0, c1, d0, d1; // s0 = a1:a0, s1 = b1:b0, s2 = c1:c0, s3 = d1:d0
    __device__ __forceinline__ void load(const uint32_t st[8]) {
        a0 = st[0]; a1 = st[1]; b0 = st[2]; b1 = st[3];
        c0 = st[4]; c1 = st[5]; d0 = st[6]; d1 = st[7];
    }
    __device__ __forceinline__ void store(uint32_t st[8]) const {
        st[0] = a0; st[1] = a1; st[2] = b0; st[3] = b1;
        st[4] = c0; st[5] = c1; st[6] = d0; st[7] = d1;
    }
    // Returns the high word of the 64-bit output; the reference's draw is its top 24 bits.
    __device__ __forceinline__ uint32_t next_top() {
        // output = rotl(s1 * 5, 7) * 9, both multiplications as one 64-bit shift-add each
        // (v_lshl_add_u64); only the high word of the result is needed
        const uint64_t s1 = (static_cast<uint64_t>(b1) << 32) | b0;
        const uint64_t pp = lshl_add_u64<2>(s1, s1);
        const uint32_t plo = static_cast<uint32_t>(pp), phi = static_cast<uint32_t>(pp >> 32);
        const uint32_t rhi = __builtin_amdgcn_alignbit(phi, plo, 25); // R = rotl(P, 7)
        const uint32_t rlo = __builtin_amdgcn_alignbit(plo, phi, 25);
        const uint64_t rr = (static_cast<uint64_t>(rhi) << 32) | rlo;
        const uint32_t top = static_cast<uint32_t>(lshl_add_u64<3>(rr, rr) >> 32);
        // state transition; three-input XORs (v_bitop3_b32, truth table 0x96) fold the chained updates.
        // s1 << 17 as ONE 64-bit shift: tools/probe/step_variants.hip measured the two-instruction form
        // (v_lshlrev_b32 + v_alignbit_b32) 2.3 us per launch slower on the bench ensemble's random stream
        uint64_t tt;
        asm("v_lshlrev_b64 %0, 17, %1" : "=v"(tt) : "v"(s1));
        const uint32_t t0 = static_cast<uint32_t>(tt), t1 = static_cast<uint32_t>(tt >> 32);
        const uint32_t e0 = d0 ^ b0, e1 = d1 ^ b1;                      // s3 ^ s1
        const uint32_t nb0 = __builtin_amdgcn_bitop3_b32(b0, c0, a0, 0x96); // s1 ^ (s2 ^ s0)
        const uint32_t nb1 = __builtin_amdgcn_bitop3_b32(b1, c1, a1, 0x96);
        const uint32_t nc0 = __builtin_amdgcn_bitop3_b32(c0, a0, t0, 0x96); // (s2 ^ s0) ^ (s1 << 17)
        const uint32_t nc1 = __builtin_amdgcn_bitop3_b32(c1, a1, t1, 0x96);
        a0 ^= e0; a1 ^= e1;                                             // s0 ^ (s3 ^ s1)
        b0 = nb0; b1 = nb1;
        c0 = nc0; c1 = nc1;
        d0 = __builtin_amdgcn_alignbit(e1, e0, 19);                     // rotl(s3 ^ s1, 45)
        d1 = __builtin_amdgcn_alignbit(e0, e1, 19);
        return top; // the 24-bit draw is top >> 8
    }
};

// state <- M * state with M as nibble tables in LDS (or global memory), layout [64 positions][2 halves][16 values][4]:
// the sixteen 16-byte half rows a ds_read_b128 can pick for one (position, half) are 256 contiguous bytes = all 64
// banks once, so the sixteen lanes of a read group never conflict whatever their nibbles are (equal nibbles
// broadcast).  With whole 32-byte rows side by side (round 2) values v and v + 8 shared their banks and 46 % of the
// random-stream kernel's LDS cycles were conflict cycles (SQ_LDS_BANK_CONFLICT, profiles/r02_pmc_per_launch.csv).
__device__ __forceinline__ void jump_state_lds(uint32_t st[8], const uint32_t *tab) {
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 64; p += 2) { // two table rows per round: one three-input XOR (v_bitop3) folds both in
        const uint32_t nib0 = (st[p >> 3] >> ((p & 7) * 4)) & 15u;
        const uint32_t nib1 = (st[(p + 1) >> 3] >> (((p + 1) & 7) * 4)) & 15u;
        const uint4 *e0 = reinterpret_cast<const uint4 *>(tab + p * 128 + nib0 * 4);
        const uint4 *e1 = reinterpret_cast<const uint4 *>(tab + (p + 1) * 128 + nib1 * 4);
        const uint4 a = e0[0], b = e0[16], c = e1[0], e = e1[16]; // second half: + 64 dwords
        acc[0] = __builtin_amdgcn_bitop3_b32(acc[0], a.x, c.x, 0x96);
        acc[1] = __builtin_amdgcn_bitop3_b32(acc[1], a.y, c.y, 0x96);
        acc[2] = __builtin_amdgcn_bitop3_b32(acc[2], a.z, c.z, 0x96);
        acc[3] = __builtin_amdgcn_bitop3_b32(acc[3], a.w, c.w, 0x96);
        acc[4] = __builtin_amdgcn_bitop3_b32(acc[4], b.x, e.x, 0x96);
        acc[5] = __builtin_amdgcn_bitop3_b32(acc[5], b.y, e.y, 0x96);
        acc[6] = __builtin_amdgcn_bitop3_b32(acc[6], b.z, e.z, 0x96);
        acc[7] = __builtin_amdgcn_bitop3_b32(acc[7], b.w, e.w, 0x96);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = acc[j];
}


// The same jump by a whole wave from a table in device memory: lane p looks up the row of nibble p (one 32-byte entry
// per lane instead of 64 dependent-looking loads in one), the rows are folded by an XOR butterfly, and every lane
// ends up with the new state.  st must hold the same state in every lane; all 64 lanes must be active.
__device__ __forceinline__ void jump_state_wave(uint32_t st[8], const uint32_t *tab) {
    const uint32_t p = threadIdx.x & 63u;
    uint32_t word = st[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) word = (p >> 3) == static_cast<uint32_t>(k) ? st[k] : word;
    const uint32_t nib = (word >> ((p & 7u) * 4)) & 15u;
    const uint4 *e = reinterpret_cast<const uint4 *>(tab + p * 128 + nib * 4);
    const uint4 a = e[0], b = e[16]; // second half: + 64 dwords
    uint32_t acc[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] ^= __shfl_xor(acc[j], off);
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = acc[j];
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which
// would stall every schedule pass on the global prefetch issued just before it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Deferred tempering swaps (masks: kernels.hip, 'Deferred tempering swaps').
// One round on a window of three consecutive words; returns the new middle word.  (prev, next)
// are the old neighbouring words; only their bits next to the middle word matter.
__device__ __forceinline__ uint32_t swap_round_mid(uint32_t prev, uint32_t v, uint32_t next,
                                                   uint32_t m_prev, uint32_t m_cur) {
    const uint32_t t = (v ^ ((v >> 1) | (next << 31))) & m_cur;      // pairs whose lower bit is in v
    const uint32_t t_prev31 = ((prev >> 31) ^ (v & 1u)) & (m_prev >> 31); // pair (31 of prev, 0 of v)
    return v ^ t ^ ((t << 1) | t_prev31);
}

// Both rounds for the middle word of a window (prev, v, next); masks are [round][word-1..word+1].
__device__ __forceinline__ uint32_t apply_swaps_window(uint32_t prev, uint32_t v, uint32_t next,
                                                       const uint32_t mp[2], const uint32_t mc[2],
                                                       const uint32_t mn[2]) {
    // round 0 on all three words (the outer two only need to be right next to the middle word,
    // which they are: their far neighbours cannot reach bit 31 of prev / bit 0 of next in one round
    // except through pairs that are themselves inside the window)
    const uint32_t v0 = swap_round_mid(prev, v, next, mp[0], mc[0]);
    const uint32_t p0 = swap_round_mid(0u, prev, v, 0u, mp[0]);
    const uint32_t n0 = swap_round_mid(v, next, 0u, mc[0], mn[0] & 0x7FFFFFFFu);
    return swap_round_mid(p0, v0, n0, mp[1], mc[1]);
}


// Workgroups go to the eight XCDs round-robin (blockIdx % 8) and every XCD has its own L2.  The words of one realisation
// share its schedule table (and the overlap reads the partner replicas' words), so they get block indices eight apart:
// the table then enters one L2 once per launch instead of once per word (raw FETCH_SIZE of the sweep pair: 25.2 -> 18.9 MB).
__device__ __forceinline__ void tile_of_block(const SimDims &d, uint32_t bid, uint32_t &real, uint32_t &w) {
    const uint32_t group = 8u * d.W, eights = d.n_real / 8u;
    if (bid < eights * group) {
        real = (bid / group) * 8u + (bid & 7u);
        w = (bid % group) >> 3;
    } else { // the realisations left over when their number is not a multiple of eight
        const uint32_t rest = bid - eights * group;
        real = eights * 8u + rest / d.W;
        w = rest % d.W;
    }
}


// Schedule record of one site (uint4, 16-bit fields holding LDS byte offsets = 4 * site index):
//   x = site | f0 << 16,  y = b0 | f1 << 16,  z = b1 | f2 << 16,  w = b2 | jmeta << 16
// Each realisation has N + 1 records; record N is a dummy whose site and neighbours all point
// at the spare LDS slot N, so idle lanes of a pass run the same code without a branch.
struct SiteRec {
    uint32_t site, nb[6], jm; // byte offsets into a u32[N + 1] LDS array
    __device__ __forceinline__ explicit SiteRec(const uint4 r) {
        site = r.x & 0xFFFFu;
        nb[0] = r.x >> 16;
        nb[1] = r.y & 0xFFFFu;
        nb[2] = r.y >> 16;
        nb[3] = r.z & 0xFFFFu;
        nb[4] = r.z >> 16;
        nb[5] = r.w & 0xFFFFu;
        jm = r.w >> 16;
    }
};

__device__ __forceinline__ uint32_t lds_word(const uint32_t *base, uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + byte_off);
}


// Cooperative global -> LDS copy of n dwords: loads are issued in batches of 8 before the first
// store, so a wave keeps 8 requests in flight instead of one round trip per iteration.
__device__ __forceinline__ void copy_to_lds(uint32_t *dst, const uint32_t *__restrict__ src,
                                            uint32_t n, uint32_t tid, uint32_t nthreads) {
    // 16-byte path (both pointers 16-byte aligned, n a multiple of 4): 8 dwordx4 loads in flight
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (n & 3u) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        const uint32_t n4 = n >> 2;
        for (uint32_t base = 0; base < n4; base += 8 * nthreads) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * nthreads + tid;
                v[k] = i < n4 ? s4[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * nthreads + tid;
                if (i < n4) d4[i] = v[k];
            }
        }
        return;
    }
    for (uint32_t base = 0; base < n; base += 8 * nthreads) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t i = base + k * nthreads + tid;
            v[k] = i < n ? src[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t i = base + k * nthreads + tid;
            if (i < n) dst[i] = v[k];
        }
    }
}


struct BigRec {
    uint32_t site, nb[6], jm;
    __device__ __forceinline__ BigRec(const uint4 a, const uint4 b) {
        site = a.x; nb[0] = a.y; nb[1] = a.z; nb[2] = a.w;
        nb[3] = b.x; nb[4] = b.y; nb[5] = b.z; jm = b.w;
    }
};

__device__ __forceinline__ uint32_t coherent_load(const uint32_t *ptr) {
    return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store(uint32_t *ptr, uint32_t v) {
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Waits until the L2 has acknowledged every global store / atomic this wave has issued (s_waitcnt
// vmcnt(0); expcnt and lgkmcnt left alone).  A workgroup barrier does not imply it: its fence is
// workgroup-scope, which needs no vmcnt wait on one CU, so a store issued just before the barrier
// can still be in flight when another wave -- or, after a progress flag, another workgroup -- reads
// the location through the L2.
__device__ __forceinline__ void drain_global_stores() { __builtin_amdgcn_s_waitcnt(0x0F70); }


// expf as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c, glibc >= 2.27): x N / ln2 = k + r with
// N = 32, exp(x) = 2^(k/N) 2^(r/N) ~ T[k mod N] 2^(k div N) (C0 r^3 + C1 r^2 + C2 r + 1), all in f64 with
// one final rounding.  T[i] = bits(2^(i/32)) - (i << 47).  The engine compares it with the host's
// expf on 2^24 arguments before the first use on a device (ensure_device_exp).
__device__ const unsigned long long kExpfTab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

__device__ __forceinline__ float dev_expf(float x) {
    const double xd = static_cast<double>(x);
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) { // |x| >= 88 or nan
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return __fadd_rn(x, x);
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double z = __dmul_rn(0x1.71547652b82fep+0 * 32, xd);
    double kd = __dadd_rn(z, 0x1.8p+52);
    const unsigned long long ki = static_cast<unsigned long long>(__double_as_longlong(kd));
    kd = __dsub_rn(kd, 0x1.8p+52);
    const double r = __dsub_rn(z, kd);
    const unsigned long long t = kExpfTab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double(static_cast<long long>(t));
    z = __dadd_rn(__dmul_rn(0x1.c6af84b912394p-5 / 32 / 32 / 32, r), 0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double r2 = __dmul_rn(r, r);
    double y = __dadd_rn(__dmul_rn(0x1.62e42ff0c52d6p-1 / 32, r), 1.0);
    y = __dadd_rn(__dmul_rn(z, r2), y);
    y = __dmul_rn(y, s);
    return static_cast<float>(y);
}

// gen::<f32>() < p on the 2^24 grid of draws  <=>  draw < ceil(p 2^24)
__device__ __forceinline__ uint32_t draw_threshold(float p) {
    const double scaled = ceil(static_cast<double>(p) * 16777216.0);
    return scaled <= 0.0 ? 0u : (scaled >= 16777216.0 ? (1u << 24) : static_cast<uint32_t>(scaled));
}
// bond probabilities of the cluster moves for a satisfied bond of strength |J| at temperature T
// (overlap.rs:430: 1 - exp(-4 |J| / T); fk.rs:74,119: 1 - exp(-2 |J| / T); overlap.rs:601-603, 792-793:
// r = exp(-2 |J| / T), blue 1 - r r, red 1 - r), in the reference's f32 operation order
__device__ __forceinline__ uint32_t jorg_threshold(float j_abs, float temp) {
    return draw_threshold(__fsub_rn(1.0f, dev_expf(__fdiv_rn(__fmul_rn(-4.0f, j_abs), temp))));
}
__device__ __forceinline__ uint32_t fk_threshold(float j_abs, float temp) {
    return draw_threshold(__fsub_rn(1.0f, dev_expf(__fdiv_rn(__fmul_rn(-2.0f, j_abs), temp))));
}
__device__ __forceinline__ uint32_t cmr_blue_threshold(float j_abs, float temp) {
    const float r = dev_expf(__fdiv_rn(__fmul_rn(-2.0f, j_abs), temp));
    return draw_threshold(__fsub_rn(1.0f, __fmul_rn(r, r)));
}

// The 12 neighbour slots of a site (f0, b0, ..., f5, b5; u16 each) fetched with three 8-byte loads, so
// a serial search pays one memory latency per site instead of one per neighbour.
struct NbrRow {
    uint64_t q[3];
    __device__ __forceinline__ NbrRow(const uint16_t *nbr_site, uint32_t site) {
        const uint64_t *row = reinterpret_cast<const uint64_t *>(nbr_site + static_cast<size_t>(site) * 12);
        q[0] = row[0];
        q[1] = row[1];
        q[2] = row[2];
    }
    __device__ __forceinline__ uint32_t at(uint32_t dir) const {
        const uint64_t v = dir < 4 ? q[0] : (dir < 8 ? q[1] : q[2]);
        return static_cast<uint32_t>(v >> ((dir & 3u) * 16)) & 0xFFFFu;
    }
};

} // namespace pa
