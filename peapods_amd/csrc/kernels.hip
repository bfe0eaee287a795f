// HIP kernels of the MI355X spin-sim engine (gfx950 only; wave = 64 lanes).
//
//   k_rng        reference-exact random stream -> acceptance-level bit planes
//   k_sweep      level-ordered Metropolis sweep on 32 slots per word (+ energy / magnetisation)
//   k_overlap    replica-overlap and link-overlap dot products
//   k_accumulate running f64 statistics, P(q) histograms
//   k_pt         parallel-tempering swaps (bit-column exchange)
//
// Decisions are integer throughout (24-bit draw < table threshold); floating
// point appears only where the reference has it (one f32 division per
// observable, the f32 PT criterion, f64 running sums) and is written with
// explicit IEEE intrinsics so nothing is contracted or reassociated.
#include "engine.hpp"

namespace pa {

// ---------------------------------------------------------------------------
// xoshiro256** on the device.  State is kept as four u64; hipcc lowers the
// 64-bit shifts/rotates to v_alignbit / v_lshl_add pairs.
// ---------------------------------------------------------------------------
struct DevRng {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ uint64_t next() {
        const uint64_t m5 = s1 * 5;
        const uint64_t result = ((m5 << 7) | (m5 >> 57)) * 9;
        const uint64_t t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 45) | (s3 >> 19);
        return result;
    }
};

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

// state <- M * state with M given as nibble tables [64][16][8].
__device__ __forceinline__ void jump_state(uint32_t st[8], const uint32_t *__restrict__ tab) {
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 64; ++p) {
        const uint32_t nib = (st[p >> 3] >> ((p & 7) * 4)) & 15u;
        const uint4 *e = reinterpret_cast<const uint4 *>(tab + (p * 16 + nib) * 8);
        const uint4 a = e[0], b = e[1];
        acc[0] ^= a.x;
        acc[1] ^= a.y;
        acc[2] ^= a.z;
        acc[3] ^= a.w;
        acc[4] ^= b.x;
        acc[5] ^= b.y;
        acc[6] ^= b.z;
        acc[7] ^= b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = acc[j];
}

// ---------------------------------------------------------------------------
// k_rng: one wave = 32 slots of one tile word x 2 consecutive site blocks.
// Lane l: slot bit = l & 31, block = 2*bp + (l >> 5).  Every lane walks its
// block's sites in order, drawing the reference's next 24-bit value for its
// slot (sweep.rs:181) and comparing it with the slot's three thresholds; the
// ballot of those comparisons IS the 32-slot word of that site.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rng(SimDims d, DevPtrs p) {
    const uint32_t task = blockIdx.x;
    const uint32_t bp = task % d.BP;
    const uint32_t w = (task / d.BP) % d.W;
    const uint32_t real = task / (d.BP * d.W);
    const uint32_t lane = threadIdx.x;
    const uint32_t bit = lane & 31u, half = lane >> 5;
    const uint32_t block = 2 * bp + half;
    const uint32_t slot = w * 32 + bit;
    const bool slot_ok = slot < d.n_sys;
    const bool active = slot_ok && block < d.B;

    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        load_cursor(p.cursor, real, d.SP, slot, st);
        if (block > 0) jump_state(st, p.jump + static_cast<size_t>(block) * (64 * 16 * 8));
    }
    DevRng rng;
    rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
    rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
    rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
    rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];

    uint32_t ca0 = 0, ca1 = 0, ca2 = 0, cb0 = 0, cb1 = 0, cb2 = 0;
    if (slot_ok) {
        const uint32_t t = slot % d.n_temps;
        ca0 = p.thr[0 * d.n_temps + t];
        ca1 = p.thr[1 * d.n_temps + t];
        ca2 = p.thr[2 * d.n_temps + t];
        cb0 = p.thr[3 * d.n_temps + t];
        cb1 = p.thr[4 * d.n_temps + t];
        cb2 = p.thr[5 * d.n_temps + t];
    }

    const uint32_t begin = block * d.BS;
    const uint32_t end = active ? min(d.N, begin + d.BS) : begin;
    uint2 *planes = p.planes + static_cast<size_t>(real) * d.N * d.W;
    const uint32_t *mpar = p.mpar + static_cast<size_t>(real) * d.MW;

    for (uint32_t j = 0; j < d.BS; ++j) {
        const uint32_t site = begin + j;
        const bool valid = site < end; // false for idle lanes and past a short last block
        uint32_t level = 0;
        if (valid) {
            const uint32_t draw = static_cast<uint32_t>(rng.next() >> 40);
            uint32_t c0 = ca0, c1 = ca1, c2 = ca2;
            if (d.has_odd && ((mpar[site >> 5] >> (site & 31u)) & 1u)) {
                c0 = cb0;
                c1 = cb1;
                c2 = cb2;
            }
            level = (draw < c0) + (draw < c1) + (draw < c2);
        }
        const unsigned long long m0 = __ballot(level & 1u);
        const unsigned long long m1 = __ballot(level & 2u);
        if (bit == 0 && valid) {
            const uint32_t x = half ? static_cast<uint32_t>(m0 >> 32) : static_cast<uint32_t>(m0);
            const uint32_t y = half ? static_cast<uint32_t>(m1 >> 32) : static_cast<uint32_t>(m1);
            planes[static_cast<size_t>(site) * d.W + w] = make_uint2(x, y);
        }
    }
    // The last block ends exactly N draws after the cursor: that is the next cursor.
    if (active && block == d.B - 1) {
        // (a short last block drew only its own sites, see `valid` above)
        st[0] = static_cast<uint32_t>(rng.s0);
        st[1] = static_cast<uint32_t>(rng.s0 >> 32);
        st[2] = static_cast<uint32_t>(rng.s1);
        st[3] = static_cast<uint32_t>(rng.s1 >> 32);
        st[4] = static_cast<uint32_t>(rng.s2);
        st[5] = static_cast<uint32_t>(rng.s2 >> 32);
        st[6] = static_cast<uint32_t>(rng.s3);
        st[7] = static_cast<uint32_t>(rng.s3 >> 32);
        store_cursor(p.cursor, real, d.SP, slot, st);
    }
}

// ---------------------------------------------------------------------------
// Bit-sliced helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) {
    return (a & b) | (c & (a ^ b));
}

// Accept mask for one site word.  nb[k]: neighbour words in the order f0,b0,f1,b1,f2,b2;
// jm: neg bits [0..5], nz bits [6..11], h = ceil(m/2) in [12..13].
//   U = number of unsatisfied non-zero bonds (dE = 2U - m), L = acceptance level of the draw.
//   flip  <=>  U + L >= h     (dE >= 0  or  draw < table[dE]), sweep.rs:170-185.
__device__ __forceinline__ uint32_t accept_mask(uint32_t s, const uint32_t nb[6], uint32_t jm,
                                                uint32_t l0, uint32_t l1) {
    uint32_t u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t neg = 0u - ((jm >> k) & 1u);
        const uint32_t nz = 0u - ((jm >> (6 + k)) & 1u);
        u[k] = (s ^ nb[k] ^ neg) & nz;
    }
    const uint32_t s1 = u[0] ^ u[1] ^ u[2], c1 = maj3(u[0], u[1], u[2]);
    const uint32_t s2 = u[3] ^ u[4] ^ u[5], c2 = maj3(u[3], u[4], u[5]);
    const uint32_t U0 = s1 ^ s2, c3 = s1 & s2;
    const uint32_t U1 = c1 ^ c2 ^ c3, U2 = maj3(c1, c2, c3);
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t t2 = U2 ^ k1, t3 = U2 & k1;
    const uint32_t hi = t2 | t3;
    const uint32_t ge1 = t0 | t1 | hi, ge2 = t1 | hi, ge3 = (t0 & t1) | hi;
    const uint32_t h = (jm >> 12) & 3u;
    return h == 0 ? 0xFFFFFFFFu : (h == 1 ? ge1 : (h == 2 ? ge2 : ge3));
}

// Byte-sliced vertical counters: c[j] holds four byte counters for bits j, j+8, j+16, j+24.
__device__ __forceinline__ void byte_count(uint32_t c[8], uint32_t x) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] += (x >> j) & 0x01010101u;
}

// Reduce NSET sets of byte counters over the 256-thread block.  Thread b < 32
// receives in out[set] the total for bit b.  scratch: [4][NSET*16] dwords of LDS.
template <int NSET>
__device__ __forceinline__ void block_bit_totals(const uint32_t (*c)[8], uint32_t *scratch,
                                                 uint32_t out[NSET]) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t lo = c[s][j] & 0x00FF00FFu;        // bits j (low half), j+16 (high half)
            uint32_t hi = (c[s][j] >> 8) & 0x00FF00FFu; // bits j+8, j+24
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                lo += __shfl_xor(lo, off);
                hi += __shfl_xor(hi, off);
            }
            if (lane == 0) {
                scratch[wave * (NSET * 16) + s * 16 + j * 2 + 0] = lo;
                scratch[wave * (NSET * 16) + s * 16 + j * 2 + 1] = hi;
            }
        }
    }
    __syncthreads();
    if (tid < 32) {
        const uint32_t j = tid & 7u, byte = tid >> 3;
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            uint32_t total = 0;
            for (int wv = 0; wv < 4; ++wv) {
                const uint32_t v = scratch[wv * (NSET * 16) + s * 16 + j * 2 + (byte & 1u)];
                total += (byte >> 1) ? (v >> 16) : (v & 0xFFFFu);
            }
            out[s] = total;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// k_sweep: workgroup = (tile, word).  The word's N site values live in LDS; the
// dependency levels run in order with one barrier each, which reproduces the
// reference's sequential site order (sweep.rs:58-61) bit for bit.
// Epilogue: energy sum_{i,d} J s_i s_fwd and magnetisation sum_i s_i for the
// word's 32 slots (energy.rs:92-109), by byte-sliced vertical counters.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sweep(SimDims d, DevPtrs p, int with_energy,
                                               int do_sweep) {
    extern __shared__ uint32_t lds[];
    uint32_t *sw = lds;               // [N]
    uint32_t *scratch = lds + d.N;    // [4 * 48]
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * d.N * d.W;
    for (uint32_t i = tid; i < d.N; i += 256) sw[i] = Sg[static_cast<size_t>(i) * d.W + w];
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint2 *planes = p.planes + static_cast<size_t>(real) * d.N * d.W;
    const uint16_t *jmeta = p.jmeta + static_cast<size_t>(real) * d.N;
    const uint4 *sched = reinterpret_cast<const uint4 *>(p.sched);
    __syncthreads();

    if (do_sweep) {
        for (uint32_t lvl = 0; lvl < d.n_levels; ++lvl) {
            const uint32_t begin = p.level_start[lvl], end = p.level_start[lvl + 1];
            for (uint32_t pos = begin + tid; pos < end; pos += 256) {
                const uint4 r0 = sched[2 * pos], r1 = sched[2 * pos + 1];
                const uint32_t site = r0.x;
                const uint32_t jm = jmeta[pos];
                const uint2 lv = planes[static_cast<size_t>(site) * d.W + w];
                uint32_t nb[6];
                nb[0] = sw[r0.y];
                nb[1] = sw[r0.z];
                nb[2] = sw[r0.w];
                nb[3] = sw[r1.x];
                nb[4] = sw[r1.y];
                nb[5] = sw[r1.z];
                const uint32_t s = sw[site];
                const uint32_t acc = accept_mask(s, nb, jm, lv.x, lv.y) & valid_bits;
                sw[site] = s ^ acc;
            }
            __syncthreads();
        }
        for (uint32_t i = tid; i < d.N; i += 256) Sg[static_cast<size_t>(i) * d.W + w] = sw[i];
    }

    if (with_energy) {
        uint32_t c[3][8];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[s][j] = 0;
        for (uint32_t pos = tid; pos < d.N; pos += 256) {
            const uint4 r0 = sched[2 * pos], r1 = sched[2 * pos + 1];
            const uint32_t jm = jmeta[pos];
            const uint32_t s = sw[r0.x];
            const uint32_t f0 = sw[r0.y], f1 = sw[r0.w], f2 = sw[r1.y];
            const uint32_t u0 = (s ^ f0 ^ (0u - ((jm >> 0) & 1u))) & (0u - ((jm >> 6) & 1u));
            const uint32_t u1 = (s ^ f1 ^ (0u - ((jm >> 2) & 1u))) & (0u - ((jm >> 8) & 1u));
            const uint32_t u2 = (s ^ f2 ^ (0u - ((jm >> 4) & 1u))) & (0u - ((jm >> 10) & 1u));
            byte_count(c[0], u0 ^ u1 ^ u2);      // weight 1
            byte_count(c[1], maj3(u0, u1, u2));  // weight 2
            byte_count(c[2], s);                 // spins equal to -1
        }
        uint32_t tot[3];
        block_bit_totals<3>(c, scratch, tot);
        if (tid < 32 && w * 32 + tid < d.n_sys) {
            const uint32_t slot = w * 32 + tid;
            const int32_t unsat = static_cast<int32_t>(tot[0] + 2 * tot[1]);
            p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = p.nz_bonds[real] - 2 * unsat;
            p.Msum[static_cast<size_t>(real) * d.n_sys + slot] =
                static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[2]);
        }
    }
}

// ---------------------------------------------------------------------------
// k_overlap: workgroup = (tile, word).  Replica 2p and 2p+1 at temperature t sit
// n_temps bit positions apart, so Y = V ^ (V >> n_temps) lines every pair up in
// the bits of the even replica.  V = S gives q_i; V = S ^ S(fwd_d) gives the
// bond products, and (q_i q_k) = +1 <=> the two replicas agree on the bond
// (overlap.rs:259-282).  Counts per bit by byte-sliced counters.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_overlap(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    uint32_t *A = lds;             // word w
    uint32_t *B0 = lds + d.N;      // word w + q
    uint32_t *B1 = lds + 2 * d.N;  // word w + q + 1
    uint32_t *scratch = lds + 3 * d.N;
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t q = d.n_temps >> 5, r = d.n_temps & 31u;
    const uint32_t *Sg = p.S + static_cast<size_t>(real) * d.N * d.W;
    const uint32_t wa = w + q, wb = w + q + 1;
    for (uint32_t i = tid; i < d.N; i += 256) {
        A[i] = Sg[static_cast<size_t>(i) * d.W + w];
        B0[i] = wa < d.W ? Sg[static_cast<size_t>(i) * d.W + wa] : 0u;
        B1[i] = wb < d.W ? Sg[static_cast<size_t>(i) * d.W + wb] : 0u;
    }
    __syncthreads();
    const uint4 *sched = reinterpret_cast<const uint4 *>(p.sched);
    uint32_t c[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    for (uint32_t pos = tid; pos < d.N; pos += 256) {
        const uint4 r0 = sched[2 * pos], r1 = sched[2 * pos + 1];
        const uint32_t site = r0.x;
        const uint32_t fw[3] = {r0.y, r0.w, r1.y};
        const uint32_t a = A[site];
        const uint32_t sh = r ? ((B0[site] >> r) | (B1[site] << (32 - r))) : B0[site];
        const uint32_t y = a ^ sh; // 1 where the pair disagrees at this site
        uint32_t z[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < static_cast<int>(d.nn)) {
                const uint32_t fa = A[fw[k]];
                const uint32_t fsh = r ? ((B0[fw[k]] >> r) | (B1[fw[k]] << (32 - r))) : B0[fw[k]];
                z[k] = y ^ fa ^ fsh; // 1 where q_i * q_k = -1
            } else {
                z[k] = 0;
            }
        }
        byte_count(c[0], y);
        byte_count(c[1], z[0] ^ z[1] ^ z[2]);
        byte_count(c[2], maj3(z[0], z[1], z[2]));
    }
    uint32_t tot[3];
    block_bit_totals<3>(c, scratch, tot);
    if (tid < 32) {
        const uint32_t k = w * 32 + tid;
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if ((rep & 1u) == 0 && rep + 1 < d.n_replicas && (rep >> 1) < d.n_pairs) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[0]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) -
                            2 * static_cast<int32_t>(tot[1] + 2 * tot[2]);
        }
    }
}

// ---------------------------------------------------------------------------
// k_accumulate: one thread per (tile, temperature); the reference's f64 running
// sums in its order (replica by replica, pair by pair: mod.rs:555-578,
// overlap.rs:259-325, stats.rs:17-27).
// ---------------------------------------------------------------------------
__global__ void k_accumulate(SimDims d, DevPtrs p, int with_overlap) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.n_real * d.n_temps) return;
    const uint32_t real = idx / d.n_temps, t = idx % d.n_temps;
    const float n_f = static_cast<float>(d.N);
    double *mom = p.mom + static_cast<size_t>(real) * 5 * d.n_temps;
    double m0 = mom[0 * d.n_temps + t], m1 = mom[1 * d.n_temps + t], m2s = mom[2 * d.n_temps + t];
    double e0 = mom[3 * d.n_temps + t], e1 = mom[4 * d.n_temps + t];
    for (uint32_t rep = 0; rep < d.n_replicas; ++rep) {
        const size_t slot = static_cast<size_t>(real) * d.n_sys + rep * d.n_temps + t;
        const float mag = __fdiv_rn(static_cast<float>(p.Msum[slot]), n_f);
        const float mag2 = __fmul_rn(mag, mag);
        const float mag4 = __fmul_rn(mag2, mag2);
        const float energy = __fdiv_rn(static_cast<float>(p.Esum[slot]), n_f);
        const double ed = static_cast<double>(energy);
        m0 = __dadd_rn(m0, static_cast<double>(mag));
        m1 = __dadd_rn(m1, static_cast<double>(mag2));
        m2s = __dadd_rn(m2s, static_cast<double>(mag4));
        e0 = __dadd_rn(e0, ed);
        e1 = __dadd_rn(e1, __dmul_rn(ed, ed));
    }
    mom[0 * d.n_temps + t] = m0;
    mom[1 * d.n_temps + t] = m1;
    mom[2 * d.n_temps + t] = m2s;
    mom[3 * d.n_temps + t] = e0;
    mom[4 * d.n_temps + t] = e1;

    if (!with_overlap || d.n_pairs == 0) return;
    double *ovm = p.ovm + static_cast<size_t>(real) * 6 * d.n_temps;
    double o[6];
    for (int k = 0; k < 6; ++k) o[k] = ovm[k * d.n_temps + t];
    const float n_bonds_f = static_cast<float>(d.N * d.nn);
    const size_t hbase = (static_cast<size_t>(real) * d.n_temps + t) * (d.N + 1);
    for (uint32_t pr = 0; pr < d.n_pairs; ++pr) {
        const size_t oidx = (static_cast<size_t>(real) * d.n_pairs + pr) * d.n_temps + t;
        const int32_t dot_spin = p.dot_spin[oidx], dot_link = p.dot_link[oidx];
        const float ql = __fdiv_rn(static_cast<float>(dot_link), n_bonds_f);
        const float qv = __fdiv_rn(static_cast<float>(dot_spin), n_f);
        const float q2 = __fmul_rn(qv, qv);
        const float q4 = __fmul_rn(q2, q2);
        const float ql2 = __fmul_rn(ql, ql);
        const float ql4 = __fmul_rn(ql2, ql2);
        o[0] = __dadd_rn(o[0], static_cast<double>(qv));
        o[1] = __dadd_rn(o[1], static_cast<double>(q2));
        o[2] = __dadd_rn(o[2], static_cast<double>(q4));
        o[3] = __dadd_rn(o[3], static_cast<double>(ql));
        o[4] = __dadd_rn(o[4], static_cast<double>(ql2));
        o[5] = __dadd_rn(o[5], static_cast<double>(ql4));
        const size_t bin = hbase + static_cast<size_t>((dot_spin + static_cast<int32_t>(d.N)) / 2);
        p.hist[bin] += 1;
        p.qlq[bin] = __dadd_rn(p.qlq[bin], static_cast<double>(ql));
        p.ql2q[bin] = __dadd_rn(p.ql2q[bin], static_cast<double>(ql2));
    }
    for (int k = 0; k < 6; ++k) ovm[k * d.n_temps + t] = o[k];
}

// ---------------------------------------------------------------------------
// k_pt: one workgroup per tile.  Thread 0 replays the reference's tempering
// event for every replica ladder in order (mod.rs:766-792, tempering.rs:20-102,
// realization.rs:73-120), drawing from the generator of system r*n_temps; the
// accepted swaps are then applied to the tile as adjacent bit-column exchanges
// by all threads.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void swap_u32(uint32_t *a, uint32_t *b) {
    const uint32_t t = *a;
    *a = *b;
    *b = t;
}

__global__ __launch_bounds__(256) void k_pt(SimDims d, DevPtrs p, PtParams pt) {
    __shared__ uint32_t masks[2][64]; // up to 2048 slots per tile
    const uint32_t real = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        masks[0][tid] = 0;
        masks[1][tid] = 0;
    }
    __syncthreads();
    if (tid == 0 && d.n_temps >= 2) {
        uint32_t *sid = p.system_ids + static_cast<size_t>(real) * d.n_sys;
        uint32_t *sof = p.slot_of + static_cast<size_t>(real) * d.n_sys;
        int32_t *E = p.Esum + static_cast<size_t>(real) * d.n_sys;
        int32_t *M = p.Msum + static_cast<size_t>(real) * d.n_sys;
        uint64_t *att = p.edge_att + static_cast<size_t>(real) * d.n_edges;
        uint64_t *accn = p.edge_acc + static_cast<size_t>(real) * d.n_edges;
        uint64_t *trips = p.round_trips + static_cast<size_t>(real) * d.n_sys;
        uint8_t *trip = p.trip_state + static_cast<size_t>(real) * d.n_sys;
        const float n_f = static_cast<float>(d.N);
        for (uint32_t rep = 0; rep < d.n_replicas; ++rep) {
            const uint32_t offset = rep * d.n_temps;
            uint32_t st[8];
            load_cursor(p.cursor, real, d.SP, sof[offset], st);
            DevRng rng;
            rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
            rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
            rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
            rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
            const uint32_t n_edges = d.n_temps - 1;
            const uint32_t n_attempts = pt.schedule == 0 ? 1u : n_edges;
            for (uint32_t a = 0; a < n_attempts; ++a) {
                uint32_t edge;
                uint32_t round = 0;
                if (pt.schedule == 0) {
                    // rand 0.8.5 gen_range(0..n_temps-1) on usize: widening multiply + zone
                    const uint64_t range = n_edges;
                    const uint64_t zone = (range << __clzll(range)) - 1;
                    for (;;) {
                        const uint64_t v = rng.next();
                        const uint64_t lo = v * range;
                        if (lo <= zone) {
                            edge = static_cast<uint32_t>(__umul64hi(v, range));
                            break;
                        }
                    }
                } else {
                    // edges of parity p first, then 1-p (tempering.rs:59-70)
                    const uint32_t n_first = (n_edges + 1 - pt.first_parity) / 2; // edges p, p+2, ...
                    if (a < n_first) {
                        edge = pt.first_parity + 2 * a;
                    } else {
                        edge = (1 - pt.first_parity) + 2 * (a - n_first);
                        round = 1;
                    }
                }
                const uint32_t sl = offset + edge, sr = offset + edge + 1;
                const uint32_t left = sid[sl], right = sid[sr];
                const float e1 = __fdiv_rn(static_cast<float>(E[sl]), n_f);
                const float e2 = __fdiv_rn(static_cast<float>(E[sr]), n_f);
                const float delta =
                    __fmul_rn(__fmul_rn(n_f, __fsub_rn(e2, e1)), p.dbeta[edge]);
                const uint32_t draw = static_cast<uint32_t>(rng.next() >> 40);
                const float log_rand = p.ln_table[draw];
                const bool accepted = delta >= log_rand;
                att[edge] += 1;
                if (accepted) {
                    accn[edge] += 1;
                    // record_arrival(left, edge + 1), record_arrival(right, edge)
                    const uint32_t arr_sys[2] = {left, right};
                    const uint32_t arr_slot[2] = {edge + 1, edge};
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t sys = arr_sys[k], slt = arr_slot[k];
                        if (slt == pt.hot_slot) {
                            if (trip[sys] == 2) trips[sys] += 1;
                            trip[sys] = 1;
                        } else if (slt == pt.cold_slot && trip[sys] == 1) {
                            trip[sys] = 2;
                        }
                    }
                    sid[sl] = right;
                    sid[sr] = left;
                    sof[right] = sl;
                    sof[left] = sr;
                    const int32_t te = E[sl];
                    E[sl] = E[sr];
                    E[sr] = te;
                    const int32_t tm = M[sl];
                    M[sl] = M[sr];
                    M[sr] = tm;
                    for (int j = 0; j < 8; ++j)
                        swap_u32(&p.cursor[(static_cast<size_t>(real) * 8 + j) * d.SP + sl],
                                 &p.cursor[(static_cast<size_t>(real) * 8 + j) * d.SP + sr]);
                    masks[round][sl >> 5] |= 1u << (sl & 31u);
                }
            }
            st[0] = static_cast<uint32_t>(rng.s0);
            st[1] = static_cast<uint32_t>(rng.s0 >> 32);
            st[2] = static_cast<uint32_t>(rng.s1);
            st[3] = static_cast<uint32_t>(rng.s1 >> 32);
            st[4] = static_cast<uint32_t>(rng.s2);
            st[5] = static_cast<uint32_t>(rng.s2 >> 32);
            st[6] = static_cast<uint32_t>(rng.s3);
            st[7] = static_cast<uint32_t>(rng.s3 >> 32);
            store_cursor(p.cursor, real, d.SP, sof[offset], st);
        }
    }
    __syncthreads();
    // Exchange bit columns (k, k+1) for every set mask bit k; round 0 then round 1.
    bool any = false;
    for (uint32_t w = 0; w < d.W; ++w) any = any || masks[0][w] || masks[1][w];
    if (!any) return;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * d.N * d.W;
    for (uint32_t i = tid; i < d.N; i += 256) {
        uint32_t *row = Sg + static_cast<size_t>(i) * d.W;
        for (int round = 0; round < 2; ++round) {
            uint32_t carry_t = 0; // bit 31 of the previous word's t
            for (uint32_t w = 0; w < d.W; ++w) {
                const uint32_t v = row[w];
                const uint32_t next = (w + 1 < d.W) ? row[w + 1] : 0u;
                const uint32_t shr1 = (v >> 1) | (next << 31);
                const uint32_t t = (v ^ shr1) & masks[round][w];
                row[w] = v ^ t ^ ((t << 1) | carry_t);
                carry_t = t >> 31;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
size_t sweep_lds_bytes(const SimDims &d) { return (static_cast<size_t>(d.N) + 4 * 48) * 4; }
static size_t overlap_lds_bytes(const SimDims &d) {
    return (3 * static_cast<size_t>(d.N) + 4 * 48) * 4;
}

hipError_t configure_kernels(const SimDims &d) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(sweep_lds_bytes(d)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_overlap),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(overlap_lds_bytes(d)));
}

hipError_t launch_rng(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_rng, dim3(d.n_real * d.W * d.BP), dim3(64), 0, stream, d, p);
    return hipGetLastError();
}

hipError_t launch_sweep(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                        hipStream_t stream) {
    hipLaunchKernelGGL(k_sweep, dim3(d.n_real * d.W), dim3(256), sweep_lds_bytes(d), stream, d, p,
                       with_energy ? 1 : 0, do_sweep ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_overlap(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_overlap, dim3(d.n_real * d.W), dim3(256), overlap_lds_bytes(d), stream, d,
                       p);
    return hipGetLastError();
}

hipError_t launch_accumulate(const SimDims &d, const DevPtrs &p, bool with_overlap,
                             hipStream_t stream) {
    const uint32_t total = d.n_real * d.n_temps;
    hipLaunchKernelGGL(k_accumulate, dim3((total + 63) / 64), dim3(64), 0, stream, d, p,
                       with_overlap ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_pt(const SimDims &d, const DevPtrs &p, const PtParams &pt, hipStream_t stream) {
    hipLaunchKernelGGL(k_pt, dim3(d.n_real), dim3(256), 0, stream, d, p, pt);
    return hipGetLastError();
}

} // namespace pa
