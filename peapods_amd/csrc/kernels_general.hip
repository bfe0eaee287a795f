// General-coupling path (gfx950): couplings are arbitrary f32 (Gaussian glasses), or the
// temperatures are outside the unit-coupling table's domain, so the reference takes its
// logarithmic rule (sweep.rs:35-48, 247-284):
//
//     flip  <=>  -s_i * h_i  >=  (T / 2) * ln(u)            Metropolis
//     flip  <=>  -s_i * h_i  >=  (T / 2) * ln(u / (1 - u))  Gibbs
//
// with h_i accumulated in f32 in the reference's order (fwd, bwd per direction, sweep.rs:8-19)
// and u = (next_u32 >> 8) * 2^-24.  Everything that decides a flip is reproduced bit for bit:
//   * the f32 additions run in the same order on values that are exact sign flips of J;
//   * ln is the platform libm's logf in the reference; dev_logf below restates glibc's
//     algorithm (sysdeps/ieee754/flt-32/e_logf.c, glibc >= 2.27) in f64 and the engine checks it
//     against the host's logf on all 2^24 possible draws before the first general sweep;
//   * spins stay bit-sliced (bit k of a tile word = temperature slot k), the site order is the
//     same dependency-level schedule and the random stream the same per-system xoshiro256**.
//
//   k_rng_f     random stream -> f32 thresholds (T/2) ln(..) per (site, slot)
//   k_sweep_f   level-ordered sweep, lane = (site, slot)
//   k_bonds_f   bond sign words  S_i ^ S_fwd(i,d)  for the energy chain
//   k_energy_f  the reference's sequential f32 energy sum per system (energy.rs:92-109)
#include "engine.hpp"

#include <cmath>
#include <cstdlib>

#include "device_util.hpp"
#include "bitcount.hpp"

namespace pa {

// ---------------------------------------------------------------------------
// logf as glibc computes it: x = 2^k z, z in [OFF, 2 OFF); 16-entry table of 1/c and ln c;
// ln(z/c) by a cubic in r = z/c - 1; all in f64, one final rounding to f32.
// ---------------------------------------------------------------------------
__device__ const double kLogfInvC[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
    0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
    0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
__device__ const double kLogfLogC[16] = {
    -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
    -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
    -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
    0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};

// tab: [16][2] doubles (1/c, ln c); the kernels stage it in LDS (a per-lane index into constant
// memory would be a global gather in front of every draw).
// Only draws reach it: x = u = k 2^-24 or x = u / (1 - u), i.e. zero or a positive normal number, for which
// glibc's special cases reduce to log(1) = +0 and log(0) = -inf -- applied as selects after the arithmetic, so
// the draw loop has no branches and unrolls.  (The self-test below compares all 2^24 values of both arguments
// with the host's logf before the first general sweep.)
__device__ __forceinline__ float dev_logf(float x, const double2 *tab) {
    const uint32_t ix = __float_as_uint(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int32_t k = static_cast<int32_t>(tmp) >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double2 entry = tab[i];
    const double invc = entry.x, logc = entry.y;
    const double z = static_cast<double>(__uint_as_float(iz));
    const double r = __dsub_rn(__dmul_rn(z, invc), 1.0);
    const double y0 = __dadd_rn(logc, __dmul_rn(static_cast<double>(k), 0x1.62e42fefa39efp-1));
    const double r2 = __dmul_rn(r, r);
    double y = __dadd_rn(__dmul_rn(0x1.5575b0be00b6ap-2, r), -0x1.ffffef20a4123p-2);
    y = __dadd_rn(__dmul_rn(-0x1.00ea348b88334p-2, r2), y);
    y = __dadd_rn(__dmul_rn(y, r2), __dadd_rn(y0, r));
    float result = static_cast<float>(y);
    result = ix == 0x3f800000u ? 0.0f : result;
    result = ix == 0u ? -INFINITY : result;
    return result;
}

__device__ __forceinline__ void stage_logf_table(double2 *tab, uint32_t tid) {
    if (tid < 16) tab[tid] = make_double2(kLogfInvC[tid], kLogfLogC[tid]);
}

__device__ __forceinline__ float uniform_of_draw(uint32_t draw24) {
    return __fmul_rn(static_cast<float>(draw24), 0x1p-24f); // exact: gen::<f32>()
}
__device__ __forceinline__ float logit_arg(float u) { return __fdiv_rn(u, __fsub_rn(1.0f, u)); }

// out_ln[k] = ln(k 2^-24), out_logit[k] = ln(u / (1 - u)): compared with the host libm by the
// engine before the first general-coupling sweep.
__global__ void k_log_selftest(float *out_ln, float *out_logit) {
    __shared__ double2 tab[16];
    stage_logf_table(tab, threadIdx.x);
    __syncthreads();
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << 24)) return;
    const float u = uniform_of_draw(k);
    out_ln[k] = dev_logf(u, tab);
    out_logit[k] = dev_logf(logit_arg(u), tab);
}

// s * j for a spin stored as bit `bit` of `word` (1 <=> -1): an exact sign flip.
__device__ __forceinline__ float signed_by_spin(float j, uint32_t word, uint32_t bit) {
    return __uint_as_float(__float_as_uint(j) ^ (((word >> bit) & 1u) << 31));
}

// ---------------------------------------------------------------------------
// k_rng_f: same work split as k_rng (4 waves share one site block and its jump table; lanes
// 0-31 / 32-63 are the 32 slots of two tile words).  Every lane walks the block's sites in
// order and stores the f32 threshold of its slot: thr_f[tile word][schedule position of the site][32 slots].
// ---------------------------------------------------------------------------
template <bool GIBBS>
__global__ __launch_bounds__(256) void k_rng_f(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t tab[]; // [64 * 16 * 8] jump table, then the logf table
    double2 *log_tab = reinterpret_cast<double2 *>(tab + 64 * 16 * 8);
    const uint32_t block = blockIdx.x % d.B;
    const uint32_t group = blockIdx.x / d.B;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t bit = lane & 31u, half = lane >> 5;
    const uint32_t n_tw = d.n_real * d.W;
    const uint32_t tw = (group * 4 + wave) * 2 + half;
    const bool tw_ok = tw < n_tw;
    const uint32_t real = tw_ok ? tw / d.W : 0, w = tw_ok ? tw % d.W : 0;
    const uint32_t slot = w * 32 + bit;
    const bool active = tw_ok && slot < d.n_sys;

    stage_logf_table(log_tab, tid);
    if (block > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.jump + static_cast<size_t>(block) * (64 * 16 * 8));
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = tid; i < 64 * 16 * 2; i += 256) dst[i] = src[i];
    }
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) load_cursor(p.cursor, real, d.SP, slot, st);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((group * 4 + wave) * 2) >= n_tw) return; // idle wave
    if (block > 0) jump_state_lds(st, tab);
    Rng32 rng;
    rng.load(st);

    const float half_t = active ? __fdiv_rn(p.temps[slot % d.n_temps], 2.0f) : 1.0f;
    const uint32_t begin = block * d.BS;
    const uint32_t len = min(d.BS, d.N - begin);
    // filed at the site's position in the sweep's schedule: k_sweep_f then knows a threshold's address
    // without first reading the site record
    float *out = p.thr_f + static_cast<size_t>(tw_ok ? tw : 0) * d.N * 32 + bit;
    // four draws per round: the logarithms (a table look-up and a dozen dependent f64 operations each) overlap
    uint32_t j = 0;
    for (; j + 4 <= len; j += 4) {
        float u[4], l[4];
        uint32_t at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u[k] = uniform_of_draw(rng.next_top() >> 8);
            at[k] = p.pos_of[begin + j + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) l[k] = dev_logf(GIBBS ? logit_arg(u[k]) : u[k], log_tab);
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[static_cast<size_t>(at[k]) * 32] = __fmul_rn(half_t, l[k]);
        }
    }
    for (; j < len; ++j) {
        const float u = uniform_of_draw(rng.next_top() >> 8);
        const float l = dev_logf(GIBBS ? logit_arg(u) : u, log_tab);
        if (active) out[static_cast<size_t>(p.pos_of[begin + j]) * 32] = __fmul_rn(half_t, l);
    }
    if (active && block == d.B - 1) {
        rng.store(st);
        store_cursor(p.cursor_out, real, d.SP, slot, st);
    }
}

// ---------------------------------------------------------------------------
// k_sweep_f: workgroup = (realisation, tile word), 512 threads.  A work item is (site of the
// current dependency level, group of 4 slots): the record decode, the neighbour words and the
// couplings are fetched once per item and serve four systems; the item's four decisions are
// XORed into the site's word with one LDS (or L2) atomic.  Records and thresholds of the next
// round are requested before the current round is computed.
// !BIG: the word's spins and the realisation's couplings ([site] -> float4 of forward bonds)
//       are staged in LDS.
//  BIG: both stay in global memory (spins through agent-scope atomics, as in k_sweep_big).
// ---------------------------------------------------------------------------
constexpr uint32_t kSweepFThreads = 512;
// couplings per site in the LDS copy (float4 up to three directions, two float4 up to six)
template <int NN> struct JRow { static constexpr int kWidth = NN <= 3 ? 4 : 8; };

template <bool BIG, int NN>
__global__ __launch_bounds__(kSweepFThreads) void k_sweep_f(SimDims d, DevPtrs p, int with_bonds) {
    constexpr int JW = JRow<NN>::kWidth;
    extern __shared__ uint32_t lds[];
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t tw = real * d.W + w;
    uint32_t *Sg = p.S + static_cast<size_t>(tw) * d.N;
    const float *Jg = p.Jf + static_cast<size_t>(real) * d.N * NN;
    const float4 *thr = reinterpret_cast<const float4 *>(p.thr_f + static_cast<size_t>(tw) * d.N * 32);
    const uint32_t n_pad = (d.N + 3u) & ~3u;
    uint32_t *sw = lds;                                      // [N]
    float *Jl = reinterpret_cast<float *>(lds + n_pad);      // [N][JW] forward couplings of each site
    if (!BIG) {
        copy_to_lds(sw, Sg, d.N, tid, kSweepFThreads);
        for (uint32_t i = tid; i < d.N; i += kSweepFThreads)
#pragma unroll
            for (int dd = 0; dd < JW; ++dd) Jl[i * JW + dd] = dd < NN ? Jg[i * NN + dd] : 0.0f;
        __syncthreads();
    }
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint32_t g = tid & 7u, shift = g * 4;              // this lane's four slots: bits shift..shift+3
    const uint32_t valid4 = (valid_bits >> shift) & 15u;
    // neighbour table in schedule order, shared by all realisations: site, f0, b0, f1, b1, ...
    const uint4 *nbr = reinterpret_cast<const uint4 *>(p.nbr_wide);

    struct Item {
        uint4 r[NN <= 3 ? 2 : 4]; // site + 2 NN neighbours
        float4 t;                 // thresholds of the four slots
    };
    auto fetch = [&](uint32_t pos, bool live_item) -> Item {
        Item it;
#pragma unroll
        for (int k = 0; k < (NN <= 3 ? 2 : 4); ++k) it.r[k] = make_uint4(0, 0, 0, 0);
        it.t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!live_item) return it;
#pragma unroll
        for (int k = 0; k < (NN <= 3 ? 2 : 4); ++k) it.r[k] = nbr[static_cast<size_t>(pos) * 4 + k];
        it.t = thr[static_cast<size_t>(pos) * 8 + g];
        return it;
    };
    // s * J for the lane's four slots of neighbour word `word`, added to h[0..3]
    auto add_bond = [&](float h[4], uint32_t word, float j) {
        const uint32_t x = word >> shift;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            h[k] = __fadd_rn(h[k], __uint_as_float(__builtin_amdgcn_bitop3_b32(
                                       x << (31 - k), 0x80000000u, __float_as_uint(j), 0x6A)));
    };
    auto run = [&](const Item &it) {
        const uint32_t *flat = reinterpret_cast<const uint32_t *>(it.r);
        const uint32_t site = flat[0];
        const uint32_t s = BIG ? coherent_load(Sg + site) : sw[site];
        float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < NN; ++dd) { // sweep.rs:11-17: forward then backward bond of each direction
            const uint32_t jf = flat[1 + 2 * dd], jb = flat[2 + 2 * dd];
            const float cf = BIG ? Jg[static_cast<size_t>(site) * NN + dd] : Jl[site * JW + dd];
            const float cb = BIG ? Jg[static_cast<size_t>(jb) * NN + dd] : Jl[jb * JW + dd];
            add_bond(h, BIG ? coherent_load(Sg + jf) : sw[jf], cf);
            add_bond(h, BIG ? coherent_load(Sg + jb) : sw[jb], cb);
        }
        // eng_change = -s_i * h (sweep.rs:43-44): s_i = +1 (bit clear) negates h
        const uint32_t xs = (~s) >> shift;
        const float tk[4] = {it.t.x, it.t.y, it.t.z, it.t.w};
        uint32_t mask = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = __uint_as_float(__builtin_amdgcn_bitop3_b32(xs << (31 - k), 0x80000000u,
                                                                        __float_as_uint(h[k]), 0x6A));
            mask |= (e >= tk[k]) ? (1u << k) : 0u;
        }
        mask &= valid4;
        if (mask) {
            if (BIG)
                __hip_atomic_fetch_xor(Sg + site, mask << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                atomicXor(&sw[site], mask << shift);
        }
    };

    // A step is one round of a level: items [base, base + 512) of the level's (site, slot group) pairs.  The
    // loads of step k + 1 (record and thresholds, neither depends on the spins) are issued before step k is
    // computed and stay in flight across the level barrier; the two item buffers alternate explicitly, so no
    // register copy waits for them.  (Fetching after the barrier exposed two dependent global latencies per level:
    // 16^3, 90 levels of one round each, 2.2 us per level.)
    struct Step {
        uint32_t pass, begin, end, base;
        bool valid;
    };
    auto items_of = [](const Step &s) -> uint32_t { return (s.end - s.begin) * 8; };
    auto next_step = [&](Step s) -> Step {
        if (s.base + kSweepFThreads < items_of(s)) {
            s.base += kSweepFThreads;
            return s;
        }
        s.pass += 1;
        s.base = 0;
        s.valid = s.pass < d.n_passes;
        if (s.valid) {
            s.begin = s.end;
            s.end = p.pass_start[s.pass + 1];
        }
        return s;
    };
    auto live_in = [&](const Step &s) -> bool { return s.valid && s.base + tid < items_of(s); };
    auto fetch_step = [&](const Step &s) -> Item { return fetch(s.begin + ((s.base + tid) >> 3), live_in(s)); };
    auto level_barrier = [&](const Step &s) {
        if (s.base + kSweepFThreads < items_of(s)) return; // more rounds of this level follow
        if (BIG) {
            drain_global_stores(); // the flips of this level are in the L2 before the next level loads
            __syncthreads();
        } else {
            lds_barrier(); // orders the LDS flips only: the next step's loads stay in flight
        }
    };
    Step sa{0, p.pass_start[0], p.pass_start[1], 0, d.n_passes > 0};
    Item A = fetch_step(sa);
    while (sa.valid) {
        Step sb = next_step(sa);
        const Item B = fetch_step(sb);
        if (live_in(sa)) run(A);
        level_barrier(sa);
        if (!sb.valid) break;
        sa = next_step(sb);
        A = fetch_step(sa);
        if (live_in(sb)) run(B);
        level_barrier(sb);
    }
    if (!BIG) {
        for (uint32_t i = tid; i < d.N; i += kSweepFThreads) Sg[i] = sw[i];
        if (with_bonds) { // the bond words k_energy_f reads (k_bonds_f's job), while the new spins are still in LDS
            for (uint32_t pos = tid; pos < d.N; pos += kSweepFThreads) {
                const uint32_t *row = p.nbr_wide + static_cast<size_t>(pos) * 16;
                const uint32_t site = row[0], s = sw[site];
                uint32_t out[JW];
#pragma unroll
                for (int k = 0; k < JW; ++k) out[k] = 0;
#pragma unroll
                for (int dd = 0; dd < NN; ++dd) out[dd] = s ^ sw[row[1 + 2 * dd]];
                out[JW - 1] = s;
                uint32_t *dst = p.xw + (static_cast<size_t>(tw) * d.N + site) * JW;
#pragma unroll
                for (int k = 0; k < JW; ++k) dst[k] = out[k];
            }
        }
    }
}

// xw[tile word][site][XW] = S_i ^ S_fwd(i, d) for every direction, then S_i itself (XW = 4 for up
// to three directions, 8 for up to six): bond sign words (bit set <=> s_i s_j = -1) and the spin
// word, for k_energy_f.
template <int NN>
__global__ __launch_bounds__(256) void k_bonds_f(SimDims d, DevPtrs p) {
    constexpr int XW = JRow<NN>::kWidth;
    const uint32_t tw = blockIdx.y;
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= d.N) return;
    const uint32_t *Sg = p.S + static_cast<size_t>(tw) * d.N;
    const uint32_t *row = p.nbr_wide + static_cast<size_t>(pos) * 16;
    const uint32_t site = row[0];
    const uint32_t s = Sg[site];
    uint32_t out[XW];
#pragma unroll
    for (int k = 0; k < XW; ++k) out[k] = 0;
#pragma unroll
    for (int dd = 0; dd < NN; ++dd) out[dd] = s ^ Sg[row[1 + 2 * dd]];
    out[XW - 1] = s;
    uint32_t *dst = p.xw + (static_cast<size_t>(tw) * d.N + site) * XW;
#pragma unroll
    for (int k = 0; k < XW; ++k) dst[k] = out[k];
}

// ---------------------------------------------------------------------------
// k_energy_f: the reference's energy is a strictly sequential f32 sum over (site, direction)
// in typewriter order (energy.rs:92-107), so each system is one dependent chain of N z'
// additions (4.2 ns each on a lone wave, tools/probe/addchain.hip: 52 us at 16^3 whatever else happens;
// the kernel takes 72 us: with 7 producer waves it took 90 us, the producers are the slower side).
// Workgroup = (realisation, pair of tile words), 16 waves.  Wave 0 CONSUMES: lane = one of the
// pair's 64 slots, nothing but LDS reads and the additions.  Waves 1-15 PRODUCE the signed terms
// +-J of the next chunk of sites into LDS (sign = the bond word's bit of the lane's slot):
// one pass of a producer wave serves 4 consecutive sites of both words (4 z' terms = z'
// 16-byte LDS writes per lane).  The producers' inputs (bond words, couplings) are loaded from
// global memory two chunks ahead and staged through LDS; one barrier per chunk.
// ---------------------------------------------------------------------------
constexpr uint32_t kEnergyThreads = 1024; // 1 consumer + 15 producer waves
constexpr uint32_t kProducers = kEnergyThreads / 64 - 1;

template <int NN> struct EnergyLds {
    static constexpr uint32_t kChunk = NN <= 3 ? 64 : 32;   // sites per chunk
    static constexpr uint32_t kTerms = kChunk * NN;          // terms per chunk and word
    static constexpr uint32_t kQuads = kTerms / 4;
    static constexpr int XW = JRow<NN>::kWidth;
    float4 terms[2][2][kQuads][32];   // [buffer][word of the pair][term / 4][slot] -> 4 terms
    uint32_t rawx[2][2][kChunk][XW];  // [buffer][word][site] bond words + spin word
    float rawj[2][kChunk][XW];        // [buffer][site] forward couplings
    uint32_t neg_total[64];
};

template <int NN>
__global__ __launch_bounds__(kEnergyThreads) void k_energy_f(SimDims d, DevPtrs p) {
    using Lds = EnergyLds<NN>;
    constexpr uint32_t kChunk = Lds::kChunk, kTerms = Lds::kTerms, kQuads = Lds::kQuads;
    constexpr int XW = Lds::XW;
    extern __shared__ uint32_t lds_raw[];
    Lds &L = *reinterpret_cast<Lds *>(lds_raw);
    const uint32_t n_pairs_w = (d.W + 1) / 2;
    const uint32_t real = blockIdx.x / n_pairs_w, w0 = (blockIdx.x % n_pairs_w) * 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, bit = lane & 31u, half = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool word_ok = w0 + half < d.W; // the pair's second word may not exist
    const uint32_t tw0 = real * d.W + w0;
    const uint32_t *__restrict__ xw = p.xw;
    const float *__restrict__ J = p.Jf + static_cast<size_t>(real) * d.N * NN;
    const uint32_t n_chunks = (d.N + kChunk - 1) / kChunk;
    if (tid < 64) L.neg_total[tid] = 0;

    // raw inputs of one chunk, element-wise over the producer threads: 2 kChunk XW bond words, then
    // kTerms couplings (at most 2 * 32 * 8 + 192 = 704 <= 3 x 448 elements)
    constexpr uint32_t kRawX = 2 * kChunk * XW, kRaw = kRawX + kTerms;
    constexpr uint32_t kPerThread = (kRaw + kProducers * 64 - 1) / (kProducers * 64);
    const uint32_t pt = tid - 64; // producer thread index (meaningless for wave 0)
    uint32_t reg[kPerThread];
    auto load_raw = [&](uint32_t chunk) {
        if (chunk >= n_chunks) return;
        const uint32_t base = chunk * kChunk;
#pragma unroll
        for (uint32_t r = 0; r < kPerThread; ++r) {
            const uint32_t e = r * kProducers * 64 + pt;
            uint32_t v = 0;
            if (e < kRawX) {
                const uint32_t wsel = e / (kChunk * XW), k = (e / XW) % kChunk, c = e % XW;
                if (base + k < d.N && w0 + wsel < d.W)
                    v = xw[(static_cast<size_t>(tw0 + wsel) * d.N + base + k) * XW + c];
            } else if (e < kRaw) {
                const size_t q = static_cast<size_t>(base) * NN + (e - kRawX);
                if (q < static_cast<size_t>(d.N) * NN) v = __float_as_uint(J[q]);
            }
            reg[r] = v;
        }
    };
    auto stash_raw = [&](uint32_t chunk) {
        if (chunk >= n_chunks) return;
#pragma unroll
        for (uint32_t r = 0; r < kPerThread; ++r) {
            const uint32_t e = r * kProducers * 64 + pt;
            if (e < kRawX) {
                reinterpret_cast<uint32_t *>(L.rawx[chunk & 1u])[e] = reg[r];
            } else if (e < kRaw) {
                const uint32_t q = e - kRawX;
                L.rawj[chunk & 1u][q / NN][q % NN] = __uint_as_float(reg[r]);
            }
        }
    };

    float total = 0.0f;
    uint32_t neg = 0;
    const uint32_t sh = 31u - bit;
    auto signed_term = [&](uint32_t xword, float jv) {
        return __uint_as_float(__builtin_amdgcn_bitop3_b32(xword << sh, 0x80000000u, __float_as_uint(jv), 0x6A));
    };
    // producer pass g: sites 4g .. 4g+3 of the chunk, lanes 0-31 for the pair's first word and
    // lanes 32-63 for its second; the 4 NN terms are NN consecutive quads of the lane's column
    auto produce = [&](uint32_t chunk) {
        if (chunk >= n_chunks) return;
        const uint32_t buf = chunk & 1u;
        for (uint32_t g = wave - 1; g < kChunk / 4; g += kProducers) {
            float t[4 * NN];
#pragma unroll
            for (uint32_t s4 = 0; s4 < 4; ++s4) {
                const uint32_t *x = L.rawx[buf][half][4 * g + s4];
                const float *j = L.rawj[buf][4 * g + s4];
#pragma unroll
                for (int dd = 0; dd < NN; ++dd) t[s4 * NN + dd] = signed_term(x[dd], j[dd]);
                neg += (x[XW - 1] >> bit) & 1u;
            }
#pragma unroll
            for (uint32_t q = 0; q < NN; ++q)
                L.terms[buf][half][g * NN + q][bit] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
        }
    };
    auto add4 = [&](const float4 v) {
        total = __fadd_rn(total, v.x);
        total = __fadd_rn(total, v.y);
        total = __fadd_rn(total, v.z);
        total = __fadd_rn(total, v.w);
    };
    auto consume = [&](uint32_t chunk) {
        const uint32_t n_terms = min(kChunk, d.N - chunk * kChunk) * NN;
        const float4(*buf)[32] = L.terms[chunk & 1u][half];
        if (n_terms == kTerms) {
            // full chunk: LDS reads run one batch of 8 quads ahead of the additions
            constexpr uint32_t kB = 8;
            static_assert(kQuads % kB == 0, "chunk terms must split into read batches");
            float4 cur[kB], nxt[kB];
#pragma unroll
            for (uint32_t m = 0; m < kB; ++m) cur[m] = buf[m][bit];
#pragma unroll
            for (uint32_t b = 0; b < kQuads / kB; ++b) {
                if (b + 1 < kQuads / kB) {
#pragma unroll
                    for (uint32_t m = 0; m < kB; ++m) nxt[m] = buf[(b + 1) * kB + m][bit];
                }
#pragma unroll
                for (uint32_t m = 0; m < kB; ++m) add4(cur[m]);
#pragma unroll
                for (uint32_t m = 0; m < kB; ++m) cur[m] = nxt[m];
            }
            return;
        }
        uint32_t q = 0;
        for (; q < n_terms / 4; ++q) add4(buf[q][bit]);
        const uint32_t rest = n_terms & 3u;
        if (rest) {
            const float4 v = buf[q][bit];
            total = __fadd_rn(total, v.x);
            if (rest > 1) total = __fadd_rn(total, v.y);
            if (rest > 2) total = __fadd_rn(total, v.z);
        }
    };

    // prologue: raw(0), raw(1) in LDS, raw(2) in flight, terms(0) produced
    if (wave > 0) {
        load_raw(0);
        stash_raw(0);
        load_raw(1);
        stash_raw(1);
        load_raw(2);
    }
    __syncthreads();
    if (wave > 0) produce(0);
    __syncthreads();
    // The chain of additions is the critical path (4.2 ns per addition on a lone wave, tools/probe/addchain.hip):
    // the consumer wave outranks the producer waves on its SIMD, and the chunk barrier orders LDS traffic only
    // (everything the waves exchange goes through LDS; the producers' global loads of chunk + 3 stay in flight).
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    for (uint32_t chunk = 0; chunk < n_chunks; ++chunk) {
        if (wave == 0) {
            consume(chunk);
        } else {
            produce(chunk + 1);     // from raw(chunk + 1), staged before the last barrier
            stash_raw(chunk + 2);   // overwrites raw(chunk), which produce(chunk) finished with
            load_raw(chunk + 3);
        }
        lds_barrier();
    }
    if (wave == 0) __builtin_amdgcn_s_setprio(0);
    if (wave > 0) atomicAdd(&L.neg_total[lane], neg);
    __syncthreads();
    const uint32_t slot = (w0 + half) * 32 + bit;
    if (tid < 64 && word_ok && slot < d.n_sys) {
        p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = __float_as_int(total); // raw f32 bits
        p.Msum[static_cast<size_t>(real) * d.n_sys + slot] =
            static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(L.neg_total[lane]);
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// k_energy_unit_f: unit couplings on the general path (four to six forward directions: bcc, fcc).  Every term of
// the reference's sequential f32 sum is -1, 0 or +1, so every partial sum is an integer below 2^24 in magnitude:
// exact in f32 in any order.  The chain of k_energy_f is then unnecessary -- count in parallel, convert once:
// E = (non-zero bonds) - 2 * (bonds whose term is -1), the latter counted for all 32 slots of a word at once.
// Workgroup = (realisation, tile word), 256 threads, 16 sites per thread and turn: the NN words
// v = (bond word ^ [J < 0]) & [J != 0] of a site go through a carry-save adder into three weight planes, which the
// byte-sliced counters of the on-chip kernels (count16, block_bit_totals) add up -- 99 -> a few us per launch at 16^3 fcc.
// ---------------------------------------------------------------------------
template <int NN>
__global__ __launch_bounds__(256) void k_energy_unit_f(SimDims d, DevPtrs p) {
    constexpr int XW = JRow<NN>::kWidth;
    __shared__ uint32_t scratch[5 * 64];
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t *xw = p.xw + static_cast<size_t>(real * d.W + w) * d.N * XW;
    const float *J = p.Jf + static_cast<size_t>(real) * d.N * NN;
    uint32_t c[4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    int nz_bonds = 0; // the same for every slot
#pragma unroll 1
    for (uint32_t base = 0; base < d.N; base += 16 * 256) {
        uint32_t q1[16], q2[16], q4[16], qs[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t site = base + k * 256 + tid;
            q1[k] = q2[k] = q4[k] = qs[k] = 0;
            if (site < d.N) {
                const uint32_t *x = xw + static_cast<size_t>(site) * XW;
                uint32_t v[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int dd = 0; dd < NN; ++dd) {
                    const float j = J[site * NN + dd]; // -1, 0, +1
                    const uint32_t nz = j != 0.0f ? 0xFFFFFFFFu : 0u, neg = j < 0.0f ? 0xFFFFFFFFu : 0u;
                    v[dd] = (x[dd] ^ neg) & nz; // 1 where the bond's term is -1
                    nz_bonds += j != 0.0f ? 1 : 0;
                }
                uint32_t s1, c1, s2, c2;
                full_add(v[0], v[1], v[2], s1, c1);
                full_add(v[3], v[4], v[5], s2, c2);
                const uint32_t c3 = s1 & s2;
                q1[k] = s1 ^ s2;
                q2[k] = c1 ^ c2 ^ c3;
                q4[k] = maj3(c1, c2, c3);
                qs[k] = x[XW - 1];
            }
        }
        count16(q1, c[0]);
        count16(q2, c[1]);
        count16(q4, c[2]);
        count16(qs, c[3]);
    }
    // the non-zero bonds of the realisation: every thread counted those of its sites
    __shared__ int nz_total;
    if (tid == 0) nz_total = 0;
    __syncthreads();
    atomicAdd(&nz_total, nz_bonds);
    uint32_t tot[4];
    block_bit_totals<4>(c, scratch, tot); // (its barriers also publish nz_total)
    const uint32_t slot = w * 32 + tid;
    if (tid < 32 && slot < d.n_sys) {
        const int minus = static_cast<int>(tot[0] + 2 * tot[1] + 4 * tot[2]);
        p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = __float_as_int(static_cast<float>(nz_total - 2 * minus)); // raw f32 bits
        p.Msum[static_cast<size_t>(real) * d.n_sys + slot] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[3]);
    }
}

size_t sweep_f_lds_bytes(const SimDims &d) {
    const size_t jw = d.nn <= 3 ? 4 : 8;
    return ((static_cast<size_t>(d.N) + 3) / 4 * 4 + static_cast<size_t>(d.N) * jw) * 4;
}

template <int NN> static hipError_t configure_general_nn() {
    for (const void *fn : {reinterpret_cast<const void *>(k_sweep_f<false, NN>),
                           reinterpret_cast<const void *>(k_energy_f<NN>)}) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t configure_general_kernels() {
    hipError_t e = configure_general_nn<1>();
    if (e == hipSuccess) e = configure_general_nn<2>();
    if (e == hipSuccess) e = configure_general_nn<3>();
    if (e == hipSuccess) e = configure_general_nn<4>();
    if (e == hipSuccess) e = configure_general_nn<5>();
    if (e == hipSuccess) e = configure_general_nn<6>();
    return e;
}

hipError_t launch_log_selftest(float *out_ln, float *out_logit, hipStream_t stream) {
    PA_LAUNCH(k_log_selftest, dim3((1u << 24) / 256), dim3(256), 0, stream, out_ln, out_logit);
    return hipGetLastError();
}

hipError_t launch_rng_f(const SimDims &d, const DevPtrs &p, bool gibbs, hipStream_t stream) {
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2;
    const uint32_t groups = (n_pairs_tw + 3) / 4;
    const size_t lds = 64 * 16 * 8 * 4 + 16 * sizeof(double2);
    if (gibbs)
        PA_LAUNCH(k_rng_f<true>, dim3(groups * d.B), dim3(256), lds, stream, d, p);
    else
        PA_LAUNCH(k_rng_f<false>, dim3(groups * d.B), dim3(256), lds, stream, d, p);
    return hipGetLastError();
}

template <int NN>
static void launch_sweep_f_nn(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep, hipStream_t stream) {
    if (do_sweep) {
        const dim3 grid(d.n_real * d.W), block(kSweepFThreads);
        if (d.big)
            PA_LAUNCH((k_sweep_f<true, NN>), grid, block, 0, stream, d, p, 0);
        else
            PA_LAUNCH((k_sweep_f<false, NN>), grid, block, sweep_f_lds_bytes(d), stream, d, p, with_energy ? 1 : 0);
    }
    if (with_energy) {
        if (!do_sweep || d.big) // (the on-chip sweep has written the bond words itself)
            PA_LAUNCH(k_bonds_f<NN>, dim3((d.N + 255) / 256, d.n_real * d.W), dim3(256), 0, stream, d, p);
        if (d.unit && static_cast<uint64_t>(d.N) * NN < (1u << 24)) // every partial sum is an exact f32 integer
            PA_LAUNCH(k_energy_unit_f<NN>, dim3(d.n_real * d.W), dim3(256), 0, stream, d, p);
        else
            PA_LAUNCH(k_energy_f<NN>, dim3(d.n_real * ((d.W + 1) / 2)), dim3(kEnergyThreads),
                               sizeof(EnergyLds<NN>), stream, d, p);
    }
}

hipError_t launch_sweep_f(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                          hipStream_t stream) {
    switch (d.nn) {
    case 1: launch_sweep_f_nn<1>(d, p, with_energy, do_sweep, stream); break;
    case 2: launch_sweep_f_nn<2>(d, p, with_energy, do_sweep, stream); break;
    case 3: launch_sweep_f_nn<3>(d, p, with_energy, do_sweep, stream); break;
    case 4: launch_sweep_f_nn<4>(d, p, with_energy, do_sweep, stream); break;
    case 5: launch_sweep_f_nn<5>(d, p, with_energy, do_sweep, stream); break;
    default: launch_sweep_f_nn<6>(d, p, with_energy, do_sweep, stream); break;
    }
    return hipGetLastError();
}

} // namespace pa
