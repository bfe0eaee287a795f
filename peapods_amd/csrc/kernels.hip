// HIP kernels of the MI355X spin-sim engine (gfx950 only; wave = 64 lanes).
//
//   k_rng        reference-exact random stream -> acceptance-level bit planes
//   k_sweep      level-ordered Metropolis / Gibbs sweep on 32 slots per word (+ energy / magnetisation)
//   k_overlap    replica-overlap and link-overlap dot products
//   k_rng_overlap         both of the above in one launch (overlap hides under the random stream)
//   k_measure    running f64 statistics, P(q) histograms, parallel tempering
//   k_houdayer / k_jorg   overlap-cluster moves, single-cluster variants
//   k_sweep_pipe, k_*_big the same algorithm for lattices that do not fit the LDS
//   k_apply_swaps         materialises deferred tempering swaps
// (kernels_general.hip: arbitrary f32 couplings; kernels_cluster.hip: union-find cluster moves)
//
// Decisions are integer throughout (24-bit draw < table threshold); floating
// point appears only where the reference has it (one f32 division per
// observable, the f32 PT criterion, f64 running sums) and is written with
// explicit IEEE intrinsics so nothing is contracted or reassociated.
#include "engine.hpp"

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "device_util.hpp"
#include "bitcount.hpp"

namespace pa {

// v_writelane_b32: park a wave-uniform (SGPR) value in one lane of a VGPR.  hipcc has no
// builtin for it; the lane select is an inline constant, so no manual wait states apply
// (the ISA's writelane hazard concerns a VALU-written lane-select SGPR only).
template <int LANE> __device__ __forceinline__ uint32_t park_in_lane(uint32_t vec, uint32_t value) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(value), "n"(LANE));
    return vec;
}

// Esum entry -> f32 energy sum: an exact integer on the unit-coupling path, raw f32 bits of the
// reference's sequential sum on the general path.
__device__ __forceinline__ float energy_sum(int32_t raw, uint32_t general) {
    return general ? __int_as_float(raw) : static_cast<float>(raw);
}

// The tempering criterion N (e2 - e1) (1/T1 - 1/T2) >= ln u (tempering.rs:88-91) in the reference's f32 order.
__device__ __forceinline__ bool pt_accept(float n_f, int32_t e_left_raw, int32_t e_right_raw, uint32_t general,
                                          float dbeta, float ln_u) {
    const float e1 = __fdiv_rn(energy_sum(e_left_raw, general), n_f);
    const float e2 = __fdiv_rn(energy_sum(e_right_raw, general), n_f);
    const float delta = __fmul_rn(__fmul_rn(n_f, __fsub_rn(e2, e1)), dbeta);
    return delta >= ln_u;
}

// PRESHIFTED: the thresholds were shifted left by 8 on the host (all of them < 2^24), so the
// high word is compared directly: (top >> 8) < c  <=>  top < (c << 8).
template <int JJ, bool PRESHIFTED>
__device__ __forceinline__ void rng_step(Rng32 &rng, uint32_t ca0, uint32_t ca1, uint32_t ca2,
                                         uint32_t &keep0, uint32_t &keep1) {
    const uint32_t draw = PRESHIFTED ? rng.next_top() : rng.next_top() >> 8;
    // thresholds fall with |dE| (checked on the host): level bit1 = (draw < c1),
    // bit0 = (draw < c0) ^ (draw < c1) ^ (draw < c2)
    const unsigned long long l0 = __ballot(draw < ca0);
    const unsigned long long l1 = __ballot(draw < ca1);
    const unsigned long long l2 = __ballot(draw < ca2);
    const unsigned long long m0 = l0 ^ l1 ^ l2;
    keep0 = park_in_lane<JJ>(keep0, static_cast<uint32_t>(m0));
    keep0 = park_in_lane<JJ + 32>(keep0, static_cast<uint32_t>(m0 >> 32));
    keep1 = park_in_lane<JJ>(keep1, static_cast<uint32_t>(l1));
    keep1 = park_in_lane<JJ + 32>(keep1, static_cast<uint32_t>(l1 >> 32));
}

// The same step on a lattice with zero couplings: sites with an odd number of non-zero bonds take the second
// threshold set.  `par` holds the parity bits of the chunk's 32 sites -- per lane: the two tile words of a wave may
// belong to different realisations -- and dk = ca ^ cb, so a threshold is ca ^ (dk & mask): one v_bfe_i32 and three
// v_bitop3 per step, and the chunk stays unrolled with its immediate lane selects.
template <int JJ, bool PRESHIFTED>
__device__ __forceinline__ void rng_step_par(Rng32 &rng, const uint32_t ca[3], const uint32_t dk[3], uint32_t par,
                                             uint32_t &keep0, uint32_t &keep1) {
    const uint32_t odd = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(par), JJ, 1));
    rng_step<JJ, PRESHIFTED>(rng, ca[0] ^ (dk[0] & odd), ca[1] ^ (dk[1] & odd), ca[2] ^ (dk[2] & odd), keep0, keep1);
}
template <bool PRESHIFTED, int... JJ>
__device__ __forceinline__ void rng_chunk32_par(Rng32 &rng, const uint32_t ca[3], const uint32_t dk[3], uint32_t par,
                                                uint32_t &keep0, uint32_t &keep1, std::integer_sequence<int, JJ...>) {
    (rng_step_par<JJ, PRESHIFTED>(rng, ca, dk, par, keep0, keep1), ...);
}

template <bool PRESHIFTED, int... JJ>
__device__ __forceinline__ void rng_chunk32(Rng32 &rng, uint32_t ca0, uint32_t ca1, uint32_t ca2,
                                            uint32_t &keep0, uint32_t &keep1,
                                            std::integer_sequence<int, JJ...>) {
    (rng_step<JJ, PRESHIFTED>(rng, ca0, ca1, ca2, keep0, keep1), ...);
}

// ---------------------------------------------------------------------------
// The same 32 draws without ballots.  On gfx950 a compare into an SGPR pair and a v_writelane each hold the SIMD
// about 4.2 cycles, v_sub / v_xor / v_bitop3 / v_lshrrev about 2.3 (tools/probe/valu_issue.hip,
// profiles/r02_valu_issue.json), so per draw: draw24 - c has its sign set exactly when draw24 < c (both below 2^25),
// the three signs fold into the two level bits, and one v_alignbit each shifts them into per-lane accumulators
// ((acc << 1) | sign).  After 32 draws lane = slot holds 32 sites; a 32 x 32 bit transpose inside each half wave
// (five butterfly stages: ds_swizzle xor-exchange, rotate, bit-field select) turns that into lane = site, bit = slot
// -- the tile words.  7 slow instructions per draw become 2 slow + 5 fast, + 2 x 5 x 3 per 32 draws.
// ---------------------------------------------------------------------------
struct TransposeLanes {
    uint32_t amt[5], keep[5]; // per stage s = 16, 8, 4, 2, 1: rotate-right amount of the partner's word, bits this lane keeps
    __device__ __forceinline__ explicit TransposeLanes(uint32_t lane_in_half) {
        const uint32_t low[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint32_t s = 16u >> k;
            const bool upper = (lane_in_half & s) != 0;
            amt[k] = upper ? s : 32u - s;
            keep[k] = upper ? ~low[k] : low[k];
        }
    }
};

// rows = lanes of a half wave, columns = bits: returns the transposed matrix's row of this lane
__device__ __forceinline__ uint32_t transpose32(uint32_t x, const TransposeLanes &tl) {
#define PA_SWZ(v, mask) static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), ((mask) << 10) | 0x1F))
    uint32_t q;
    q = PA_SWZ(x, 16); q = __builtin_amdgcn_alignbit(q, q, tl.amt[0]); x = (x & tl.keep[0]) | (q & ~tl.keep[0]);
    q = PA_SWZ(x, 8);  q = __builtin_amdgcn_alignbit(q, q, tl.amt[1]); x = (x & tl.keep[1]) | (q & ~tl.keep[1]);
    q = PA_SWZ(x, 4);  q = __builtin_amdgcn_alignbit(q, q, tl.amt[2]); x = (x & tl.keep[2]) | (q & ~tl.keep[2]);
    q = PA_SWZ(x, 2);  q = __builtin_amdgcn_alignbit(q, q, tl.amt[3]); x = (x & tl.keep[3]) | (q & ~tl.keep[3]);
    q = PA_SWZ(x, 1);  q = __builtin_amdgcn_alignbit(q, q, tl.amt[4]); x = (x & tl.keep[4]) | (q & ~tl.keep[4]);
#undef PA_SWZ
    return x;
}

// c0 >= c1 >= c2: the slot's three UNSHIFTED thresholds (<= 2^24).  On return lane r of a half wave holds the two
// level planes of the site drawn at step 31 - r.
__device__ __forceinline__ void rng_chunk32_signs(Rng32 &rng, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t &keep0,
                                                  uint32_t &keep1, const TransposeLanes &tl) {
    uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const uint32_t draw = rng.next_top() >> 8;
        const uint32_t d0 = draw - c0, d1 = draw - c1, d2 = draw - c2;
        const uint32_t x = __builtin_amdgcn_bitop3_b32(d0, d1, d2, 0x96); // sign = bit 0 of the level
        acc0 = __builtin_amdgcn_alignbit(acc0, x, 31);                    // (acc << 1) | (x >> 31)
        acc1 = __builtin_amdgcn_alignbit(acc1, d1, 31);                   // bit 1 of the level: draw < c1
    }
    keep0 = transpose32(acc0, tl);
    keep1 = transpose32(acc1, tl);
}

// Gibbs: seven thresholds c[0] >= ... >= c[6] (dE = e_max, e_max - 2, ...; checked on the host),
// so the comparisons are nested and the level L = #{j : draw < c[j]} in 0..7 has the bits
//   bit2 = b3,  bit1 = b1 ^ b3 ^ b5,  bit0 = b0 ^ ... ^ b6      (b_j = draw < c[j]).
__device__ __forceinline__ void gibbs_level_ballots(uint32_t draw, const uint32_t c[7],
                                                    unsigned long long &m0, unsigned long long &m1,
                                                    unsigned long long &m2) {
    const unsigned long long b0 = __ballot(draw < c[0]), b1 = __ballot(draw < c[1]);
    const unsigned long long b2 = __ballot(draw < c[2]), b3 = __ballot(draw < c[3]);
    const unsigned long long b4 = __ballot(draw < c[4]), b5 = __ballot(draw < c[5]);
    const unsigned long long b6 = __ballot(draw < c[6]);
    m2 = b3;
    m1 = b1 ^ b3 ^ b5;
    m0 = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6;
}

template <int JJ>
__device__ __forceinline__ void rng_step_gibbs(Rng32 &rng, const uint32_t c[7], uint32_t &keep0,
                                               uint32_t &keep1, uint32_t &keep2) {
    const uint32_t draw = rng.next_top() >> 8;
    unsigned long long m0, m1, m2;
    gibbs_level_ballots(draw, c, m0, m1, m2);
    keep0 = park_in_lane<JJ>(keep0, static_cast<uint32_t>(m0));
    keep0 = park_in_lane<JJ + 32>(keep0, static_cast<uint32_t>(m0 >> 32));
    keep1 = park_in_lane<JJ>(keep1, static_cast<uint32_t>(m1));
    keep1 = park_in_lane<JJ + 32>(keep1, static_cast<uint32_t>(m1 >> 32));
    keep2 = park_in_lane<JJ>(keep2, static_cast<uint32_t>(m2));
    keep2 = park_in_lane<JJ + 32>(keep2, static_cast<uint32_t>(m2 >> 32));
}

template <int JJ>
__device__ __forceinline__ void rng_step_gibbs_par(Rng32 &rng, const uint32_t ce[7], const uint32_t dk[7], uint32_t par,
                                                   uint32_t &keep0, uint32_t &keep1, uint32_t &keep2) {
    const uint32_t odd = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(par), JJ, 1));
    uint32_t c[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) c[j] = ce[j] ^ (dk[j] & odd);
    rng_step_gibbs<JJ>(rng, c, keep0, keep1, keep2);
}
template <int... JJ>
__device__ __forceinline__ void rng_chunk32_gibbs_par(Rng32 &rng, const uint32_t ce[7], const uint32_t dk[7], uint32_t par,
                                                      uint32_t &keep0, uint32_t &keep1, uint32_t &keep2,
                                                      std::integer_sequence<int, JJ...>) {
    (rng_step_gibbs_par<JJ>(rng, ce, dk, par, keep0, keep1, keep2), ...);
}

template <int... JJ>
__device__ __forceinline__ void rng_chunk32_gibbs(Rng32 &rng, const uint32_t c[7], uint32_t &keep0,
                                                  uint32_t &keep1, uint32_t &keep2,
                                                  std::integer_sequence<int, JJ...>) {
    (rng_step_gibbs<JJ>(rng, c, keep0, keep1, keep2), ...);
}

// ---------------------------------------------------------------------------
// k_rng: workgroup = 4 waves that share ONE site block b (so one jump table,
// staged in LDS); each wave serves two tile words: lanes 0-31 the 32 slots of
// tile word 2k, lanes 32-63 those of tile word 2k+1.  Every lane walks the
// block's sites in order, drawing the reference's next 24-bit value for its
// slot (sweep.rs:181) and comparing it with the slot's three thresholds; the
// wave ballots of the comparisons ARE the 32-slot words of that site.  Ballot
// words are parked in lane (site & 31) with v_writelane and stored 32 sites at
// a time, fully coalesced.
// ---------------------------------------------------------------------------
// GIBBS: seven thresholds per parity class and a third level plane (planes_hi); thresholds are
// compared unshifted (a Gibbs threshold can be 2^24 = "every draw").
// REPLAY: the tempering event of the previous sweep has not been applied to the cursors (it runs as a measurement
// block of this very launch, PtParams::relay).  Every workgroup therefore replays the swap decisions of the
// replica ladders its tile words belong to (one lane per ladder: the generator of system r * n_temps as the
// previous sweep left it, gen_range for the edge, the f32 criterion -- the same code the event runs) and each lane
// then fetches the state of the system that sits in its slot AFTER the event: its own slot's, or the neighbouring
// slot's when its edge was accepted; the lane that receives the ladder's generator skips the draws the event used.
template <bool PRESHIFTED, bool GIBBS, int REPLAY = 0>
__device__ __forceinline__ void rng_block(const SimDims &d, const DevPtrs &p, uint32_t bid, uint32_t *tab,
                                          uint32_t first_parity = 0) {
    // tab: [64 * 16 * 8] dwords of LDS for the block's jump table, then the replay scratch
    const uint32_t block = bid % d.B;
    const uint32_t group = bid / d.B;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nthreads = blockDim.x, nw = nthreads >> 6; // 4 waves (k_rng, k_rng_relay)
    const uint32_t bit = lane & 31u, half = lane >> 5;
    const uint32_t n_tw = d.n_real * d.W;
    const uint32_t tw = (group * nw + wave) * 2 + half;
    const bool tw_ok = tw < n_tw;
    // (small indices are divided through host-made reciprocals: a runtime division is ~25 VALU instructions, and the
    // prologue of this block had eight of them in front of every 128-site stretch of the stream)
    const uint32_t real = tw_ok ? div_small(tw, d.inv_W) : 0, w = tw_ok ? tw - real * d.W : 0;
    const uint32_t slot = w * 32 + bit;
    const bool active = tw_ok && slot < d.n_sys;
    const uint32_t slot_rep = div_small(slot, d.inv_temps), slot_t = slot - slot_rep * d.n_temps; // slot = rep * n_temps + t

    // the block's jump table: every 16-byte piece is requested before the first one is stored (copy_to_lds), so the
    // workgroup pays one memory round trip for its 32 KB, not eight
    if (block > 0) copy_to_lds(tab, p.jump + static_cast<size_t>(block) * (64 * 16 * 8), 64 * 16 * 8, tid, nthreads);
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t src_slot = slot, skip_draws = 0;
    if (REPLAY == 2) {
        // full_ladder (tempering.rs:59-70): the ladder's generator draws its n_temps - 1 uniforms in order, the edges of
        // the first parity are decided side by side, then those of the other parity on the exchanged energies -- as in
        // measure_block, which runs the same event in this launch to record it.  A slot's generator is found by walking
        // the two rounds backwards.  Scratch behind the jump table: draws[tasks][n_edges], then two accepted-edge masks
        // per task (n_temps <= 32, checked on the host).
        const uint32_t n_edges = d.n_temps - 1;
        const uint32_t tw0 = group * 2 * nw, real0 = div_small(tw0, d.inv_W);
        const uint32_t real1 = div_small(min(tw0 + 2 * nw - 1u, n_tw - 1u), d.inv_W);
        const uint32_t n_tasks = (real1 - real0 + 1u) * d.n_replicas;
        uint32_t *draws = tab + 64 * 16 * 8;
        uint32_t *dec = draws + n_tasks * n_edges;
        if (tid < n_tasks) {
            const uint32_t task_real = div_small(tid, d.inv_replicas);
            const uint32_t r = real0 + task_real, offset = (tid - task_real * d.n_replicas) * d.n_temps;
            uint32_t gs[8];
            load_cursor(p.cursor, r, d.SP, p.slot_of[static_cast<size_t>(r) * d.n_sys + offset], gs);
            DevRng g;
            g.s0 = (static_cast<uint64_t>(gs[1]) << 32) | gs[0];
            g.s1 = (static_cast<uint64_t>(gs[3]) << 32) | gs[2];
            g.s2 = (static_cast<uint64_t>(gs[5]) << 32) | gs[4];
            g.s3 = (static_cast<uint64_t>(gs[7]) << 32) | gs[6];
            for (uint32_t a = 0; a < n_edges; ++a) draws[tid * n_edges + a] = static_cast<uint32_t>(g.next_lane() >> 40);
            dec[2 * tid] = 0;
            dec[2 * tid + 1] = 0;
        }
        __syncthreads();
        // one (ladder, draw) item per lane and turn, at most four turns (checked on the host): the item's edge, its four
        // candidate energies, its coupling difference and its logarithm -- every global value of both rounds is
        // requested at once, so the prologue pays one memory round trip, not one per round
        const uint32_t n_first = (n_edges + 1 - first_parity) / 2; // edges of the first parity come first
        const float n_f = static_cast<float>(d.N);
        const uint32_t n_items = n_tasks * n_edges;
        uint32_t it_edge[4], it_task[4];
        int32_t e0[4], e1[4], em[4], ep[4];
        float it_db[4], it_ln[4];
        bool it_first[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t idx = tid + k * nthreads;
            it_task[k] = 0xFFFFFFFFu;
            if (idx < n_items) {
                const uint32_t task = idx / n_edges, a = idx % n_edges;
                it_first[k] = a < n_first;
                const uint32_t edge = it_first[k] ? first_parity + 2 * a : (1 - first_parity) + 2 * (a - n_first);
                const uint32_t task_real = div_small(task, d.inv_replicas);
                const int32_t *E = p.Esum + static_cast<size_t>(real0 + task_real) * d.n_sys + (task - task_real * d.n_replicas) * d.n_temps;
                it_task[k] = task;
                it_edge[k] = edge;
                e0[k] = E[edge];
                e1[k] = E[edge + 1];
                em[k] = edge > 0 ? E[edge - 1] : 0;           // (read for round 1 only)
                ep[k] = edge + 2 < d.n_temps ? E[edge + 2] : 0;
                it_db[k] = p.dbeta[edge];
                it_ln[k] = p.ln_table[draws[idx]];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) // round 0
            if (it_task[k] != 0xFFFFFFFFu && it_first[k] && pt_accept(n_f, e0[k], e1[k], d.general, it_db[k], it_ln[k]))
                atomicOr(&dec[2 * it_task[k]], 1u << it_edge[k]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) // round 1, on the exchanged energies
            if (it_task[k] != 0xFFFFFFFFu && !it_first[k]) {
                const uint32_t edge = it_edge[k], m0 = dec[2 * it_task[k]];
                const int32_t el = (edge > 0 && ((m0 >> (edge - 1)) & 1u)) ? em[k] : e0[k];
                const int32_t er = ((m0 >> (edge + 1)) & 1u) ? ep[k] : e1[k];
                if (pt_accept(n_f, el, er, d.general, it_db[k], it_ln[k])) atomicOr(&dec[2 * it_task[k] + 1], 1u << edge);
            }
        __syncthreads();
        if (active) {
            const uint32_t rep = slot_rep, t = slot_t;
            const uint32_t task = (real - real0) * d.n_replicas + rep;
            const uint32_t m0 = dec[2 * task], m1 = dec[2 * task + 1];
            uint32_t s1 = t; // where the system now at temperature t sat between the rounds, then before the event
            if ((m1 >> t) & 1u) s1 = t + 1;
            else if (t > 0 && ((m1 >> (t - 1)) & 1u)) s1 = t - 1;
            uint32_t s0 = s1;
            if ((m0 >> s1) & 1u) s0 = s1 + 1;
            else if (s1 > 0 && ((m0 >> (s1 - 1)) & 1u)) s0 = s1 - 1;
            src_slot = rep * d.n_temps + s0;
            if (p.system_ids[static_cast<size_t>(real) * d.n_sys + src_slot] == rep * d.n_temps) skip_draws = n_edges;
        }
    }
    if (REPLAY == 1) {
        uint32_t *dec = tab + 64 * 16 * 8; // [<= 256] edge | accepted << 16 | draws << 17 per (realisation, ladder)
        const uint32_t tw0 = group * 2 * nw, real0 = div_small(tw0, d.inv_W);
        const uint32_t real1 = div_small(min(tw0 + 2 * nw - 1u, n_tw - 1u), d.inv_W);
        const uint32_t n_tasks = (real1 - real0 + 1u) * d.n_replicas; // (<= 256: rng_relay_supported)
        if (tid < n_tasks) {
            const uint32_t task_real = div_small(tid, d.inv_replicas);
            const uint32_t r = real0 + task_real, offset = (tid - task_real * d.n_replicas) * d.n_temps;
            uint32_t gs[8];
            load_cursor(p.cursor, r, d.SP, p.slot_of[static_cast<size_t>(r) * d.n_sys + offset], gs);
            DevRng g;
            g.s0 = (static_cast<uint64_t>(gs[1]) << 32) | gs[0];
            g.s1 = (static_cast<uint64_t>(gs[3]) << 32) | gs[2];
            g.s2 = (static_cast<uint64_t>(gs[5]) << 32) | gs[4];
            g.s3 = (static_cast<uint64_t>(gs[7]) << 32) | gs[6];
            // rand 0.8.5 gen_range(0..n_temps-1) on usize (tempering.rs:33), then the uniform draw (:89)
            const uint64_t range = d.n_temps - 1;
            const uint64_t zone = (range << __clzll(range)) - 1;
            uint32_t n_draws = 0, edge;
            for (;;) {
                const uint64_t v = g.next_lane();
                ++n_draws;
                if (v * range <= zone) {
                    edge = static_cast<uint32_t>(__umul64hi(v, range));
                    break;
                }
            }
            const uint32_t draw = static_cast<uint32_t>(g.next_lane() >> 40);
            ++n_draws;
            const int32_t *E = p.Esum + static_cast<size_t>(r) * d.n_sys + offset + edge;
            const bool accepted = pt_accept(static_cast<float>(d.N), E[0], E[1], d.general, p.dbeta[edge], p.ln_table[draw]);
            dec[tid] = edge | (accepted ? 1u << 16 : 0u) | (n_draws << 17);
        }
        __syncthreads();
        if (active) {
            const uint32_t rep = slot_rep, t = slot_t;
            const uint32_t v = dec[(real - real0) * d.n_replicas + rep];
            const uint32_t edge = v & 0xFFFFu;
            if ((v >> 16) & 1u) {
                if (t == edge) src_slot = slot + 1;
                else if (t == edge + 1) src_slot = slot - 1;
            }
            if (p.system_ids[static_cast<size_t>(real) * d.n_sys + src_slot] == rep * d.n_temps) skip_draws = v >> 17;
        }
    }
    if (active) load_cursor(p.cursor, real, d.SP, src_slot, st);
    if (REPLAY != 0 && skip_draws) { // the ladder's generator: the event's draws come before this sweep's
        DevRng g;
        g.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
        g.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
        g.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
        g.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
        for (uint32_t k = 0; k < skip_draws; ++k) g.advance();
        st[0] = static_cast<uint32_t>(g.s0); st[1] = static_cast<uint32_t>(g.s0 >> 32);
        st[2] = static_cast<uint32_t>(g.s1); st[3] = static_cast<uint32_t>(g.s1 >> 32);
        st[4] = static_cast<uint32_t>(g.s2); st[5] = static_cast<uint32_t>(g.s2 >> 32);
        st[6] = static_cast<uint32_t>(g.s3); st[7] = static_cast<uint32_t>(g.s3 >> 32);
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((group * nw + wave) * 2) >= n_tw) return; // idle wave
    if (block > 0) jump_state_lds(st, tab);
    Rng32 rng;
    rng.load(st);

    const uint32_t begin = block * d.BS;
    const uint32_t len = min(d.BS, d.N - begin); // uniform over the workgroup
    uint2 *planes = p.planes + static_cast<size_t>(tw_ok ? tw : 0) * d.planes_stride;
    const uint32_t *mpar = p.mpar + static_cast<size_t>(real) * d.MW;
    // where a site's acceptance planes go: by site, or (k_sweep_slab) at its padded schedule position
    auto filed = [&](uint32_t site) -> uint32_t {
        return d.planes_perm ? p.reg_pos[site] : (d.pipe_lds ? p.pos_of[site] : site);
    };

    if (GIBBS) {
        uint32_t ce[7] = {0, 0, 0, 0, 0, 0, 0}, co[7] = {0, 0, 0, 0, 0, 0, 0};
        if (active) {
            const uint32_t t = slot_t;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                ce[j] = p.thr_gibbs[(0 + j) * d.n_temps + t];
                co[j] = p.thr_gibbs[(7 + j) * d.n_temps + t];
            }
        }
        uint32_t *planes_hi = p.planes_hi + static_cast<size_t>(tw_ok ? tw : 0) * d.planes_stride;
        for (uint32_t j0 = 0; j0 < len; j0 += 32) {
            uint32_t keep0 = 0, keep1 = 0, keep2 = 0;
            const uint32_t chunk = min(32u, len - j0);
            if (chunk == 32 && !d.has_odd) {
                rng_chunk32_gibbs(rng, ce, keep0, keep1, keep2, std::make_integer_sequence<int, 32>{});
            } else if (chunk == 32 && ((begin + j0) & 31u) == 0) { // one parity word covers the chunk
                const uint32_t par = mpar[(begin + j0) >> 5]; // (the lane's realisation)
                uint32_t dk[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) dk[j] = ce[j] ^ co[j];
                rng_chunk32_gibbs_par(rng, ce, dk, par, keep0, keep1, keep2, std::make_integer_sequence<int, 32>{});
            } else {
                for (uint32_t jj = 0; jj < chunk; ++jj) {
                    const uint32_t site = begin + j0 + jj;
                    const uint32_t draw = rng.next_top() >> 8;
                    const bool odd = d.has_odd && ((mpar[site >> 5] >> (site & 31u)) & 1u);
                    uint32_t c[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) c[j] = odd ? co[j] : ce[j];
                    unsigned long long m0, m1, m2;
                    gibbs_level_ballots(draw, c, m0, m1, m2);
                    if (bit == jj) {
                        keep0 = half ? static_cast<uint32_t>(m0 >> 32) : static_cast<uint32_t>(m0);
                        keep1 = half ? static_cast<uint32_t>(m1 >> 32) : static_cast<uint32_t>(m1);
                        keep2 = half ? static_cast<uint32_t>(m2 >> 32) : static_cast<uint32_t>(m2);
                    }
                }
            }
            if (tw_ok && bit < chunk) {
                const uint32_t at = filed(begin + j0 + bit);
                planes[at] = make_uint2(keep0, keep1);
                planes_hi[at] = keep2;
            }
        }
        if (active && block == d.B - 1) {
            rng.store(st);
            store_cursor(p.cursor_out, real, d.SP, slot, st);
        }
        return;
    }

    uint32_t ca0 = 0, ca1 = 0, ca2 = 0, cb0 = 0, cb1 = 0, cb2 = 0;
    if (active) {
        const uint32_t t = slot_t;
        const uint32_t *thr = PRESHIFTED ? p.thr + 6 * d.n_temps : p.thr; // second half: c << 8
        ca0 = thr[0 * d.n_temps + t];
        ca1 = thr[1 * d.n_temps + t];
        ca2 = thr[2 * d.n_temps + t];
        cb0 = thr[3 * d.n_temps + t];
        cb1 = thr[4 * d.n_temps + t];
        cb2 = thr[5 * d.n_temps + t];
    }

    // Measured on MI355X (bench workload): the sign form below is bit-equal but SLOWER, k_rng 51 -> 61 us per
    // launch, although it trades five 4.2-cycle instructions for 2.3-cycle ones by the isolated-stream numbers of
    // tools/probe/valu_issue.hip -- in this mix every instruction of the sign form costs ~4.2 cycles, while the
    // compares-into-SGPR and v_writelane of the ballot form overlap with other waves' work.  Kept as the
    // alternative (flip the constant), not used.
    constexpr bool kBallots = true;
    const TransposeLanes tl(bit);
    uint32_t ua0 = 0, ua1 = 0, ua2 = 0; // the even set again, unshifted, for the sign form
    if (active) {
        const uint32_t t = slot_t;
        ua0 = p.thr[0 * d.n_temps + t];
        ua1 = p.thr[1 * d.n_temps + t];
        ua2 = p.thr[2 * d.n_temps + t];
    }
    for (uint32_t j0 = 0; j0 < len; j0 += 32) {
        uint32_t keep0 = 0, keep1 = 0;
        const uint32_t chunk = min(32u, len - j0);
        if (chunk == 32 && !d.has_odd && !kBallots) {
            rng_chunk32_signs(rng, ua0, ua1, ua2, keep0, keep1, tl);
            // lane r holds the site of step 31 - r: the store below is the same 256 coalesced bytes, mirrored
            if (tw_ok) planes[filed(begin + j0 + (31u - bit))] = make_uint2(keep0, keep1);
            continue;
        }
        if (chunk == 32 && !d.has_odd) {
            rng_chunk32<PRESHIFTED>(rng, ca0, ca1, ca2, keep0, keep1,
                                    std::make_integer_sequence<int, 32>{});
        } else if (chunk == 32 && ((begin + j0) & 31u) == 0) { // one parity word covers the chunk
            const uint32_t par = mpar[(begin + j0) >> 5]; // (the lane's realisation)
            const uint32_t ca[3] = {ca0, ca1, ca2}, dk[3] = {ca0 ^ cb0, ca1 ^ cb1, ca2 ^ cb2};
            rng_chunk32_par<PRESHIFTED>(rng, ca, dk, par, keep0, keep1, std::make_integer_sequence<int, 32>{});
        } else {
            for (uint32_t jj = 0; jj < chunk; ++jj) {
                const uint32_t site = begin + j0 + jj;
                const uint32_t draw = PRESHIFTED ? rng.next_top() : rng.next_top() >> 8;
                uint32_t c0 = ca0, c1 = ca1, c2 = ca2;
                if (d.has_odd && ((mpar[site >> 5] >> (site & 31u)) & 1u)) {
                    c0 = cb0;
                    c1 = cb1;
                    c2 = cb2;
                }
                const unsigned long long l0 = __ballot(draw < c0);
                const unsigned long long l1 = __ballot(draw < c1);
                const unsigned long long l2 = __ballot(draw < c2);
                const unsigned long long m0 = l0 ^ l1 ^ l2;
                const uint32_t x = half ? static_cast<uint32_t>(m0 >> 32) : static_cast<uint32_t>(m0);
                const uint32_t y = half ? static_cast<uint32_t>(l1 >> 32) : static_cast<uint32_t>(l1);
                if (bit == jj) {
                    keep0 = x;
                    keep1 = y;
                }
            }
        }
        if (tw_ok && bit < chunk) planes[filed(begin + j0 + bit)] = make_uint2(keep0, keep1);
    }
    // The last block ends exactly N draws after the cursor: that is the next cursor.
    if (active && block == d.B - 1) {
        rng.store(st);
        store_cursor(p.cursor_out, real, d.SP, slot, st);
    }
}

template <bool PRESHIFTED, bool GIBBS>
__global__ __launch_bounds__(256) void k_rng(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds_dyn[];
    rng_block<PRESHIFTED, GIBBS>(d, p, blockIdx.x, lds_dyn);
}

// ---------------------------------------------------------------------------
// Bit-sliced helpers
// ---------------------------------------------------------------------------

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

// Gibbs accept mask (heat-bath rule on the 2^24 grid, sweep.rs:262-284): L = number of the
// parity class's thresholds the draw is below (3 planes), counted from dE = e_max = 2 z' (even m)
// or 2 z' - 1 (odd m) downwards, so the accepted energy changes are the top L values and
//   flip  <=>  2U - m >= e_max - 2 (L - 1)  <=>  U + L >= z' + 1 + floor(m / 2) =: h.
// The comparison is the carry out of the 4-bit addition (U + L) + (16 - h).
__device__ __forceinline__ uint32_t accept_mask_gibbs(uint32_t s, const uint32_t nb[6], uint32_t jm,
                                                      uint32_t nn, uint32_t l0, uint32_t l1,
                                                      uint32_t l2) {
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
    const uint32_t t2 = U2 ^ l2 ^ k1, t3 = maj3(U2, l2, k1);
    const uint32_t m = __popc((jm >> 6) & 63u);
    const uint32_t K = 16u - (nn + 1u + (m >> 1)); // 16 - h, h in 2..7
    const uint32_t K0 = 0u - (K & 1u), K1 = 0u - ((K >> 1) & 1u), K2 = 0u - ((K >> 2) & 1u),
                   K3 = 0u - ((K >> 3) & 1u);
    const uint32_t a0 = t0 & K0;
    const uint32_t a1 = maj3(t1, K1, a0);
    const uint32_t a2 = maj3(t2, K2, a1);
    return maj3(t3, K3, a2);
}

// Accept mask when every one of the 2*NN bonds of every site is non-zero (+-J and ferromagnets):
// m = 2 NN, h = NN, no non-zero masks, no per-site case split.
template <int NN>
__device__ __forceinline__ uint32_t accept_mask_dense(uint32_t s, const uint32_t nb[6], uint32_t jm,
                                                      uint32_t l0, uint32_t l1) {
    uint32_t u[6];
#pragma unroll
    for (int k = 0; k < 2 * NN; ++k)
        u[k] = s ^ nb[k] ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(jm), k, 1));
    uint32_t U0, U1, U2;
    if (NN == 3) {
        const uint32_t s1 = u[0] ^ u[1] ^ u[2], c1 = maj3(u[0], u[1], u[2]);
        const uint32_t s2 = u[3] ^ u[4] ^ u[5], c2 = maj3(u[3], u[4], u[5]);
        const uint32_t c3 = s1 & s2;
        U0 = s1 ^ s2;
        U1 = c1 ^ c2 ^ c3;
        U2 = maj3(c1, c2, c3);
    } else if (NN == 2) {
        const uint32_t s1 = u[0] ^ u[1] ^ u[2], c1 = maj3(u[0], u[1], u[2]);
        const uint32_t c2 = s1 & u[3];
        U0 = s1 ^ u[3];
        U1 = c1 ^ c2;
        U2 = c1 & c2;
    } else {
        U0 = u[0] ^ u[1];
        U1 = u[0] & u[1];
        U2 = 0;
    }
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t hi = U2 | k1; // t2 | t3 = (U2 ^ k1) | (U2 & k1)
    if (NN == 3) return (t0 & t1) | hi;
    if (NN == 2) return t1 | hi;
    return t0 | t1 | hi;
}

// Heat-bath accept mask (accept_mask_gibbs) when every one of the 2*NN bonds is non-zero; nb[k] already carries the
// coupling sign (neighbour word ^ sign mask).  K = 16 - h = 15 - 2 NN is a compile-time constant here.
template <int NN>
__device__ __forceinline__ uint32_t accept_mask_gibbs_dense(uint32_t s, const uint32_t nb[6], uint32_t l0, uint32_t l1, uint32_t l2) {
    uint32_t u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = k < 2 * NN ? (s ^ nb[k]) : 0u;
    const uint32_t s1 = u[0] ^ u[1] ^ u[2], c1 = maj3(u[0], u[1], u[2]);
    const uint32_t s2 = u[3] ^ u[4] ^ u[5], c2 = maj3(u[3], u[4], u[5]);
    const uint32_t U0 = s1 ^ s2, c3 = s1 & s2;
    const uint32_t U1 = c1 ^ c2 ^ c3, U2 = maj3(c1, c2, c3);
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t t2 = U2 ^ l2 ^ k1, t3 = maj3(U2, l2, k1);
    constexpr uint32_t K = 15u - 2u * NN;
    constexpr uint32_t K0 = 0u - (K & 1u), K1 = 0u - ((K >> 1) & 1u), K2 = 0u - ((K >> 2) & 1u), K3 = 0u - ((K >> 3) & 1u);
    const uint32_t a0 = t0 & K0;
    const uint32_t a1 = maj3(t1, K1, a0);
    const uint32_t a2 = maj3(t2, K2, a1);
    return maj3(t3, K3, a2);
}

// The two rules on the words u[k] = (s ^ neighbour ^ sign) & non-zero, for sites whose bonds may be zero (the table
// path of diluted lattices): h0 / h1 are the bits of h = ceil(m / 2) as masks (accept_mask), K0..K3 the bits of
// 16 - (z' + 1 + floor(m / 2)) (accept_mask_gibbs), both prepared off the chain.
__device__ __forceinline__ void unsat_count(const uint32_t u[6], uint32_t &U0, uint32_t &U1, uint32_t &U2) {
    const uint32_t s1 = u[0] ^ u[1] ^ u[2], c1 = maj3(u[0], u[1], u[2]);
    const uint32_t s2 = u[3] ^ u[4] ^ u[5], c2 = maj3(u[3], u[4], u[5]);
    const uint32_t c3 = s1 & s2;
    U0 = s1 ^ s2;
    U1 = c1 ^ c2 ^ c3;
    U2 = maj3(c1, c2, c3);
}
__device__ __forceinline__ uint32_t accept_u_metropolis(const uint32_t u[6], uint32_t l0, uint32_t l1, uint32_t h0, uint32_t h1) {
    uint32_t U0, U1, U2;
    unsat_count(u, U0, U1, U2);
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t hi = U2 | k1;
    const uint32_t ge1 = t0 | t1 | hi, ge2 = t1 | hi, ge3 = (t0 & t1) | hi;
    const uint32_t lo_sel = ge1 | ~h0;                 // h = 0: everything, h = 1: ge1
    const uint32_t hi_sel = (ge3 & h0) | (ge2 & ~h0);  // h = 2: ge2, h = 3: ge3
    return (hi_sel & h1) | (lo_sel & ~h1);
}
__device__ __forceinline__ uint32_t accept_u_gibbs(const uint32_t u[6], uint32_t l0, uint32_t l1, uint32_t l2, uint32_t K0,
                                                   uint32_t K1, uint32_t K2, uint32_t K3) {
    uint32_t U0, U1, U2;
    unsat_count(u, U0, U1, U2);
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t t2 = U2 ^ l2 ^ k1, t3 = maj3(U2, l2, k1);
    const uint32_t a0 = t0 & K0;
    const uint32_t a1 = maj3(t1, K1, a0);
    const uint32_t a2 = maj3(t2, K2, a1);
    return maj3(t3, K3, a2);
}

// Two arrays in one go for a 256-thread workgroup: all loads of both copies are issued before the
// first LDS store when everything is 16-byte aligned and fits 16 vector loads per thread.
__device__ __forceinline__ void copy2_to_lds(uint32_t *dst_a, const uint32_t *__restrict__ src_a,
                                             uint32_t n_a, uint32_t *dst_b,
                                             const uint32_t *__restrict__ src_b, uint32_t n_b,
                                             uint32_t tid) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(dst_a) |
                           reinterpret_cast<uintptr_t>(src_b) | reinterpret_cast<uintptr_t>(dst_b)) & 15u) == 0 &&
                         ((n_a | n_b) & 3u) == 0;
    const uint32_t qa = n_a >> 2, qb = n_b >> 2; // uint4 counts
    if (aligned && qa <= 4 * 256 && qb <= 12 * 256) {
        const uint4 *a4 = reinterpret_cast<const uint4 *>(src_a), *b4 = reinterpret_cast<const uint4 *>(src_b);
        uint4 va[4], vb[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * 256 + tid;
            va[k] = i < qa ? a4[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t i = k * 256 + tid;
            vb[k] = i < qb ? b4[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * 256 + tid;
            if (i < qa) reinterpret_cast<uint4 *>(dst_a)[i] = va[k];
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t i = k * 256 + tid;
            if (i < qb) reinterpret_cast<uint4 *>(dst_b)[i] = vb[k];
        }
        return;
    }
    copy_to_lds(dst_a, src_a, n_a, tid, 256);
    copy_to_lds(dst_b, src_b, n_b, tid, 256);
}

// ---------------------------------------------------------------------------
// Deferred tempering swaps.  An accepted swap of slots (k, k+1) exchanges two adjacent bit
// columns of the tile.  masks[round][w] has bit k%32 set for every accepted pair whose lower slot
// is k = 32 w + bit; round 0 pairs are applied before round 1 pairs (full ladder: the two
// parities; single edge: round 0 only).  A pair may straddle two words (bits 31 | 0).
// ---------------------------------------------------------------------------
// Materialise pending swaps in place (any lattice size): one thread per site handles all W words.
__global__ void k_apply_swaps(SimDims d, DevPtrs p) {
    const uint32_t real = blockIdx.y;
    const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
    if (site >= d.N) return;
    if (p.swap_epoch[real] == p.applied_epoch[static_cast<size_t>(real) * d.W]) return;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * d.W * d.N;
    const uint32_t *masks = p.swap_masks + static_cast<size_t>(real) * 2 * d.W;
    for (int round = 0; round < 2; ++round) {
        uint32_t carry_t = 0; // bit 31 of the previous word's t
        uint32_t v = Sg[site];
        for (uint32_t w = 0; w < d.W; ++w) {
            const uint32_t next = (w + 1 < d.W) ? Sg[static_cast<size_t>(w + 1) * d.N + site] : 0u;
            const uint32_t t = (v ^ ((v >> 1) | (next << 31))) & masks[round * d.W + w];
            Sg[static_cast<size_t>(w) * d.N + site] = v ^ t ^ ((t << 1) | carry_t);
            carry_t = t >> 31;
            v = next;
        }
    }
}

__global__ void k_mark_applied(SimDims d, DevPtrs p) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < d.n_real * d.W) p.applied_epoch[idx] = p.swap_epoch[idx / d.W];
}

// ---------------------------------------------------------------------------
// k_sweep: workgroup = (tile, word).  The word's N site values and the two
// acceptance-level planes live in LDS; the schedule passes (dependency levels
// cut into chunks of <= 256 sites) run in order with one LDS barrier each, which
// reproduces the reference's sequential site order (sweep.rs:58-61) bit for
// bit.  The next pass's site record is prefetched while the current one runs.
// MODE 1..3: every bond non-zero with MODE forward directions (dense path);
// MODE 0: couplings may contain zeros (general path).
// Epilogue: energy sum_{i,d} J s_i s_fwd and magnetisation sum_i s_i for the
// word's 32 slots (energy.rs:92-109), by byte-sliced vertical counters.
// ---------------------------------------------------------------------------
#define PA_STAMP(k)                                                                                \
    if (STAMPS && blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = clock64();

// GIBBS: heat-bath rule; a third level plane (planes_hi) sits behind the two Metropolis ones.
template <bool STAMPS, int MODE, bool GIBBS = false>
__global__ __launch_bounds__(256) void k_sweep(SimDims d, DevPtrs p, int with_energy,
                                               int do_sweep, unsigned long long *stamps) {
    extern __shared__ uint32_t lds[];
    PA_STAMP(0)
    const uint32_t n1 = d.N + 1;                               // + the dummy slot
    const uint32_t n_pad = (n1 + 1u) & ~1u;
    uint32_t *sw = lds;                                        // [N + 1] spins of this word
    uint32_t *pl = lds + n_pad;                                // [N + 1][2] acceptance-level planes
    uint32_t *pl2 = lds + n_pad + 2 * static_cast<size_t>(n1); // [N + 1] third plane (Gibbs only)
    uint32_t *scratch = lds + n_pad + (GIBBS ? 3 : 2) * static_cast<size_t>(n1); // [16 rows][48]
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t *Sin = p.S + (static_cast<size_t>(real) * d.W + w) * d.N;  // this word's N sites
    uint32_t *Sout = p.S_next + (static_cast<size_t>(real) * d.W + w) * d.N; // double buffer
    const uint4 *rec = p.rec + static_cast<size_t>(real) * n1;
    // the few words that describe pending tempering swaps are requested first, so that their
    // latency hides behind the bulk copy below
    const uint32_t epoch = p.swap_epoch[real];
    const uint32_t applied = p.applied_epoch[static_cast<size_t>(real) * d.W + w];
    uint32_t mp[2], mc[2], mn[2];
    {
        const uint32_t *masks = p.swap_masks + static_cast<size_t>(real) * 2 * d.W;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mp[r] = w > 0 ? masks[r * d.W + w - 1] : 0u;
            mc[r] = masks[r * d.W + w];
            mn[r] = w + 1 < d.W ? masks[r * d.W + w + 1] : 0u;
        }
    }
    // spins and acceptance planes are requested together (one memory round trip, not two)
    const uint32_t *planes_in = reinterpret_cast<const uint32_t *>(
        p.planes + (static_cast<size_t>(real) * d.W + w) * d.N);
    copy2_to_lds(sw, Sin, d.N, pl, planes_in, do_sweep ? 2 * d.N : 0, tid);
    if (GIBBS && do_sweep)
        copy_to_lds(pl2, p.planes_hi + (static_cast<size_t>(real) * d.W + w) * d.N, d.N, tid, 256);
    if (do_sweep && epoch != applied) {
        // tempering swaps accepted after the previous sweep: exchange the bit columns now
        const bool window = (((mp[0] | mp[1] | mc[0] | mc[1]) >> 31) != 0); // a pair straddles words
        __syncthreads();
        for (uint32_t i = tid; i < d.N; i += 256) {
            const uint32_t v = sw[i];
            uint32_t prev = 0, next = 0;
            if (window) { // old neighbouring words (the input buffer is not written in this launch)
                if (w > 0) prev = Sin[static_cast<ptrdiff_t>(i) - static_cast<ptrdiff_t>(d.N)];
                if (w + 1 < d.W) next = Sin[static_cast<size_t>(i) + d.N];
            }
            sw[i] = apply_swaps_window(prev, v, next, mp, mc, mn);
        }
        if (tid == 0) p.applied_epoch[static_cast<size_t>(real) * d.W + w] = epoch;
    }
    if (tid == 0) {
        sw[d.N] = 0;
        pl[2 * d.N] = 0;
        pl[2 * d.N + 1] = 0;
        if (GIBBS) pl2[d.N] = 0;
    }
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    __syncthreads();
    PA_STAMP(1)

    if (do_sweep) {
        // Site records stream from global memory three passes ahead of their use (four rotating
        // register buffers, loop unrolled by four so no buffer is ever copied).
        const int dbg = STAMPS ? (with_energy >> 8) : 0; // diagnostic build only (PEAPODS_DBG)
        // pass_start[] lives in two VGPRs (lane l holds entries l and 64 + l); a pass boundary is
        // then one v_readlane away instead of a scalar-cache round trip in front of every fetch.
        const uint32_t ps_lo = p.pass_start[min(tid & 63u, d.n_passes)];
        const uint32_t ps_hi = p.pass_start[min(64u + (tid & 63u), d.n_passes)];
        auto boundary = [&](uint32_t k) -> uint32_t {
            if (k < 64) return __builtin_amdgcn_readlane(ps_lo, k);
            if (k < 128) return __builtin_amdgcn_readlane(ps_hi, k - 64);
            return p.pass_start[k];
        };
        auto fetch = [&](uint32_t pass) -> uint4 {
            if (pass >= d.n_passes) return make_uint4(0, 0, 0, 0);
            if (dbg & 2) return make_uint4(tid * 4, tid * 4, tid * 4, tid * 4);
            const uint32_t pos = boundary(pass) + tid;
            return rec[pos < boundary(pass + 1) ? pos : d.N]; // dummy record when idle
        };
        auto run = [&](const uint4 raw) {
            const SiteRec r(raw);
            uint32_t nb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                nb[k] = (MODE == 0 || k < 2 * MODE) ? lds_word(sw, r.nb[k]) : 0u;
            const uint32_t s = lds_word(sw, r.site);
            const uint2 lv =
                *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(pl) + 2 * r.site);
            uint32_t acc;
            if (GIBBS)
                acc = accept_mask_gibbs(s, nb, r.jm, d.nn, lv.x, lv.y, lds_word(pl2, r.site));
            else if (dbg & 1)
                acc = nb[0] ^ nb[1] ^ nb[2] ^ nb[3] ^ nb[4] ^ nb[5] ^ lv.x ^ lv.y;
            else if (MODE == 0)
                acc = accept_mask(s, nb, r.jm, lv.x, lv.y);
            else
                acc = accept_mask_dense<MODE == 0 ? 1 : MODE>(s, nb, r.jm, lv.x, lv.y);
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sw) + r.site) =
                s ^ (acc & valid_bits);
            if (!(dbg & 4)) lds_barrier();
        };
        // eight rotating buffers: a record is requested seven passes (about 3 us) before its use
        uint4 b0 = fetch(0), b1 = fetch(1), b2 = fetch(2), b3 = fetch(3), b4 = fetch(4),
              b5 = fetch(5), b6 = fetch(6), b7;
        const uint32_t n_run = (dbg & 8) ? 0u : d.n_passes;
        for (uint32_t pass = 0; pass < n_run; pass += 8) {
            b7 = fetch(pass + 7);
            run(b0);
            if (pass + 1 >= n_run) break;
            b0 = fetch(pass + 8);
            run(b1);
            if (pass + 2 >= n_run) break;
            b1 = fetch(pass + 9);
            run(b2);
            if (pass + 3 >= n_run) break;
            b2 = fetch(pass + 10);
            run(b3);
            if (pass + 4 >= n_run) break;
            b3 = fetch(pass + 11);
            run(b4);
            if (pass + 5 >= n_run) break;
            b4 = fetch(pass + 12);
            run(b5);
            if (pass + 6 >= n_run) break;
            b5 = fetch(pass + 13);
            run(b6);
            if (pass + 7 >= n_run) break;
            b6 = fetch(pass + 14);
            run(b7);
        }
        PA_STAMP(2)
        for (uint32_t i = tid; i < d.N; i += 256) Sout[i] = sw[i];
    }
    PA_STAMP(3)

    if (with_energy & 1) {
        uint32_t c[3][8];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[s][j] = 0;
        for (uint32_t base = 0; base < d.N; base += 16 * 256) {
            uint4 rr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { // 16 record loads in flight
                const uint32_t pos = base + k * 256 + tid;
                rr[k] = rec[pos < d.N ? pos : d.N];
            }
            uint32_t qa[16], qb[16], qc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool live_site = base + k * 256 + tid < d.N;
                const SiteRec r(rr[k]);
                const uint32_t jm = r.jm;
                const uint32_t s = lds_word(sw, r.site);
                const uint32_t f0 = lds_word(sw, r.nb[0]), f1 = lds_word(sw, r.nb[2]),
                               f2 = lds_word(sw, r.nb[4]);
                const uint32_t u0 = (s ^ f0 ^ (0u - ((jm >> 0) & 1u))) & (0u - ((jm >> 6) & 1u));
                const uint32_t u1 = (s ^ f1 ^ (0u - ((jm >> 2) & 1u))) & (0u - ((jm >> 8) & 1u));
                const uint32_t u2 = (s ^ f2 ^ (0u - ((jm >> 4) & 1u))) & (0u - ((jm >> 10) & 1u));
                qa[k] = live_site ? (u0 ^ u1 ^ u2) : 0u;   // unsatisfied bonds, weight 1
                qb[k] = live_site ? maj3(u0, u1, u2) : 0u; // weight 2
                qc[k] = live_site ? s : 0u;                // spins equal to -1
            }
            count16(qa, c[0]);
            count16(qb, c[1]);
            count16(qc, c[2]);
        }
        PA_STAMP(4)
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
    PA_STAMP(5)
}

// ---------------------------------------------------------------------------
// k_sweep_wide: the level-ordered Metropolis sweep for four to six forward directions (bcc: 8 neighbours, fcc: 12)
// with unit couplings.  The reference's table path serves any z' (sweep.rs:109-139 builds 4 z' + 1 columns,
// :170-185 decides with one integer compare); round 2 sent these lattices through the f32 kernels of the
// general-coupling path (276 / 469 us per step on the bench ensemble, five launches, 201 MB of thresholds a sweep).
// Here: the random-stream kernel files the acceptance LEVEL of every draw, L = #{k : draw < table[dE = -2k]}
// (k <= z' <= 6: three bit planes, the heat bath's seven-threshold machinery with the Metropolis table), and the sweep
// decides flip <=> U + L >= h, U = unsatisfied non-zero bonds of the site (twelve bit-sliced inputs -> four planes),
// h = ceil(m / 2), as the carry out of a five-bit addition.  Site records (two 16-byte words: thirteen LDS offsets,
// coupling signs, non-zero flags, h) stream from global memory four passes ahead.  The epilogue leaves the energies
// as f32 bits of the exact integer sum (the handle's convention) and the magnetisation sums.
// ---------------------------------------------------------------------------
struct WideRec {
    uint32_t site, nb[12], h, neg, nz;
    __device__ __forceinline__ WideRec(const uint4 a, const uint4 b) {
        site = a.x & 0xFFFFu;
        nb[0] = a.x >> 16;
        nb[1] = a.y & 0xFFFFu; nb[2] = a.y >> 16;
        nb[3] = a.z & 0xFFFFu; nb[4] = a.z >> 16;
        nb[5] = a.w & 0xFFFFu; nb[6] = a.w >> 16;
        nb[7] = b.x & 0xFFFFu; nb[8] = b.x >> 16;
        nb[9] = b.y & 0xFFFFu; nb[10] = b.y >> 16;
        nb[11] = b.z & 0xFFFFu;
        h = b.z >> 16;
        neg = b.w & 0xFFFu;
        nz = (b.w >> 12) & 0xFFFu;
    }
};

// flip <=> U + L >= h from the bit planes of U (<= 12), the draw's level L (l0..l2) and h = ceil(m / 2): the carry out
// of a five-bit addition, (U + L) + (32 - h) >= 32 (h = 0: every site flips)
__device__ __forceinline__ uint32_t accept_from_count(uint32_t U0, uint32_t U1, uint32_t U2, uint32_t U3, uint32_t l0,
                                                      uint32_t l1, uint32_t l2, uint32_t h) {
    const uint32_t a0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t a1 = U1 ^ l1 ^ k0, k1 = maj3(U1, l1, k0);
    const uint32_t a2 = U2 ^ l2 ^ k1, k2 = maj3(U2, l2, k1);
    const uint32_t a3 = U3 ^ k2, a4 = U3 & k2;
    const uint32_t K = 32u - h;
    const uint32_t K0 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(K), 0, 1));
    const uint32_t K1 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(K), 1, 1));
    const uint32_t K2 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(K), 2, 1));
    const uint32_t K3 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(K), 3, 1));
    const uint32_t K4 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(K), 4, 1));
    uint32_t c = a0 & K0;
    c = maj3(a1, K1, c);
    c = maj3(a2, K2, c);
    c = maj3(a3, K3, c);
    c = maj3(a4, K4, c);
    return h == 0 ? 0xFFFFFFFFu : c;
}

// flip mask of one site word: u[k] = unsatisfied non-zero bond k (k < 12), l0..l2 the draw's level, h = ceil(m / 2)
__device__ __forceinline__ uint32_t accept_mask_wide(const uint32_t u[12], uint32_t l0, uint32_t l1, uint32_t l2, uint32_t h) {
    uint32_t s0, s1, s2, s3, c0, c1, c2, c3;
    full_add(u[0], u[1], u[2], s0, c0);
    full_add(u[3], u[4], u[5], s1, c1);
    full_add(u[6], u[7], u[8], s2, c2);
    full_add(u[9], u[10], u[11], s3, c3);
    uint32_t t0, d0;
    full_add(s0, s1, s2, t0, d0);
    const uint32_t U0 = t0 ^ s3, e0 = t0 & s3;                 // weight 1
    uint32_t x0, y0, x1, y1;
    full_add(c0, c1, c2, x0, y0);
    full_add(c3, d0, e0, x1, y1);
    const uint32_t U1 = x0 ^ x1, z = x0 & x1;                  // weight 2
    uint32_t U2, U3;
    full_add(y0, y1, z, U2, U3);                               // weights 4, 8 (U <= 12)
    return accept_from_count(U0, U1, U2, U3, l0, l1, l2, h);
}

// The same decision with a site's twelve bonds spread over the four lanes of a quad (k_sweep_wide's thin-level loop):
// a lane holds the carry-save sum (sm, cr) of its three bonds, the quad adds its four two-bit numbers with two DPP
// exchanges (lane ^ 1, then lane ^ 2); every lane of the quad ends with the site's flip mask.
// (quad_swap1 / quad_swap2: bitcount.hpp)
__device__ __forceinline__ uint32_t accept_mask_quad(uint32_t sm, uint32_t cr, uint32_t l0, uint32_t l1, uint32_t l2, uint32_t h) {
    const uint32_t sm1 = quad_swap1(sm), cr1 = quad_swap1(cr);
    const uint32_t b0 = sm ^ sm1, k0 = sm & sm1;               // a pair of lanes: <= 6 in three bits
    const uint32_t b1 = cr ^ cr1 ^ k0, b2 = maj3(cr, cr1, k0);
    const uint32_t B0 = quad_swap2(b0), B1 = quad_swap2(b1), B2 = quad_swap2(b2);
    const uint32_t U0 = b0 ^ B0, k1 = b0 & B0;                 // the quad: <= 12 in four bits
    const uint32_t U1 = b1 ^ B1 ^ k1, k2 = maj3(b1, B1, k1);
    const uint32_t U2 = b2 ^ B2 ^ k2, U3 = maj3(b2, B2, k2);
    return accept_from_count(U0, U1, U2, U3, l0, l1, l2, h);
}

__host__ __device__ inline size_t sweep_wide_lds_words(const SimDims &d) {
    const size_t n1 = static_cast<size_t>(d.N) + 1;
    return (n1 + 1) / 2 * 2 + 2 * n1 + n1 + 5 * 64 + 8; // spins, two level planes, the third, reduction scratch
}

// (Measured and dropped: a workgroup of ONE wave for lattices none of whose levels exceeds 64 sites -- fcc at 16^3 has 256
// levels of 16 -- which needs no barrier at all: 256 against 244 us per step.  The level time there, 0.58 us, is the
// lone wave's ~135 dependent instructions, not the barrier.)
__global__ __launch_bounds__(256) void k_sweep_wide(SimDims d, DevPtrs p, int with_energy, int do_sweep) {
    constexpr int WAVES = 4;
    constexpr uint32_t kThreads = 64 * WAVES;
    extern __shared__ uint32_t lds[];
    const uint32_t n1 = d.N + 1;
    uint32_t *sw = lds;                    // [N + 1] spins of this word (+ spare slot N for idle lanes)
    uint32_t *pl = sw + (n1 + 1) / 2 * 2;  // [N + 1][2] level planes 0, 1
    uint32_t *pl2 = pl + 2 * n1;           // [N + 1] level plane 2
    uint32_t *scratch = pl2 + n1;          // [5][64]
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t *Sin = p.S + (static_cast<size_t>(real) * d.W + w) * d.N;
    uint32_t *Sout = p.S_next + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint4 *rec = p.rec_wide + static_cast<size_t>(real) * 2 * n1;
    const uint32_t *planes_in = reinterpret_cast<const uint32_t *>(p.planes + (static_cast<size_t>(real) * d.W + w) * d.N);
    copy_to_lds(sw, Sin, d.N, tid, kThreads);
    if (do_sweep) {
        copy_to_lds(pl, planes_in, 2 * d.N, tid, kThreads);
        copy_to_lds(pl2, p.planes_hi + (static_cast<size_t>(real) * d.W + w) * d.N, d.N, tid, kThreads);
    }
    if (tid == 0) {
        sw[d.N] = 0;
        pl[2 * d.N] = 0;
        pl[2 * d.N + 1] = 0;
        pl2[d.N] = 0;
    }
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    __syncthreads();
    auto unsat = [&](const WideRec &r, uint32_t s, uint32_t u[12]) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
            u[k] = (s ^ lds_word(sw, r.nb[k]) ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(r.neg), k, 1))) &
                   static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(r.nz), k, 1));
    };
    if (do_sweep && d.thin_chunks) {
        // ---- thin levels (fcc at 16^3: 256 levels of 16 sites): a level is far less than a wave of sites, and what it
        // costs is the ~135 instructions a lone wave issues for it at one per ~4.7 cycles.  Here the FIRST WAVE alone
        // sweeps, sixteen sites at a time with a site's twelve bonds over the four lanes of a quad (rec_thin: one
        // 16-byte record per lane and chunk of <= 16 sites of one level): three bonds, one carry-save add and two quad
        // exchanges per lane, ~60 instructions per chunk.  One wave: LDS operations stay in program order, no barrier
        // between levels; the chunks of a level do not read one another.
        auto thin_loop = [&](auto dense_tag) {
            constexpr bool DENSE = decltype(dense_tag)::value; // every bond of every site non-zero: no masks, h = nn for all
            const uint32_t n = d.thin_chunks;
            const uint4 *rt = p.rec_thin + static_cast<size_t>(real) * n * 64 + tid;
            constexpr uint32_t kAhead = 8; // records in flight: a chunk lasts ~0.25 us, a global load ~1.5
            uint4 ring[kAhead];
#pragma unroll
            for (uint32_t i = 0; i < kAhead; ++i) ring[i] = rt[static_cast<size_t>(min(i, n - 1)) * 64];
            const bool writer = (tid & 3u) == 0;
            const uint32_t h_all = __builtin_amdgcn_readfirstlane(d.nn);
            for (uint32_t c = 0; c < n; c += kAhead) {
#pragma unroll
                for (uint32_t i = 0; i < kAhead; ++i) {
                    if (c + i >= n) break;
                    const uint4 r = ring[i];
                    ring[i] = rt[static_cast<size_t>(min(c + i + kAhead, n - 1)) * 64];
                    const uint32_t site = r.x & 0xFFFFu;
                    const uint32_t sv = lds_word(sw, site);
                    const uint2 lv = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(pl) + 2 * site);
                    const uint32_t l2 = lds_word(pl2, site);
                    const uint32_t n0 = lds_word(sw, r.x >> 16), n1 = lds_word(sw, r.y & 0xFFFFu), n2 = lds_word(sw, r.y >> 16);
                    const int meta = static_cast<int>(r.z);
                    uint32_t u0 = sv ^ n0 ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 0, 1));
                    uint32_t u1 = sv ^ n1 ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 1, 1));
                    uint32_t u2 = sv ^ n2 ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 2, 1));
                    if (!DENSE) {
                        u0 &= static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 3, 1));
                        u1 &= static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 4, 1));
                        u2 &= static_cast<uint32_t>(__builtin_amdgcn_sbfe(meta, 5, 1));
                    }
                    uint32_t sm, cr;
                    full_add(u0, u1, u2, sm, cr);
                    // (the planes are read with the neighbours, not behind the quad exchange where only the writing lanes
                    // would want them: one LDS round trip per chunk, not two)
                    uint32_t l0 = lv.x, l1 = lv.y, l2p = l2;
                    asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2p), "+v"(sm), "+v"(cr));
                    const uint32_t acc = accept_mask_quad(sm, cr, l0, l1, l2p, DENSE ? h_all : r.z >> 8);
                    if (writer) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sw) + site) = sv ^ (acc & valid_bits);
                }
            }
        };
        if (tid < 64) {
            if (d.thin_dense) thin_loop(std::true_type{});
            else thin_loop(std::false_type{});
        }
        __syncthreads();
        for (uint32_t i = tid; i < d.N; i += kThreads) Sout[i] = sw[i];
    } else if (do_sweep) {
        auto fetch = [&](uint32_t pass, uint4 &a, uint4 &b) {
            uint32_t pos = d.N; // dummy record when idle: every offset points at the spare slot, h = 0 masked below
            if (pass < d.n_passes) {
                const uint32_t at = p.pass_start[pass] + tid;
                if (at < p.pass_start[pass + 1]) pos = at;
            }
            a = rec[2 * pos];
            b = rec[2 * pos + 1];
        };
        const uint32_t wave_first = __builtin_amdgcn_readfirstlane(tid & ~63u);
        uint32_t pass_at = 0; // the pass `run` is about to execute
        auto run = [&](const uint4 a, const uint4 b) {
            // a wave none of whose lanes has a site in this pass goes straight to the barrier (k_sweep_hc does the same)
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(p.pass_start[pass_at]), b1 = __builtin_amdgcn_readfirstlane(p.pass_start[pass_at + 1]);
            ++pass_at;
            if (b0 + wave_first >= b1) {
                lds_barrier();
                return;
            }
            const WideRec r(a, b);
            const uint32_t s = lds_word(sw, r.site);
            const uint2 lv = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(pl) + 2 * r.site);
            const uint32_t l2 = lds_word(pl2, r.site);
            uint32_t u[12];
            unsat(r, s, u);
            const uint32_t acc = accept_mask_wide(u, lv.x, lv.y, l2, r.h);
            // (idle lanes hold the dummy record: site = spare slot, whose value nobody reads)
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sw) + r.site) = s ^ (acc & valid_bits);
            lds_barrier();
        };
        uint4 a0, b0, a1, b1, a2, b2, a3, b3;
        fetch(0, a0, b0);
        fetch(1, a1, b1);
        fetch(2, a2, b2);
        for (uint32_t pass = 0; pass < d.n_passes; pass += 4) {
            fetch(pass + 3, a3, b3);
            run(a0, b0);
            if (pass + 1 >= d.n_passes) break;
            fetch(pass + 4, a0, b0);
            run(a1, b1);
            if (pass + 2 >= d.n_passes) break;
            fetch(pass + 5, a1, b1);
            run(a2, b2);
            if (pass + 3 >= d.n_passes) break;
            fetch(pass + 6, a2, b2);
            run(a3, b3);
        }
        for (uint32_t i = tid; i < d.N; i += kThreads) Sout[i] = sw[i];
    }
    if (with_energy & 1) { // energy.rs:92-109 on the swept word: unsatisfied non-zero FORWARD bonds (even slots), spins
        uint32_t c[4][8];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[s][j] = 0;
        for (uint32_t base = 0; base < d.N; base += 16 * kThreads) {
            uint32_t q1[16], q2[16], q4[16], qs[16];
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const uint32_t pos = base + k * kThreads + tid;
                const bool live_site = pos < d.N;
                const WideRec r(rec[2 * (live_site ? pos : d.N)], rec[2 * (live_site ? pos : d.N) + 1]);
                const uint32_t s = lds_word(sw, r.site);
                uint32_t f[6];
#pragma unroll
                for (int dir = 0; dir < 6; ++dir)
                    f[dir] = (s ^ lds_word(sw, r.nb[2 * dir]) ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(r.neg), 2 * dir, 1))) &
                             static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(r.nz), 2 * dir, 1));
                uint32_t s0, c0, s1, c1;
                full_add(f[0], f[1], f[2], s0, c0);
                full_add(f[3], f[4], f[5], s1, c1);
                const uint32_t dd = s0 & s1;
                q1[k] = live_site ? (s0 ^ s1) : 0u;
                q2[k] = live_site ? (c0 ^ c1 ^ dd) : 0u;
                q4[k] = live_site ? maj3(c0, c1, dd) : 0u;
                qs[k] = live_site ? s : 0u;
            }
            count16(q1, c[0]);
            count16(q2, c[1]);
            count16(q4, c[2]);
            count16(qs, c[3]);
        }
        uint32_t tot[4];
        block_bit_totals<4, WAVES>(c, scratch, tot);
        if (tid < 32 && w * 32 + tid < d.n_sys) {
            const uint32_t slot = w * 32 + tid;
            const int32_t n_unsat = static_cast<int32_t>(tot[0] + 2 * tot[1] + 4 * tot[2]);
            const float energy_sum = static_cast<float>(p.nz_bonds[real] - 2 * n_unsat); // exact: |sum| <= 6 N < 2^24
            p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = d.general ? __float_as_int(energy_sum) : static_cast<int32_t>(energy_sum);
            p.Msum[static_cast<size_t>(real) * d.n_sys + slot] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[3]);
        }
    }
}

// ---------------------------------------------------------------------------
// k_sweep_hc: the same sweep for lattices whose neighbours follow from index arithmetic (SimDims::hc: every
// hypercubic lattice; sweep.rs:8-19 reads neighbour tables, lattice.rs:44-99 builds them) and whose bonds are
// all non-zero (+-J, ferromagnets).  Differences from k_sweep:
//   * no 16-byte site records from global memory: the word's schedule table (4 bytes per site: site index,
//     six wrap flags, six coupling signs) is staged in the LDS with the spins and the acceptance planes, and
//     the seven LDS addresses of a site come from three instructions per neighbour;
//   * the pass loop is software-pipelined by hand: the table entry of pass p + 2 and the addresses of pass
//     p + 1 are produced while the nine LDS reads of pass p are in flight, so the dependent chain of a pass is
//     barrier -> reads -> accept logic -> write -> barrier and nothing else;
//   * the energy epilogue walks the same table in the LDS (no second pass over global records).
// LDS: (3 (N + 1) + N) words + pass boundaries = 66 KB at 16^3: two workgroups per CU.
// ---------------------------------------------------------------------------
// LDS accesses by 32-bit LDS address: the base of the dynamic LDS is a link-time constant, so going through
// generic pointers costs one v_add per access; here it is folded into the site address once.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
struct LdsPair { uint32_t x, y; };
typedef __attribute__((address_space(3))) LdsPair lds_u32x2_t;
__device__ __forceinline__ uint32_t lds_addr_of(const uint32_t *ptr) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const lds_u32_t *)ptr));
}
__device__ __forceinline__ uint32_t lds_load(uint32_t addr) { return *reinterpret_cast<const lds_u32_t *>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ uint2 lds_load2(uint32_t addr) { // 8-byte aligned address: one ds_read_b64
    const lds_u32x2_t *q = reinterpret_cast<const lds_u32x2_t *>(static_cast<uintptr_t>(addr));
    return make_uint2(q->x, q->y);
}
__device__ __forceinline__ void lds_store(uint32_t addr, uint32_t v) { *reinterpret_cast<lds_u32_t *>(static_cast<uintptr_t>(addr)) = v; }

struct HcSite {
    uint32_t site;    // LDS address of the site's word
    uint32_t nb[6];   // LDS addresses of f0, b0, f1, b1, f2, b2
    uint32_t site4;   // byte offset of the site inside the word array
    uint32_t neg;     // coupling sign bits of those bonds
};

// sw_addr: LDS address of the word array
template <int NN>
__device__ __forceinline__ HcSite hc_decode(const SimDims &d, uint32_t e, uint32_t sw_addr) {
    HcSite h;
    h.site4 = (e & ((1u << kHcSiteBits) - 1u)) << 2;
    h.site = sw_addr + h.site4;
    h.neg = e >> kHcNegShift;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < 2 * NN) {
            // site + regular offset + flag * (wrapped - regular): v_bfe_u32, v_add_u32, v_mad_i32_i24 (written
            // out: the compiler turns the product into and / compare / select / add3)
            const uint32_t flag = __builtin_amdgcn_ubfe(e, kHcFlagShift + k, 1);
            const uint32_t base = h.site + static_cast<uint32_t>(d.hc_d0[k]);
            uint32_t addr;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr) : "v"(flag), "s"(d.hc_dd[k]), "v"(base));
            h.nb[k] = addr;
        } else {
            h.nb[k] = h.site;
        }
    }
    return h;
}

__host__ __device__ inline size_t sweep_hc_lds_words(const SimDims &d, bool gibbs = false) {
    const size_t n1 = static_cast<size_t>(d.N) + 1, n_pad = (n1 + 3) & ~size_t(3);
    const size_t tab = (static_cast<size_t>(d.N) + 3) & ~size_t(3);
    return n_pad + ((2 * n1 + 3) & ~size_t(3)) + tab + ((static_cast<size_t>(d.n_passes) + 6 + 3) & ~size_t(3)) + 5 * 48 +
           (gibbs ? n_pad : 0); // the heat bath's third acceptance plane
}

// STAMPS: diagnostic build, every workgroup files s_memrealtime (100 MHz) stamps of its phases in stamps[block][8]
#define PA_HC_STAMP(k)                                                                             \
    if (STAMPS && threadIdx.x == 0) stamps[bid * 8 + (k)] = wall_clock64();
// Energy sum_{i,d} J s_i s_fwd and magnetisation sum_i s_i of the 32 slots of one tile word (energy.rs:92-109) from the
// word in the LDS (sw_addr) and the schedule table (`tab`: LDS or global), by byte-sliced vertical counters.  For the
// first 256 threads of a workgroup (the others must have left); scratch: [5][48] words of LDS.
template <int NN>
__device__ __forceinline__ void hc_energy(const SimDims &d, const DevPtrs &p, uint32_t real, uint32_t w, const uint32_t *tab,
                                          uint32_t sw_addr, uint32_t *scratch, int32_t *Esum, int32_t *Msum) {
    const uint32_t tid = threadIdx.x;
    uint32_t c[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
#pragma unroll 1
    for (uint32_t base = 0; base < d.N; base += 16 * 256) {
        uint32_t qa[16], qb[16], qc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t pos = base + k * 256 + tid;
            const bool live_site = pos < d.N;
            const HcSite h = hc_decode<NN>(d, tab[live_site ? pos : 0u], sw_addr);
            const uint32_t s = lds_load(h.site);
            uint32_t u[3] = {0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < NN; ++q) // unsatisfied non-zero forward bonds (neg: bits 0-5, nz: bits 6-11 of h.neg)
                u[q] = (s ^ lds_load(h.nb[2 * q]) ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(h.neg), 2 * q, 1))) &
                       static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(h.neg), 6 + 2 * q, 1));
            qa[k] = live_site ? (u[0] ^ u[1] ^ u[2]) : 0u;   // weight 1
            qb[k] = live_site ? maj3(u[0], u[1], u[2]) : 0u; // weight 2
            qc[k] = live_site ? s : 0u;                      // spins equal to -1
        }
        count16(qa, c[0]);
        count16(qb, c[1]);
        count16(qc, c[2]);
    }
    uint32_t tot[3];
    block_bit_totals<3>(c, scratch, tot);
    if (tid < 32 && w * 32 + tid < d.n_sys) {
        const uint32_t slot = w * 32 + tid;
        const int32_t unsat = static_cast<int32_t>(tot[0] + 2 * tot[1]);
        Esum[static_cast<size_t>(real) * d.n_sys + slot] = p.nz_bonds[real] - 2 * unsat;
        Msum[static_cast<size_t>(real) * d.n_sys + slot] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[2]);
    }
}

// The same sums by all 512 threads of the sweep workgroup, for 512 <= N <= 4096 (at most eight sites per thread, so
// a 16-lane row sums its byte counters without widening: 8 x 16 = 128).  Four DPP steps leave the row sums in every
// lane; the 32 rows file theirs in `part` ([32][24] words of LDS, the dead acceptance planes) and 96 threads add them up.
__device__ __forceinline__ void count8(const uint32_t w[8], uint32_t c[8]) {
    uint32_t s0, s1, c0, c1, t0, d0;
    full_add(w[0], w[1], w[2], s0, c0);
    full_add(w[3], w[4], w[5], s1, c1);
    full_add(s0, s1, w[6], t0, d0);
    const uint32_t p0 = t0 ^ w[7], d1 = t0 & w[7]; // weight 1
    uint32_t u0, e0;
    full_add(c0, c1, d0, u0, e0);
    const uint32_t p1 = u0 ^ d1, e1 = u0 & d1;     // weight 2
    const uint32_t p2 = e0 ^ e1, p3 = e0 & e1;     // weights 4, 8
#pragma unroll
    for (int j = 0; j < 8; ++j)
        c[j] = ((p0 >> j) & 0x01010101u) + (((p1 >> j) & 0x01010101u) << 1) + (((p2 >> j) & 0x01010101u) << 2) +
               (((p3 >> j) & 0x01010101u) << 3);
}

template <int NN>
__device__ __forceinline__ void hc_energy_wide(const SimDims &d, const DevPtrs &p, uint32_t real, uint32_t w, const uint32_t *tab,
                                               uint32_t sw_addr, uint32_t *part, uint32_t *scratch) {
    const uint32_t tid = threadIdx.x;
    const int32_t nz = p.nz_bonds[real]; // (asked for now: a global round trip at the end would be fully exposed)
    uint32_t qa[8], qb[8], qc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t pos = k * 512 + tid;
        const bool live_site = pos < d.N;
        const HcSite h = hc_decode<NN>(d, tab[live_site ? pos : 0u], sw_addr);
        const uint32_t s = lds_load(h.site);
        uint32_t u[3] = {0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < NN; ++q) // unsatisfied non-zero forward bonds
            u[q] = (s ^ lds_load(h.nb[2 * q]) ^ static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(h.neg), 2 * q, 1))) &
                   static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(h.neg), 6 + 2 * q, 1));
        qa[k] = live_site ? (u[0] ^ u[1] ^ u[2]) : 0u;   // weight 1
        qb[k] = live_site ? maj3(u[0], u[1], u[2]) : 0u; // weight 2
        qc[k] = live_site ? s : 0u;                      // spins equal to -1
    }
    // (the quad form of bitcount.hpp: a quad adds its four bit-sliced counts, each of its lanes spreads two of the eight
    // bit-position classes, and a row of 16 lanes sums lanes of equal class -- bytes of at most 8 x 16 = 128)
    uint32_t c2[3][2];
    {
        const uint32_t *in[3] = {qa, qb, qc};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            uint32_t pl4[4], pl6[6];
            planes8(in[s], pl4);
            quad_add_planes<4>(pl4, pl6);
            spread_pair<6>(pl6, tid & 3u, c2[s]);
        }
    }
    __syncthreads(); // (every read of the planes is behind us: `part` may be written)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (uint32_t jj = 0; jj < 2; ++jj) {
            uint32_t v = c2[s][jj];
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xF, 0xF, false));
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xF, 0xF, false));
            if ((tid & 12u) == 12u) part[(tid >> 4) * 24 + s * 8 + 2 * (tid & 3u) + jj] = v;
        }
    __syncthreads();
    if (tid < 96) { // thread = (set, slot); slot = j + 8 * byte of the sliced counters
        const uint32_t set = tid >> 5, slot = tid & 31u, j = slot & 7u, shift = (slot >> 3) * 8;
        uint32_t sum = 0;
#pragma unroll 8
        for (uint32_t q = 0; q < 32; ++q) sum += (part[q * 24 + set * 8 + j] >> shift) & 0xFFu;
        scratch[tid] = sum;
    }
    __syncthreads();
    if (tid < 32 && w * 32 + tid < d.n_sys) {
        const uint32_t slot = w * 32 + tid;
        const int32_t unsat = static_cast<int32_t>(scratch[tid] + 2 * scratch[32 + tid]);
        p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = nz - 2 * unsat;
        p.Msum[static_cast<size_t>(real) * d.n_sys + slot] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(scratch[64 + tid]);
    }
}

// Workgroup = (realisation, word), 512 threads in two sets of four waves.  A pass is a chain -- nine LDS reads, the
// accept logic, one write, the barrier -- and a lone wave issues one instruction every ~4.7 cycles however independent
// they are (profiles/r02_valu_issue.json), so everything that is not on the chain is done by OTHER waves: the sets take
// the passes in turn, and while one computes pass p the other fetches the table entry of pass p + 1 and decodes its
// seven LDS addresses, then meets it at the barrier.  Between two barriers one set does "reads -> accept -> write" and
// nothing else.
template <int NN, bool STAMPS, bool GIBBS, bool DENSE>
__device__ __forceinline__ void sweep_hc_block(const SimDims &d, const DevPtrs &p, uint32_t bid, int with_energy, int do_sweep,
                                               unsigned long long *stamps, uint32_t *lds) {
    PA_HC_STAMP(0)
    constexpr uint32_t kThreads = 512;
    const uint32_t n1 = d.N + 1;
    const uint32_t n_pad = (n1 + 3u) & ~3u;
    uint32_t *sw = lds;                                   // [N + 1] spins of this word (+ spare slot N)
    uint32_t *pl = sw + n_pad;                            // [N + 1][2] acceptance-level planes
    uint32_t *tab = pl + ((2 * n1 + 3u) & ~3u);           // [N] schedule table of this realisation
    uint32_t *ps = tab + ((d.N + 3u) & ~3u);              // [n_passes + 6] pass boundaries (padded with N)
    uint32_t *scratch = ps + ((d.n_passes + 6u + 3u) & ~3u); // [5][48]
    uint32_t *hi = scratch + 5 * 48;                      // [N + 1] heat bath: third acceptance plane
    uint32_t real, w;
    tile_of_block(d, bid, real, w);
    const uint32_t tid = threadIdx.x;
    const uint32_t *Sin = p.S + (static_cast<size_t>(real) * d.W + w) * d.N;
    uint32_t *Sout = p.S_next + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint32_t *tab_g = p.hc_tab + static_cast<size_t>(real) * d.N;
    uint32_t mp[2] = {0, 0}, mc[2] = {0, 0}, mn[2] = {0, 0};
    // (requested first: their latency hides behind the bulk copy)
    const uint32_t epoch = p.swap_epoch[real];
    const uint32_t applied = p.applied_epoch[static_cast<size_t>(real) * d.W + w];
    {
        const uint32_t *masks = p.swap_masks + static_cast<size_t>(real) * 2 * d.W;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mp[r] = w > 0 ? masks[r * d.W + w - 1] : 0u;
            mc[r] = masks[r * d.W + w];
            mn[r] = w + 1 < d.W ? masks[r * d.W + w + 1] : 0u;
        }
    }
    // spins, acceptance planes and the schedule table are requested together (one memory round trip)
    const uint32_t *planes_in =
        reinterpret_cast<const uint32_t *>(p.planes + (static_cast<size_t>(real) * d.W + w) * d.N);
    {
        const uint32_t qa = d.N >> 2, qb = do_sweep ? (2 * d.N) >> 2 : 0u, qc = d.N >> 2;
        const bool wide = (d.N & 3u) == 0 && qa <= 2 * kThreads && (2 * d.N) >> 2 <= 4 * kThreads &&
                          ((reinterpret_cast<uintptr_t>(Sin) | reinterpret_cast<uintptr_t>(planes_in) |
                            reinterpret_cast<uintptr_t>(tab_g)) & 15u) == 0;
        if (wide) {
            const uint4 *a4 = reinterpret_cast<const uint4 *>(Sin), *b4 = reinterpret_cast<const uint4 *>(planes_in),
                        *c4 = reinterpret_cast<const uint4 *>(tab_g);
            uint4 va[2], vb[4], vc[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) va[k] = k * kThreads + tid < qa ? a4[k * kThreads + tid] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) vb[k] = k * kThreads + tid < qb ? b4[k * kThreads + tid] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) vc[k] = k * kThreads + tid < qc ? c4[k * kThreads + tid] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k * kThreads + tid < qa) reinterpret_cast<uint4 *>(sw)[k * kThreads + tid] = va[k];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k * kThreads + tid < qb) reinterpret_cast<uint4 *>(pl)[k * kThreads + tid] = vb[k];
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k * kThreads + tid < qc) reinterpret_cast<uint4 *>(tab)[k * kThreads + tid] = vc[k];
        } else {
            copy_to_lds(sw, Sin, d.N, tid, kThreads);
            if (do_sweep) copy_to_lds(pl, planes_in, 2 * d.N, tid, kThreads);
            copy_to_lds(tab, tab_g, d.N, tid, kThreads);
        }
    }
    if (GIBBS && do_sweep) copy_to_lds(hi, p.planes_hi + (static_cast<size_t>(real) * d.W + w) * d.N, d.N, tid, kThreads);
    for (uint32_t k = tid; k < d.n_passes + 6; k += kThreads) ps[k] = k <= d.n_passes ? p.pass_start[k] : d.N;
    if (do_sweep && epoch != applied) {
        // tempering swaps accepted after the previous sweep: exchange the bit columns now
        const bool window = (((mp[0] | mp[1] | mc[0] | mc[1]) >> 31) != 0); // a pair straddles words
        __syncthreads();
        for (uint32_t i = tid; i < d.N; i += kThreads) {
            const uint32_t v = sw[i];
            uint32_t prev = 0, next = 0;
            if (window) { // old neighbouring words (the input buffer is not written in this launch)
                if (w > 0) prev = Sin[static_cast<ptrdiff_t>(i) - static_cast<ptrdiff_t>(d.N)];
                if (w + 1 < d.W) next = Sin[static_cast<size_t>(i) + d.N];
            }
            sw[i] = apply_swaps_window(prev, v, next, mp, mc, mn);
        }
        if (tid == 0) p.applied_epoch[static_cast<size_t>(real) * d.W + w] = epoch;
    }
    if (tid == 0) {
        sw[d.N] = 0;
        pl[2 * d.N] = 0;
        pl[2 * d.N + 1] = 0;
        if (GIBBS) hi[d.N] = 0;
    }
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    __syncthreads();

    PA_HC_STAMP(1)
    const uint32_t sw_addr = lds_addr_of(sw), pl_addr = lds_addr_of(pl), tab_addr = lds_addr_of(tab);
    const uint32_t hi_addr = GIBBS ? lds_addr_of(hi) : 0u;
    if (do_sweep) {
        const uint32_t spare = sw_addr + d.N * 4;
        const uint32_t ps_addr = lds_addr_of(ps);
        const uint32_t set = __builtin_amdgcn_readfirstlane(tid >> 8), lane = tid & 255u;
        // what a pass needs, all of it prepared by `fetch`: nine LDS addresses and the coupling signs as masks
        HcSite cur;
        uint32_t wr_cur = spare, pl_cur = pl_addr, negm[6] = {0, 0, 0, 0, 0, 0};
        uint32_t nzm[6] = {0, 0, 0, 0, 0, 0}, rule[4] = {0, 0, 0, 0}; // diluted lattices: non-zero masks, the site's rule bits
        // A dependency level of a 16^3 lattice has 1 ... 192 sites (46 hyperplanes, 89 on average), so of the four
        // waves of a set only ceil(sites / 64) have a site to update: 88 of 184 wave-passes.  The others skip the
        // pass altogether -- a wave-uniform scalar branch to the barrier -- which halves the instructions this
        // kernel issues in its pass loop (3.8 M of 7.8 M per launch of the bench ensemble) beside the VALU-bound
        // random stream of the other stream group.
        const uint32_t wave_first = __builtin_amdgcn_readfirstlane(lane & ~63u);
        bool wave_live = false; // (uniform) this wave has a site in the pass `cur` describes
        auto fetch = [&](uint32_t b0_v, uint32_t b1_v) { // the pass whose positions are [b0, b1)
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(b0_v), b1 = __builtin_amdgcn_readfirstlane(b1_v);
            wave_live = b0 + wave_first < b1;
            if (!wave_live) return;
            const uint32_t pos = b0 + lane;
            const bool active = pos < b1;
            cur = hc_decode<NN>(d, lds_load(tab_addr + (active ? pos : 0u) * 4), sw_addr);
            wr_cur = active ? cur.site : spare;
            pl_cur = pl_addr + 2 * cur.site4;
#pragma unroll
            for (int k = 0; k < 2 * NN; ++k) negm[k] = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(cur.neg), k, 1));
            if (!DENSE) {
#pragma unroll
                for (int k = 0; k < 2 * NN; ++k) nzm[k] = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(cur.neg), 6 + k, 1));
                const uint32_t m = __popc((cur.neg >> 6) & 63u); // non-zero bonds of the site
                // Metropolis: h = ceil(m / 2) (accept_mask); heat bath: K = 16 - (z' + 1 + floor(m / 2)) (accept_mask_gibbs)
                const uint32_t code = GIBBS ? 16u - (NN + 1u + (m >> 1)) : (m + 1u) >> 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) rule[k] = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(code), k, 1));
#pragma unroll
                for (int k = 0; k < 2 * NN; ++k) asm volatile("" : "+v"(nzm[k]));
                asm volatile("" : "+v"(rule[0]), "+v"(rule[1]), "+v"(rule[2]), "+v"(rule[3]));
            }
            // all of it is computed HERE, before the barrier the set is about to wait at (left alone, the compiler sinks
            // the address arithmetic to its first use, behind the barrier and onto the chain)
            asm volatile("" : "+v"(cur.site), "+v"(wr_cur), "+v"(pl_cur));
#pragma unroll
            for (int k = 0; k < 2 * NN; ++k) asm volatile("" : "+v"(cur.nb[k]), "+v"(negm[k]));
        };
        auto compute = [&]() { // reads -> accept -> write: the chain between two barriers
            if (!wave_live) return;
            const uint2 lv = lds_load2(pl_cur);
            const uint32_t l2 = GIBBS ? lds_load(hi_addr + cur.site4) : 0u;
            const uint32_t s = lds_load(cur.site);
            uint32_t nb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) nb[k] = k < 2 * NN ? (lds_load(cur.nb[k]) ^ negm[k]) : 0u;
            uint32_t acc;
            if (DENSE) {
                acc = GIBBS ? accept_mask_gibbs_dense<NN>(s, nb, lv.x, lv.y, l2) : accept_mask_dense<NN>(s, nb, 0u, lv.x, lv.y);
            } else {
                uint32_t u[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] = k < 2 * NN ? ((s ^ nb[k]) & nzm[k]) : 0u;
                acc = GIBBS ? accept_u_gibbs(u, lv.x, lv.y, l2, rule[0], rule[1], rule[2], rule[3])
                            : accept_u_metropolis(u, lv.x, lv.y, rule[0], rule[1]);
            }
            lds_store(wr_cur, s ^ (acc & valid_bits));
        };
        // set 0 takes the even passes, set 1 the odd ones (two loops, one barrier sequence: every wave meets the same
        // number of barriers); ps is padded, passes past the last are empty
        if (set == 0) {
            fetch(ps[0], ps[1]);
            uint32_t nb0 = ps[2], nb1 = ps[3];
#pragma unroll 1
            for (uint32_t pass = 0; pass < d.n_passes; pass += 2) {
                compute(); // pass
                lds_barrier();
                const uint32_t f0 = lds_load(ps_addr + (pass + 4) * 4), f1 = lds_load(ps_addr + (pass + 5) * 4);
                fetch(nb0, nb1); // pass + 2
                nb0 = f0;
                nb1 = f1;
                lds_barrier();
            }
        } else {
            uint32_t nb0 = ps[1], nb1 = ps[2];
#pragma unroll 1
            for (uint32_t pass = 0; pass < d.n_passes; pass += 2) {
                const uint32_t f0 = lds_load(ps_addr + (pass + 3) * 4), f1 = lds_load(ps_addr + (pass + 4) * 4);
                fetch(nb0, nb1); // pass + 1
                nb0 = f0;
                nb1 = f1;
                lds_barrier();
                compute(); // pass + 1
                lds_barrier();
            }
        }
        PA_HC_STAMP(2)
        for (uint32_t i = tid; i < d.N; i += kThreads) Sout[i] = sw[i];
    }
    PA_HC_STAMP(3)

    if (with_energy & 1) {
        PA_HC_STAMP(4)
        if (d.N >= 512 && d.N <= 8 * kThreads) {
            hc_energy_wide<NN>(d, p, real, w, tab, sw_addr, pl, scratch);
        } else {
            if (tid >= 256) return; // (a barrier waits for the surviving waves only)
            hc_energy<NN>(d, p, real, w, tab, sw_addr, scratch, p.Esum, p.Msum);
        }
    }
    PA_HC_STAMP(5)
}

// Register budget of five waves per SIMD (<= 96 VGPRs; the compiler lands at 77 without spilling, against 123 when
// given 128): a sweep workgroup's two waves per SIMD then fit beside THREE random-stream waves of 112 registers.
template <int NN, bool STAMPS = false, bool GIBBS = false, bool DENSE = true>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(5))) void
k_sweep_hc(SimDims d, DevPtrs p, int with_energy, int do_sweep,
                                                     unsigned long long *stamps) {
    extern __shared__ uint32_t lds[];
    sweep_hc_block<NN, STAMPS, GIBBS, DENSE>(d, p, blockIdx.x, with_energy, do_sweep, stamps, lds);
}

// EXPERIMENT (developer switch PEAPODS_FLAGGED, sweep-only entry points): the random stream and the sweep of ONE Metropolis
// sweep in one launch, the sweep workgroup of a tile word released by a counter as soon as the stream workgroups of its
// words have stored their planes -- the dependency-flag form the round-2 all-in-one kernel (k_rs) did not try.  Stream
// workgroups come first in block order (= dispatch order: every one of them is resident or done before a sweep workgroup
// takes a slot, so the waiting cannot starve them) and have eight waves, so that one launch geometry (512 threads, the
// sweep's LDS) holds as many stream waves per CU as the plain launch does.  Release: every stream lane fences its stores
// (agent scope: the planes cross XCDs), one lane bumps flags[group]; acquire: one lane polls, the workgroup fences.
// Measured slower than the two launches (profiles/NOTES.md, round 4) -- kept for the record, not on any product path.
template <int NN, bool DENSE>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(4))) void
k_stream_sweep_flagged(SimDims d, DevPtrs p, uint32_t n_stream, uint32_t *flags, uint32_t target, uint32_t *timed_out) {
    extern __shared__ uint32_t lds_dyn[];
    if (blockIdx.x < n_stream) {
        rng_block<true, false>(d, p, blockIdx.x, lds_dyn);
        drain_global_stores(); // every wave's stores have left the CU ...
        __syncthreads();
        if (threadIdx.x == 0) { // ... and ONE lane publishes them (the release writes back the L2: once per workgroup, not per wave)
            __threadfence();
            __hip_atomic_fetch_add(&flags[blockIdx.x / d.B], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const uint32_t bid = blockIdx.x - n_stream; // tile word; a stream workgroup of eight waves serves sixteen of them
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while (__hip_atomic_load(&flags[bid / 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { // (~1 s: a lost release must not hang the device)
                atomicOr(timed_out, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __threadfence(); // acquire, once per workgroup
    }
    __syncthreads();
    sweep_hc_block<NN, false, false, DENSE>(d, p, bid, 0, 1, nullptr, lds_dyn);
}

// ---------------------------------------------------------------------------
// k_overlap: workgroup = (tile, word), 256 threads.  Replica 2p and 2p+1 at
// temperature t sit n_temps bit positions apart, so Y = V ^ (V >> n_temps) lines
// every pair up in the bits of the even replica.  V = S gives q_i; V = S ^ S(fwd_d)
// gives the bond products, and q_i q_k = +1 <=> both replicas agree on the bond
// (overlap.rs:259-282).  Per-bit counts by byte-sliced counters -> dot_spin, dot_link.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void overlap_block(const SimDims &d, const DevPtrs &p, uint32_t bid, uint32_t *lds) {
    const uint32_t n1 = d.N + 1;
    uint32_t *Y = lds;                 // [N + 1] pair-difference word: 1 where replicas 2p, 2p+1 disagree
    uint32_t *scratch = lds + n1;      // [17][48]
    uint32_t real, w;
    tile_of_block(d, bid, real, w);
    const uint32_t tid = threadIdx.x;
    const uint32_t q = d.n_temps >> 5, r = d.n_temps & 31u;
    const uint32_t wa = w + q, wb = w + q + 1;
    const uint32_t *Sg = p.S + static_cast<size_t>(real) * d.W * d.N;
    const uint4 *rec = p.rec + static_cast<size_t>(real) * n1;
    {
        const uint32_t *A = Sg + static_cast<size_t>(w) * d.N;
        const uint32_t *B0 = wa < d.W ? Sg + static_cast<size_t>(wa) * d.N : nullptr;
        const uint32_t *B1 = (r && wb < d.W) ? Sg + static_cast<size_t>(wb) * d.N : nullptr;
        for (uint32_t base = 0; base < d.N; base += 8 * 256) { // 8 x 3 loads in flight
            uint32_t va[8], v0[8], v1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * 256 + tid;
                va[k] = i < d.N ? A[i] : 0u;
                v0[k] = (B0 && i < d.N) ? B0[i] : 0u;
                v1[k] = (B1 && i < d.N) ? B1[i] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * 256 + tid;
                if (i < d.N) Y[i] = va[k] ^ (r ? ((v0[k] >> r) | (v1[k] << (32 - r))) : v0[k]);
            }
        }
        if (tid == 0) Y[d.N] = 0;
    }
    __syncthreads();
    uint32_t c[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    // (up to 16 sites per thread -- every lattice of at most 4096 sites -- are counted once: the quad form of bitcount.hpp,
    // a third fewer instructions than spreading and summing every lane's own counters)
    const bool one_chunk = d.N <= 16 * 256;
    uint32_t c2[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    auto tally = [&](const uint32_t (&qa)[16], const uint32_t (&qb)[16], const uint32_t (&qc)[16]) {
        if (one_chunk) {
            const uint32_t *in[3] = {qa, qb, qc};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                uint32_t pl5[5], pl7[7];
                planes16(in[s], pl5);
                quad_add_planes<5>(pl5, pl7);
                spread_pair<7>(pl7, tid & 3u, c2[s]);
            }
        } else {
            count16(qa, c[0]);
            count16(qb, c[1]);
            count16(qc, c[2]);
        }
    };
    if (d.hc) { // implicit neighbours: 4 bytes per site from the schedule table instead of a 16-byte record
        const uint32_t *tab_g = p.hc_tab + static_cast<size_t>(real) * d.N;
        const uint32_t y_addr = lds_addr_of(Y);
#pragma unroll 1
        for (uint32_t base = 0; base < d.N; base += 16 * 256) {
            uint32_t ee[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t pos = base + k * 256 + tid;
                ee[k] = tab_g[pos < d.N ? pos : 0u];
            }
            uint32_t qa[16], qb[16], qc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool live_site = base + k * 256 + tid < d.N;
                const HcSite h = hc_decode<3>(d, ee[k], y_addr); // (slots of absent directions: flag 0, offset 0)
                const uint32_t y = lds_load(h.site);
                uint32_t z[3] = {0u, 0u, 0u};
#pragma unroll
                for (int dir = 0; dir < 3; ++dir) // 1 where q_i * q_k = -1
                    if (dir < static_cast<int>(d.nn)) z[dir] = y ^ lds_load(h.nb[2 * dir]);
                qa[k] = live_site ? y : 0u;
                qb[k] = live_site ? (z[0] ^ z[1] ^ z[2]) : 0u;
                qc[k] = live_site ? maj3(z[0], z[1], z[2]) : 0u;
            }
            tally(qa, qb, qc);
        }
    } else {
#pragma unroll 1
        for (uint32_t base = 0; base < d.N; base += 16 * 256) {
            uint4 rr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t pos = base + k * 256 + tid;
                rr[k] = rec[pos < d.N ? pos : d.N];
            }
            uint32_t qa[16], qb[16], qc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool live_site = base + k * 256 + tid < d.N;
                const SiteRec sr(rr[k]);
                const uint32_t y = lds_word(Y, sr.site);
                uint32_t z[3];
#pragma unroll
                for (int dir = 0; dir < 3; ++dir) // 1 where q_i * q_k = -1
                    z[dir] = dir < static_cast<int>(d.nn) ? (y ^ lds_word(Y, sr.nb[2 * dir])) : 0u;
                qa[k] = live_site ? y : 0u;
                qb[k] = live_site ? (z[0] ^ z[1] ^ z[2]) : 0u;
                qc[k] = live_site ? maj3(z[0], z[1], z[2]) : 0u;
            }
            tally(qa, qb, qc);
        }
    }
    uint32_t tot[3];
    if (one_chunk) block_bit_totals_quad<3>(c2, scratch, tot);
    else block_bit_totals<3>(c, scratch, tot);
    if (tid < 32) {
        const uint32_t k = w * 32 + tid;
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if (k < d.n_sys && (rep & 1u) == 0 && rep + 1 < d.n_replicas) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[0]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) -
                            2 * static_cast<int32_t>(tot[1] + 2 * tot[2]);
        }
    }
}

// The same measurement by all 1024 threads of a workgroup, for lattices of at most 4096 sites (at most four per thread):
// where the overlap is not hidden behind a random-stream launch (it rides with the Houdayer seed workgroups, which wait
// for nothing else) its 256-thread form is the longer part of that launch.  Counts in the quad form of bitcount.hpp:
// a row of 16 lanes sums lanes of equal class (bytes of at most 4 x 16 = 64), the 64 rows file their sums in `part`
// and 96 threads add them up.  LDS: [N + 1] + 64 x 24 + 96 words.
__device__ __forceinline__ void overlap_block_1024(const SimDims &d, const DevPtrs &p, uint32_t bid, uint32_t *lds) {
    const uint32_t n1 = d.N + 1;
    uint32_t *Y = lds;
    uint32_t *part = lds + n1;        // [64][24]
    uint32_t *total = part + 64 * 24; // [96]
    uint32_t real, w;
    tile_of_block(d, bid, real, w);
    const uint32_t tid = threadIdx.x;
    const uint32_t q = d.n_temps >> 5, r = d.n_temps & 31u;
    const uint32_t wa = w + q, wb = w + q + 1;
    const uint32_t *Sg = p.S + static_cast<size_t>(real) * d.W * d.N;
    const uint32_t *A = Sg + static_cast<size_t>(w) * d.N;
    const uint32_t *B0 = wa < d.W ? Sg + static_cast<size_t>(wa) * d.N : nullptr;
    const uint32_t *B1 = (r && wb < d.W) ? Sg + static_cast<size_t>(wb) * d.N : nullptr;
    {
        uint32_t va[4], v0[4], v1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * 1024 + tid;
            va[k] = i < d.N ? A[i] : 0u;
            v0[k] = (B0 && i < d.N) ? B0[i] : 0u;
            v1[k] = (B1 && i < d.N) ? B1[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * 1024 + tid;
            if (i < d.N) Y[i] = va[k] ^ (r ? ((v0[k] >> r) | (v1[k] << (32 - r))) : v0[k]);
        }
        if (tid == 0) Y[d.N] = 0;
    }
    __syncthreads();
    uint32_t qa[4], qb[4], qc[4];
    if (d.hc) {
        const uint32_t *tab_g = p.hc_tab + static_cast<size_t>(real) * d.N;
        const uint32_t y_addr = lds_addr_of(Y);
        uint32_t ee[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pos = k * 1024 + tid;
            ee[k] = tab_g[pos < d.N ? pos : 0u];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live_site = k * 1024 + tid < d.N;
            const HcSite h = hc_decode<3>(d, ee[k], y_addr);
            const uint32_t y = lds_load(h.site);
            uint32_t z[3] = {0u, 0u, 0u};
#pragma unroll
            for (int dir = 0; dir < 3; ++dir)
                if (dir < static_cast<int>(d.nn)) z[dir] = y ^ lds_load(h.nb[2 * dir]);
            qa[k] = live_site ? y : 0u;
            qb[k] = live_site ? (z[0] ^ z[1] ^ z[2]) : 0u;
            qc[k] = live_site ? maj3(z[0], z[1], z[2]) : 0u;
        }
    } else {
        const uint4 *rec = p.rec + static_cast<size_t>(real) * n1;
        uint4 rr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pos = k * 1024 + tid;
            rr[k] = rec[pos < d.N ? pos : d.N];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live_site = k * 1024 + tid < d.N;
            const SiteRec sr(rr[k]);
            const uint32_t y = lds_word(Y, sr.site);
            uint32_t z[3];
#pragma unroll
            for (int dir = 0; dir < 3; ++dir)
                z[dir] = dir < static_cast<int>(d.nn) ? (y ^ lds_word(Y, sr.nb[2 * dir])) : 0u;
            qa[k] = live_site ? y : 0u;
            qb[k] = live_site ? (z[0] ^ z[1] ^ z[2]) : 0u;
            qc[k] = live_site ? maj3(z[0], z[1], z[2]) : 0u;
        }
    }
    {
        const uint32_t *in[3] = {qa, qb, qc};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            uint32_t pl3[3], pl5[5], c2[2];
            planes4(in[s], pl3);
            quad_add_planes<3>(pl3, pl5);
            spread_pair<5>(pl5, tid & 3u, c2);
#pragma unroll
            for (uint32_t jj = 0; jj < 2; ++jj) {
                uint32_t v = c2[jj];
                v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xF, 0xF, false));
                v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xF, 0xF, false));
                if ((tid & 12u) == 12u) part[(tid >> 4) * 24 + s * 8 + 2 * (tid & 3u) + jj] = v;
            }
        }
    }
    __syncthreads();
    if (tid < 96) { // thread = (set, slot); slot = j + 8 * byte of the sliced counters
        const uint32_t set = tid >> 5, slot = tid & 31u, j = slot & 7u, shift = (slot >> 3) * 8;
        uint32_t sum = 0;
#pragma unroll 8
        for (uint32_t g = 0; g < 64; ++g) sum += (part[g * 24 + set * 8 + j] >> shift) & 0xFFu;
        total[tid] = sum;
    }
    __syncthreads();
    if (tid < 32) {
        const uint32_t k = w * 32 + tid;
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if (k < d.n_sys && (rep & 1u) == 0 && rep + 1 < d.n_replicas) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(total[tid]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) - 2 * static_cast<int32_t>(total[32 + tid] + 2 * total[64 + tid]);
        }
    }
}

// Lattices with four to six forward directions (bcc, fcc): the same pair-difference word, neighbours
// from the wide table, and the up to six bond terms of a site counted as two groups of three.
__global__ __launch_bounds__(256) void k_overlap_wide(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    const uint32_t n1 = d.N + 1;
    uint32_t *Y = lds;
    uint32_t *scratch = lds + n1; // [5][80]
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t q = d.n_temps >> 5, r = d.n_temps & 31u;
    const uint32_t wa = w + q, wb = w + q + 1;
    const uint32_t *Sg = p.S + static_cast<size_t>(real) * d.W * d.N;
    const uint32_t *A = Sg + static_cast<size_t>(w) * d.N;
    const uint32_t *B0 = wa < d.W ? Sg + static_cast<size_t>(wa) * d.N : nullptr;
    const uint32_t *B1 = (r && wb < d.W) ? Sg + static_cast<size_t>(wb) * d.N : nullptr;
    for (uint32_t i = tid; i < d.N; i += 256) {
        const uint32_t v0 = B0 ? B0[i] : 0u, v1 = B1 ? B1[i] : 0u;
        Y[i] = A[i] ^ (r ? ((v0 >> r) | (v1 << (32 - r))) : v0);
    }
    if (tid == 0) Y[d.N] = 0; // the idle record's slot
    __syncthreads();
    uint32_t c[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    // the sites' records (LDS byte offsets of the site and its neighbours; rec_wide, schedule order -- the sums do not
    // care) are requested eight sites at a time before any is used: a thread's sixteen sites cost two memory round
    // trips, not sixteen
    const uint4 *rec = p.rec_wide + static_cast<size_t>(real) * 2 * n1;
    for (uint32_t base = 0; p.rec_wide && base < d.N; base += 16 * 256) {
        uint32_t qa[16], s1[16], c1[16], s2[16], c2[16];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 ra[8], rb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t pos = base + (half * 8 + k) * 256 + tid, at = pos < d.N ? pos : d.N;
                ra[k] = rec[2 * at];
                rb[k] = rec[2 * at + 1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const WideRec wr(ra[k], rb[k]);
                const uint32_t y = lds_word(Y, wr.site);
                uint32_t z[6];
#pragma unroll
                for (int dir = 0; dir < 6; ++dir) // 1 where q_i * q_k = -1 (an absent direction's slot is the site itself: 0)
                    z[dir] = y ^ lds_word(Y, wr.nb[2 * dir]);
                qa[half * 8 + k] = y;
                s1[half * 8 + k] = z[0] ^ z[1] ^ z[2];
                c1[half * 8 + k] = maj3(z[0], z[1], z[2]);
                s2[half * 8 + k] = z[3] ^ z[4] ^ z[5];
                c2[half * 8 + k] = maj3(z[3], z[4], z[5]);
            }
        }
        count16(qa, c[0]);
        count16(s1, c[1]);
        count16(c1, c[2]);
        count16(s2, c[3]);
        count16(c2, c[4]);
    }
    // (lattices whose offsets the 16-bit records cannot hold: neighbour indices from the wide table)
    for (uint32_t base = 0; !p.rec_wide && base < d.N; base += 16 * 256) {
        uint32_t qa[16], s1[16], c1[16], s2[16], c2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t pos = base + k * 256 + tid;
            qa[k] = s1[k] = c1[k] = s2[k] = c2[k] = 0;
            if (pos < d.N) {
                const uint32_t *row = p.nbr_wide + static_cast<size_t>(pos) * 16;
                const uint32_t y = Y[row[0]];
                uint32_t z[6];
#pragma unroll
                for (int dir = 0; dir < 6; ++dir)
                    z[dir] = dir < static_cast<int>(d.nn) ? (y ^ Y[row[1 + 2 * dir]]) : 0u;
                qa[k] = y;
                s1[k] = z[0] ^ z[1] ^ z[2];
                c1[k] = maj3(z[0], z[1], z[2]);
                s2[k] = z[3] ^ z[4] ^ z[5];
                c2[k] = maj3(z[3], z[4], z[5]);
            }
        }
        count16(qa, c[0]);
        count16(s1, c[1]);
        count16(c1, c[2]);
        count16(s2, c[3]);
        count16(c2, c[4]);
    }
    uint32_t tot[5];
    block_bit_totals<5>(c, scratch, tot);
    if (tid < 32) {
        const uint32_t k = w * 32 + tid;
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if (k < d.n_sys && (rep & 1u) == 0 && rep + 1 < d.n_replicas) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(tot[0]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) -
                            2 * static_cast<int32_t>(tot[1] + tot[3] + 2 * (tot[2] + tot[4]));
        }
    }
}

__global__ __launch_bounds__(256) void k_overlap(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds_dyn[];
    overlap_block(d, p, blockIdx.x, lds_dyn);
}
// (lattices of at most 4096 sites, where the launch stands alone on a chain: four sites per thread, not sixteen)
__global__ __launch_bounds__(1024) void k_overlap_1024(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds_dyn[];
    overlap_block_1024(d, p, blockIdx.x, lds_dyn);
}

// One launch, two kinds of workgroups: the first n_overlap blocks measure the overlap of the
// state the previous sweep left (latency-bound, few waves), the others draw the random stream
// of the next sweep (VALU-bound); the two do not depend on each other, so the measurement hides
// under the random stream instead of costing a launch of its own.
template <bool PRESHIFTED, bool GIBBS = false>
__global__ __launch_bounds__(256) void k_rng_overlap(SimDims d, DevPtrs p, uint32_t n_overlap) {
    extern __shared__ uint32_t lds_dyn[];
    if (blockIdx.x < n_overlap)
        overlap_block(d, p, blockIdx.x, lds_dyn);
    else
        rng_block<PRESHIFTED, GIBBS>(d, p, blockIdx.x - n_overlap, lds_dyn);
}

// ---------------------------------------------------------------------------
// k_measure: workgroup = tile (one disorder realisation), 1024 threads.  Fuses
// the three per-tile phases that follow a sweep in the reference's order:
//
//  (1) overlap  Replica 2p and 2p+1 at temperature t sit n_temps bit positions
//      apart, so Y = V ^ (V >> n_temps) lines every pair up in the bits of the
//      even replica.  V = S gives q_i; V = S ^ S(fwd_d) gives the bond products,
//      and q_i q_k = +1 <=> both replicas agree on the bond (overlap.rs:259-282).
//  (2) statistics  the reference's f64 running sums in its order: replica by
//      replica, pair by pair (mod.rs:555-578, overlap.rs:283-325, stats.rs:17-27).
//  (3) parallel tempering  one lane per replica ladder replays the reference's
//      event (mod.rs:766-792, tempering.rs:20-102, realization.rs:73-120) with
//      the generator of system r*n_temps; accepted swaps exchange adjacent bit
//      columns of the tile, and the slot cursors are permuted accordingly.
// ---------------------------------------------------------------------------
struct MeasureLds {
    uint32_t *scratch; // [17][48]
    int32_t *dot_spin; // [n_sys]
    int32_t *dot_link; // [n_sys]
    int32_t *E;        // [n_sys] by slot
    int32_t *M;        // [n_sys]
    uint32_t *sid;     // [n_sys] slot -> system
    uint32_t *sid0;    // [n_sys] before the event
    uint32_t *sof0;    // [n_sys] system -> slot before the event
    uint32_t *att;     // [n_edges]
    uint32_t *acc;     // [n_edges]
    uint32_t *masks;   // [2][W]
    uint32_t *flags;   // [1] any swap
};

__host__ __device__ inline size_t measure_lds_words(const SimDims &d) {
    return 17 * 48 + 7 * static_cast<size_t>(d.n_sys) +
           2 * static_cast<size_t>(d.n_edges) + 2 * static_cast<size_t>(d.W) + 4;
}

// Works for any workgroup size >= max(n_temps, n_replicas, 2 W) with n_sys <= 2 * blockDim.x.
template <bool STAMPS>
__device__ __forceinline__ void measure_block(const SimDims &d, const DevPtrs &p, const PtParams &pt,
                                              int do_accum, int accum_overlap, int do_pt,
                                              unsigned long long *stamps, uint32_t real, uint32_t *lds) {
    PA_STAMP(0)
    MeasureLds m;
    {
        uint32_t *q = lds;
        m.scratch = q; q += 17 * 48;
        m.dot_spin = reinterpret_cast<int32_t *>(q); q += d.n_sys;
        m.dot_link = reinterpret_cast<int32_t *>(q); q += d.n_sys;
        m.E = reinterpret_cast<int32_t *>(q); q += d.n_sys;
        m.M = reinterpret_cast<int32_t *>(q); q += d.n_sys;
        m.sid = q; q += d.n_sys;
        m.sid0 = q; q += d.n_sys;
        m.sof0 = q; q += d.n_sys;
        m.att = q; q += d.n_edges;
        m.acc = q; q += d.n_edges;
        m.masks = q; q += 2 * d.W;
        m.flags = q;
    }
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    int32_t *Eg = p.Esum + static_cast<size_t>(real) * d.n_sys;
    int32_t *Mg = p.Msum + static_cast<size_t>(real) * d.n_sys;
    uint32_t *sidg = p.system_ids + static_cast<size_t>(real) * d.n_sys;
    uint32_t *sofg = p.slot_of + static_cast<size_t>(real) * d.n_sys;

    for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
        m.E[k] = Eg[k];
        m.M[k] = Mg[k];
        const uint32_t sys = sidg[k];
        m.sid[k] = sys;
        m.sid0[k] = sys;
        m.sof0[k] = sofg[k];
    }
    for (uint32_t k = tid; k < d.n_edges; k += nthreads) {
        m.att[k] = 0;
        m.acc[k] = 0;
    }
    if (tid < 2 * d.W) m.masks[tid] = 0;
    if (tid == 0) m.flags[0] = 0;
    __syncthreads();
    PA_STAMP(1)

    // ---- (1) overlap dots come from k_overlap (previous launch on this stream) --------
    if ((accum_overlap || pt.equil) && d.n_pairs)
        for (uint32_t k = tid; k < d.n_pairs * d.n_temps; k += nthreads) {
            // (the most recent overlap launch: the host exchanged dot_* and dot_*_prev after it -- an overlap
            // block of THIS launch may be writing dot_* for the next sweep already)
            m.dot_spin[k] = p.dot_spin_prev[static_cast<size_t>(real) * d.n_pairs * d.n_temps + k];
            m.dot_link[k] = p.dot_link_prev[static_cast<size_t>(real) * d.n_pairs * d.n_temps + k];
        }
    __syncthreads();
    PA_STAMP(2)
    // ---- (2) running statistics ---------------------------------------------
    if ((do_accum || accum_overlap) && tid < d.n_temps) {
        const uint32_t t = tid;
        const float n_f = static_cast<float>(d.N);
        if (do_accum) {
        double *mom = p.mom + static_cast<size_t>(real) * 5 * d.n_temps;
        double m0 = 0.0, m1 = 0.0, m2s = 0.0, e0 = 0.0, e1 = 0.0;
        if (!(pt.fresh & 1u)) {
            m0 = mom[0 * d.n_temps + t], m1 = mom[1 * d.n_temps + t], m2s = mom[2 * d.n_temps + t];
            e0 = mom[3 * d.n_temps + t], e1 = mom[4 * d.n_temps + t];
        }
        double m2_ac = 0.0;
        for (uint32_t rep = 0; rep < d.n_replicas; ++rep) {
            const uint32_t slot = rep * d.n_temps + t;
            const float mag = __fdiv_rn(static_cast<float>(m.M[slot]), n_f);
            const float mag2 = __fmul_rn(mag, mag);
            m2_ac = __dadd_rn(m2_ac, static_cast<double>(mag2)); // mod.rs:567-571
            const float mag4 = __fmul_rn(mag2, mag2);
            const float energy = __fdiv_rn(energy_sum(m.E[slot], d.general), n_f);
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
        if (pt.ac_index) // the value pushed to the m^2 autocorrelation accumulator (mod.rs:580-586)
            p.ac_m2[(static_cast<size_t>(real) * pt.ac_stride + (pt.ac_index - 1)) * d.n_temps + t] =
                static_cast<float>(__dmul_rn(m2_ac, __ddiv_rn(1.0, static_cast<double>(d.n_replicas))));
        }
        if (accum_overlap && d.n_pairs) {
            double *ovm = p.ovm + static_cast<size_t>(real) * 6 * d.n_temps;
            double o[6];
            const uint32_t row_words = (d.N + 32) / 32;
            uint32_t *seen_row = p.hist_touched + (static_cast<size_t>(real) * d.n_temps + t) * row_words;
            if (pt.fresh & 2u) { // first overlap accumulation of the run: sums from zero, no bin seen yet
                for (int k = 0; k < 6; ++k) o[k] = 0.0;
                for (uint32_t wq = 0; wq < row_words; ++wq) seen_row[wq] = 0u;
            } else {
                for (int k = 0; k < 6; ++k) o[k] = ovm[k * d.n_temps + t];
            }
            const float n_bonds_f = static_cast<float>(d.N * d.nn);
            const size_t hbase = (static_cast<size_t>(real) * d.n_temps + t) * (d.N + 1);
            double q2_ac = 0.0;
            for (uint32_t pr = 0; pr < d.n_pairs; ++pr) {
                const int32_t dot_spin = m.dot_spin[pr * d.n_temps + t];
                const int32_t dot_link = m.dot_link[pr * d.n_temps + t];
                const float ql = __fdiv_rn(static_cast<float>(dot_link), n_bonds_f);
                const float qv = __fdiv_rn(static_cast<float>(dot_spin), n_f);
                const float q2 = __fmul_rn(qv, qv);
                const float q4 = __fmul_rn(q2, q2);
                q2_ac = __dadd_rn(q2_ac, static_cast<double>(q2)); // overlap.rs:310-314
                const float ql2 = __fmul_rn(ql, ql);
                const float ql4 = __fmul_rn(ql2, ql2);
                o[0] = __dadd_rn(o[0], static_cast<double>(qv));
                o[1] = __dadd_rn(o[1], static_cast<double>(q2));
                o[2] = __dadd_rn(o[2], static_cast<double>(q4));
                o[3] = __dadd_rn(o[3], static_cast<double>(ql));
                o[4] = __dadd_rn(o[4], static_cast<double>(ql2));
                o[5] = __dadd_rn(o[5], static_cast<double>(ql4));
                const uint32_t qbin = static_cast<uint32_t>((dot_spin + static_cast<int32_t>(d.N)) / 2);
                const size_t bin = hbase + qbin;
                // row t of the bitmap belongs to this thread alone; all four loads are in flight together
                uint32_t *seen = seen_row + (qbin >> 5);
                const uint32_t seen_word = *seen;
                const uint64_t h_old = p.hist[bin];
                const double a_old = p.qlq[bin], b_old = p.ql2q[bin];
                const bool fresh = ((seen_word >> (qbin & 31u)) & 1u) == 0; // first value of this run: the old
                p.hist[bin] = (fresh ? 0ull : h_old) + 1;                   // contents are from an earlier run
                p.qlq[bin] = __dadd_rn(fresh ? 0.0 : a_old, static_cast<double>(ql));
                p.ql2q[bin] = __dadd_rn(fresh ? 0.0 : b_old, static_cast<double>(ql2));
                if (fresh) *seen = seen_word | (1u << (qbin & 31u));
            }
            for (int k = 0; k < 6; ++k) ovm[k * d.n_temps + t] = o[k];
            if (pt.ac_index_q) // mod.rs:588-594
                p.ac_q2[(static_cast<size_t>(real) * pt.ac_stride + (pt.ac_index_q - 1)) * d.n_temps + t] =
                    static_cast<float>(__dmul_rn(q2_ac, __ddiv_rn(1.0, static_cast<double>(d.n_pairs))));
        }
    }
    // ---- equilibration diagnostic (mod.rs:511-541, overlap.rs:285-287,327-332, equilibration.rs:40-58)
    if (pt.equil && tid < d.n_temps) {
        const uint32_t t = tid;
        const float n_f = static_cast<float>(d.N);
        float de = 0.0f;
        for (uint32_t rep = 0; rep < d.n_replicas; ++rep)
            de = __fadd_rn(de, __fdiv_rn(energy_sum(m.E[rep * d.n_temps + t], d.general), n_f));
        de = __fmul_rn(de, __fdiv_rn(1.0f, static_cast<float>(d.n_replicas)));
        float dq = 0.0f;
        if (d.n_pairs) {
            const float n_bonds_f = static_cast<float>(d.N * d.nn);
            for (uint32_t pr = 0; pr < d.n_pairs; ++pr)
                dq = __fadd_rn(dq, __fdiv_rn(static_cast<float>(m.dot_link[pr * d.n_temps + t]), n_bonds_f));
            dq = __fmul_rn(dq, __fdiv_rn(1.0f, static_cast<float>(d.n_pairs)));
        }
        double *sum = p.eq_sum + static_cast<size_t>(real) * 2 * d.n_temps;
        const double se = __dadd_rn(sum[t], static_cast<double>(de));
        const double sq = __dadd_rn(sum[d.n_temps + t], static_cast<double>(dq));
        sum[t] = se;
        sum[d.n_temps + t] = sq;
        if (pt.equil_ckpt) {
            double *ck = p.eq_ckpt + ((static_cast<size_t>(real) * 64 + (pt.equil_ckpt - 1)) * 2) * d.n_temps;
            const double c = static_cast<double>(pt.equil_count);
            ck[t] = __ddiv_rn(se, c);
            ck[d.n_temps + t] = __ddiv_rn(sq, c);
        }
    }
    PA_STAMP(3)
    if (!do_pt || d.n_temps < 2) return;
    __syncthreads();

    // ---- (3) parallel tempering -------------------------------------------------
    // single_random_edge: one lane per replica ladder draws the edge and decides it.  full_ladder: the ladder's lane
    // draws its n_temps - 1 uniforms in order (they are consumed one per edge in the sequence "edges of parity p
    // ascending, then parity 1 - p", tempering.rs:59-70), then the edges of one parity are decided side by side --
    // they touch disjoint slots and systems -- and the other parity follows after a barrier on the exchanged energies.
    DevRng rng;
    rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    uint64_t *trips = p.round_trips + static_cast<size_t>(real) * d.n_sys;
    uint8_t *trip = p.trip_state + static_cast<size_t>(real) * d.n_sys;
    const float n_f_pt = static_cast<float>(d.N);
    const uint32_t n_edges = d.n_temps - 1;
    auto attempt = [&](uint32_t offset, uint32_t edge, uint32_t round, uint32_t draw) {
        const uint32_t sl = offset + edge, sr = sl + 1;
        const uint32_t left = m.sid[sl], right = m.sid[sr];
        const bool accepted = pt_accept(n_f_pt, m.E[sl], m.E[sr], d.general, p.dbeta[edge], p.ln_table[draw]);
        atomicAdd(&m.att[edge], 1u);
        if (accepted) {
            atomicAdd(&m.acc[edge], 1u);
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
            m.sid[sl] = right;
            m.sid[sr] = left;
            const int32_t te = m.E[sl];
            m.E[sl] = m.E[sr];
            m.E[sr] = te;
            const int32_t tm = m.M[sl];
            m.M[sl] = m.M[sr];
            m.M[sr] = tm;
            atomicOr(&m.masks[round * d.W + (sl >> 5)], 1u << (sl & 31u));
            m.flags[0] = 1;
        }
    };
    if (tid < d.n_replicas) {
        uint32_t st[8];
        load_cursor(p.cursor, real, d.SP, m.sof0[tid * d.n_temps], st);
        rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
        rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
        rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
        rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
    }
    if (pt.schedule == 0) {
        if (tid < d.n_replicas) {
            // rand 0.8.5 gen_range(0..n_temps-1) on usize: widening multiply + zone
            const uint64_t range = n_edges;
            const uint64_t zone = (range << __clzll(range)) - 1;
            uint32_t edge;
            for (;;) {
                const uint64_t v = rng.next();
                if (v * range <= zone) {
                    edge = static_cast<uint32_t>(__umul64hi(v, range));
                    break;
                }
            }
            attempt(tid * d.n_temps, edge, 0u, static_cast<uint32_t>(rng.next() >> 40));
        }
    } else {
        uint32_t *pt_draw = reinterpret_cast<uint32_t *>(m.dot_spin); // [n_replicas][n_edges] (the dots are spent)
        if (tid < d.n_replicas)
            for (uint32_t a = 0; a < n_edges; ++a) pt_draw[tid * n_edges + a] = static_cast<uint32_t>(rng.next() >> 40);
        __syncthreads();
        const uint32_t n_first = (n_edges + 1 - pt.first_parity) / 2; // edges of parity p come first
        for (uint32_t round = 0; round < 2; ++round) {
            for (uint32_t idx = tid; idx < d.n_replicas * n_edges; idx += nthreads) {
                const uint32_t rep = idx / n_edges, a = idx % n_edges;
                if ((a >= n_first) != (round == 1)) continue;
                const uint32_t edge = round == 0 ? pt.first_parity + 2 * a : (1 - pt.first_parity) + 2 * (a - n_first);
                attempt(rep * d.n_temps, edge, round, pt_draw[idx]);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    PA_STAMP(4)
    for (uint32_t k = tid; k < d.n_edges; k += nthreads) {
        if (m.att[k]) p.edge_att[static_cast<size_t>(real) * d.n_edges + k] += m.att[k];
        if (m.acc[k]) p.edge_acc[static_cast<size_t>(real) * d.n_edges + k] += m.acc[k];
    }
    const bool any_swap = m.flags[0] != 0;
    if (pt.relay) {
        // The random-stream workgroups of this launch replay the event and take every slot's generator from the
        // slot its system comes from (rng_block), so nothing is permuted here; the new permutation goes to the
        // second pair of tables (those workgroups still read the old one), the energies are recomputed by the
        // sweep that follows before anything reads them.
        for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
            p.system_ids_out[static_cast<size_t>(real) * d.n_sys + k] = m.sid[k];
            p.slot_of_out[static_cast<size_t>(real) * d.n_sys + m.sid[k]] = k;
        }
        if (tid < 2 * d.W) p.swap_masks[static_cast<size_t>(real) * 2 * d.W + tid] = m.masks[tid];
        if (tid == 0 && any_swap) p.swap_epoch[real] += 1;
        // The cached sums must follow the systems too, but Esum is being read by those workgroups (they replay the
        // event from it): the exchanged sums are staged, and k_adopt_relay_sums files them behind this launch where
        // the sweep that follows does not recompute them anyway (pt_interval > 1 inside the warm-up).
        if (any_swap)
            for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
                p.Esum_relay[static_cast<size_t>(real) * d.n_sys + k] = m.E[k];
                p.Msum_relay[static_cast<size_t>(real) * d.n_sys + k] = m.M[k];
            }
        if (tid == 0) p.relay_swapped[real] = any_swap ? 1u : 0u;
        return;
    }
    // cursors follow their systems: new cursor[slot] = old cursor[old slot of the system now there].
    // All loads complete (registers) before any store, so the permutation needs no ordering.
    // (a thread owns up to two slots: n_sys <= 2 * blockDim.x, checked on the host)
    uint32_t moved[2][8];
    if (any_swap) {
#pragma unroll
        for (uint32_t it = 0; it < 2; ++it) {
            const uint32_t slot = tid + it * nthreads;
            if (slot < d.n_sys && m.sid[slot] != m.sid0[slot]) {
                const uint32_t src = m.sof0[m.sid[slot]];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                    moved[it][j] = p.cursor[(static_cast<size_t>(real) * 8 + j) * d.SP + src];
            }
        }
    }
    __syncthreads();
    if (any_swap) {
#pragma unroll
        for (uint32_t it = 0; it < 2; ++it) {
            const uint32_t slot = tid + it * nthreads;
            if (slot < d.n_sys && m.sid[slot] != m.sid0[slot]) {
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j)
                    p.cursor[(static_cast<size_t>(real) * 8 + j) * d.SP + slot] = moved[it][j];
            }
        }
        for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
            Eg[k] = m.E[k];
            Mg[k] = m.M[k];
            sidg[k] = m.sid[k];
            sofg[m.sid[k]] = k;
        }
    }
    __syncthreads();
    // the ladder's generator state goes to wherever system r*n_temps now sits
    if (tid < d.n_replicas) {
        const uint32_t offset = tid * d.n_temps;
        uint32_t final_slot = m.sof0[offset];
        for (uint32_t k = offset; k < offset + d.n_temps; ++k)
            if (m.sid[k] == offset) final_slot = k;
        uint32_t st[8];
        st[0] = static_cast<uint32_t>(rng.s0);
        st[1] = static_cast<uint32_t>(rng.s0 >> 32);
        st[2] = static_cast<uint32_t>(rng.s1);
        st[3] = static_cast<uint32_t>(rng.s1 >> 32);
        st[4] = static_cast<uint32_t>(rng.s2);
        st[5] = static_cast<uint32_t>(rng.s2 >> 32);
        st[6] = static_cast<uint32_t>(rng.s3);
        st[7] = static_cast<uint32_t>(rng.s3 >> 32);
        store_cursor(p.cursor, real, d.SP, final_slot, st);
    }
    PA_STAMP(5)
    // The accepted swaps exchange adjacent bit columns (k, k+1) of the tile, round 0 then round 1.
    // The exchange itself is deferred: the masks are published and the next k_sweep applies them
    // while it loads its word (k_apply_swaps materialises them for every other reader).
    if (tid < 2 * d.W) p.swap_masks[static_cast<size_t>(real) * 2 * d.W + tid] = m.masks[tid];
    if (tid == 0 && any_swap) p.swap_epoch[real] += 1;
    PA_STAMP(6)
}

template <bool STAMPS>
__global__ __launch_bounds__(1024) void k_measure(SimDims d, DevPtrs p, PtParams pt, int do_overlap,
                                                  int do_accum, int accum_overlap, int do_pt,
                                                  unsigned long long *stamps) {
    extern __shared__ uint32_t lds_dyn[];
    (void)do_overlap;
    measure_block<STAMPS>(d, p, pt, do_accum, accum_overlap, do_pt, stamps, blockIdx.x, lds_dyn);
}

// One launch, three kinds of workgroups: [0, n_overlap) measure the overlap of the state the previous sweep left,
// [n_overlap, n_overlap + n_measure) are that sweep's measurement blocks (statistics + tempering event, one per
// realisation, PtParams::relay), the rest draw the random stream of the next sweep -- with REPLAY they apply the
// event's cursor permutation themselves.  A sweep of the plain protocol is then two launches (this one and the
// sweep kernel) instead of three, and the latency-bound measurement hides under the VALU-bound random stream.
// Workgroups have 64 * SimDims::rng_waves threads (256 or 512): the random-stream waves of a workgroup share one jump
// table; an overlap block is written for 256 threads and retires the others at once (a retired wave no longer counts
// at the workgroup's barriers), a measurement block takes any size.
template <bool PRESHIFTED, int REPLAY, bool GIBBS = false>
__global__ __launch_bounds__(512) void k_rng_relay(SimDims d, DevPtrs p, uint32_t n_overlap, uint32_t n_measure, PtParams pt,
                                                   int do_accum, int accum_overlap, int do_pt) {
    extern __shared__ uint32_t lds_dyn[];
    // Block order = dispatch order: the stream workgroups FIRST, the riders last.  A CU holds four stream workgroups and
    // the bench ensemble's 1536 are a round and a half of the chip's 1024 slots, so the riders -- latency-bound, no use
    // for the vector units -- start in the half-empty second round instead of holding a third of the first one (round 3
    // had them first: k_rng_relay 44.7 us against 39.6 us for the stream workgroups alone).  Nothing in the launch waits
    // for a rider: the stream workgroups replay the tempering event themselves.
    const uint32_t n_stream = gridDim.x - n_overlap - n_measure;
    if (blockIdx.x < n_stream)
        rng_block<PRESHIFTED, GIBBS, REPLAY>(d, p, blockIdx.x, lds_dyn, pt.first_parity);
    else if (blockIdx.x < n_stream + n_measure)
        measure_block<false>(d, p, pt, do_accum, accum_overlap, do_pt, nullptr, blockIdx.x - n_stream, lds_dyn);
    else {
        if (threadIdx.x >= 256) return;
        overlap_block(d, p, blockIdx.x - n_stream - n_measure, lds_dyn);
    }
}

// ---------------------------------------------------------------------------
// k_houdayer: Houdayer isoenergetic cluster move, group size 2, single-cluster
// ("wolff") variant -- clusters/overlap.rs:35-56 (build_tasks) and :153-339.
// Workgroup = one realisation, all temperatures and pairs at once:
//   * per temperature the replicas are shuffled with pair generator t*n_pairs
//     (reverse Fisher-Yates, rand's u32 gen_range) and grouped consecutively;
//     group g then draws from pair generator t*n_pairs + g;
//   * A = "active" mask per slot (the pair's spins differ, i.e. they sum to
//     zero); both slots of a pair carry the same bit;
//   * every pair draws its seed with up to 64 probes gen_range(0..N)
//     (utils.rs:107-119), then all clusters grow together,
//     F <- A & (F | F(neighbours)), until nothing changes.  The reference's DFS
//     visits exactly the connected component of the seed in the graph "both
//     ends active", whatever the visiting order, so the result is identical;
//   * both replicas are flipped on their cluster: S ^= F.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dev_gen_range_u32(DevRng &rng, uint32_t range) {
    const uint32_t zone = (range << __clz(range)) - 1u;
    for (;;) {
        const uint32_t v = static_cast<uint32_t>(rng.next() >> 32);
        const uint64_t m = static_cast<uint64_t>(v) * range;
        if (static_cast<uint32_t>(m) <= zone) return static_cast<uint32_t>(m >> 32);
    }
}

__device__ __forceinline__ uint64_t dev_gen_range_u64(DevRng &rng, uint64_t range) {
    const uint64_t zone = (range << __clzll(range)) - 1ull;
    for (;;) {
        const uint64_t v = rng.next();
        if (v * range <= zone) return __umul64hi(v, range);
    }
}

// Two launches: k_houdayer (workgroup = realisation) pairs the replicas, builds the active masks and plants the
// seeds; k_houdayer_grow (workgroup = (realisation, tile word)) grows and flips.  A slot's cluster grows inside that
// slot's bit column only -- both slots of a pair carry the same mask and the same seed -- so the words of a tile are
// independent, and the growth, a chain of as many rounds as the largest cluster's diameter, runs on three times as
// many CUs with a third of the sites per round (94 -> 68 us per call at 12^3 x 128 samples).
__host__ __device__ inline size_t houdayer_lds_words(const SimDims &d) {
    return 3 * static_cast<size_t>(d.N) * d.W + 2 * static_cast<size_t>(d.n_temps) * d.n_pairs +
           d.n_sys + 4 + static_cast<size_t>(d.n_replicas) * d.W;
}

// Active masks of one site for a tile of NW words (registers): partners sit a multiple of n_temps bit positions
// apart (same temperature, another replica), so for every distance class c the whole tile value is shifted by
// c * n_temps bits once, XORed with itself and masked with the slots whose partner lies that far up (cmask[c]);
// the result is the pair's disagreement at the lower slot, and shifted back, at the upper one.
template <int NW>
__device__ __forceinline__ void houdayer_active_site(const uint32_t *tile, uint32_t *A, const uint32_t *cmask, uint32_t N,
                                                     uint32_t n_temps, uint32_t n_replicas, uint32_t i) {
    uint32_t V[NW], out[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        V[w] = tile[w * N + i];
        out[w] = 0;
    }
    for (uint32_t c = 1; c < n_replicas; ++c) {
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) any |= cmask[c * NW + w];
        if (!any) continue; // (uniform over the workgroup)
        const uint32_t delta = c * n_temps, wo = delta >> 5, bo = delta & 31u;
        uint32_t X[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { // bits of V, delta positions up, brought down to w
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                if (static_cast<uint32_t>(u) == w + wo) lo = V[u];
                if (static_cast<uint32_t>(u) == w + wo + 1) hi = V[u];
            }
            const uint32_t down = bo ? (lo >> bo) | (hi << (32u - bo)) : lo;
            X[w] = (V[w] ^ down) & cmask[c * NW + w];
            out[w] |= X[w];
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) { // and the same bits at the partners' positions
            uint32_t lo = 0, lower = 0;
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                if (static_cast<uint32_t>(u) + wo == static_cast<uint32_t>(w)) lo = X[u];
                if (static_cast<uint32_t>(u) + wo + 1 == static_cast<uint32_t>(w)) lower = X[u];
            }
            out[w] |= bo ? (lo << bo) | (lower >> (32u - bo)) : lo;
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) A[w * N + i] = out[w];
}

__device__ __forceinline__ void houdayer_block(const SimDims &d, const DevPtrs &p, uint32_t real, uint32_t *lds) {
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    uint32_t *tile = lds;                      // [W][N] spins
    uint32_t *A = lds + tile_words;            // [W][N] active mask
    uint32_t *F = lds + 2 * tile_words;        // [W][N] cluster membership (seeds)
    uint32_t *pair_slots = lds + 3 * tile_words; // [n_temps * n_pairs][2]
    uint32_t *partner = pair_slots + 2 * d.n_temps * d.n_pairs; // [n_sys]
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t *Sg = p.S + static_cast<size_t>(real) * tile_words;
    uint32_t *Ag = p.hd_scratch + static_cast<size_t>(real) * 2 * tile_words; // active masks, then the seeds
    uint32_t *Fg = Ag + tile_words;
    const uint32_t n_lanes = d.n_temps * d.n_pairs;
    uint64_t *prng = p.pair_rng + static_cast<size_t>(real) * n_lanes * 4;

    copy_to_lds(tile, Sg, static_cast<uint32_t>(tile_words), tid, nthreads);
    for (size_t i = tid; i < tile_words; i += nthreads) F[i] = 0;
    for (uint32_t k = tid; k < d.n_sys; k += nthreads) partner[k] = k; // unpaired: never active
    // The sums the sweep left, kept for the measurement that rides in the growth launch (k_houdayer_grow_measure): the
    // growth workgroups of that launch replace Esum / Msum with the sums after the move.
    for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
        p.Esum_pre[static_cast<size_t>(real) * d.n_sys + k] = p.Esum[static_cast<size_t>(real) * d.n_sys + k];
        p.Msum_pre[static_cast<size_t>(real) * d.n_sys + k] = p.Msum[static_cast<size_t>(real) * d.n_sys + k];
    }
    __syncthreads();

    // ---- pairing: lane (t, 0) shuffles the replicas at temperature t (overlap.rs:44-49).
    // The shuffle moves positions, so shuffling replica indices equals shuffling the systems.
    DevRng rng;
    rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    if (tid < n_lanes) {
        const uint32_t t = tid / d.n_pairs, g = tid % d.n_pairs;
        const uint64_t *st = prng + static_cast<size_t>(tid) * 4;
        rng.s0 = st[0];
        rng.s1 = st[1];
        rng.s2 = st[2];
        rng.s3 = st[3];
        if (g == 0) {
            uint32_t order[32]; // n_replicas <= 32 (checked on the host)
            for (uint32_t k = 0; k < d.n_replicas; ++k) order[k] = k;
            for (uint32_t i = d.n_replicas - 1; i >= 1; --i) {
                const uint32_t j = dev_gen_range_u32(rng, i + 1);
                const uint32_t tmp = order[i];
                order[i] = order[j];
                order[j] = tmp;
            }
            for (uint32_t gg = 0; gg < d.n_pairs; ++gg) {
                const uint32_t sa = order[2 * gg] * d.n_temps + t;
                const uint32_t sb = order[2 * gg + 1] * d.n_temps + t;
                pair_slots[2 * (t * d.n_pairs + gg) + 0] = sa;
                pair_slots[2 * (t * d.n_pairs + gg) + 1] = sb;
                partner[sa] = sb;
                partner[sb] = sa;
            }
        }
    }
    __syncthreads();

    // ---- active mask: bit k of A = spin(k) != spin(partner(k))
    uint32_t *cmask = partner + d.n_sys + 4; // [n_replicas][W] slots whose partner sits c * n_temps positions up
    if (d.W <= 4) { // the tile value of a site fits a few registers: whole-word form
        for (uint32_t k = tid; k < d.n_replicas * d.W; k += nthreads) cmask[k] = 0;
        __syncthreads();
        for (uint32_t k = tid; k < d.n_sys; k += nthreads) {
            const uint32_t q = partner[k];
            if (q > k) atomicOr(&cmask[((q - k) / d.n_temps) * d.W + (k >> 5)], 1u << (k & 31u));
        }
        __syncthreads();
        for (uint32_t i = tid; i < d.N; i += nthreads) {
            switch (d.W) {
            case 1: houdayer_active_site<1>(tile, A, cmask, d.N, d.n_temps, d.n_replicas, i); break;
            case 2: houdayer_active_site<2>(tile, A, cmask, d.N, d.n_temps, d.n_replicas, i); break;
            case 3: houdayer_active_site<3>(tile, A, cmask, d.N, d.n_temps, d.n_replicas, i); break;
            default: houdayer_active_site<4>(tile, A, cmask, d.N, d.n_temps, d.n_replicas, i); break;
            }
        }
    } else
    for (uint32_t i = tid; i < d.N; i += nthreads) {
        for (uint32_t w = 0; w < d.W; ++w) {
            uint32_t a = 0;
            const uint32_t v = tile[w * d.N + i];
            const uint32_t top = min(32u, d.n_sys - w * 32);
            for (uint32_t b = 0; b < top; ++b) {
                const uint32_t q = partner[w * 32 + b];
                const uint32_t pv = (tile[(q >> 5) * d.N + i] >> (q & 31u)) & 1u;
                a |= (((v >> b) & 1u) ^ pv) << b;
            }
            A[w * d.N + i] = a;
        }
    }
    __syncthreads();

    // ---- seeds (utils.rs:107-119): up to 64 probes; no active probe -> this pair does nothing
    if (tid < n_lanes) {
        const uint32_t sa = pair_slots[2 * tid], sb = pair_slots[2 * tid + 1];
        for (int probe = 0; probe < 64; ++probe) {
            const uint32_t i = static_cast<uint32_t>(dev_gen_range_u64(rng, d.N));
            if ((A[(sa >> 5) * d.N + i] >> (sa & 31u)) & 1u) {
                atomicOr(&F[(sa >> 5) * d.N + i], 1u << (sa & 31u));
                atomicOr(&F[(sb >> 5) * d.N + i], 1u << (sb & 31u));
                break;
            }
        }
        uint64_t *st = prng + static_cast<size_t>(tid) * 4;
        st[0] = rng.s0;
        st[1] = rng.s1;
        st[2] = rng.s2;
        st[3] = rng.s3;
    }
    __syncthreads();
    for (size_t i = tid; i < tile_words; i += nthreads) {
        Ag[i] = A[i];
        Fg[i] = F[i];
    }
}

__global__ __launch_bounds__(1024) void k_houdayer(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    houdayer_block(d, p, blockIdx.x, lds);
}

// The same launch with the overlap measurement of the state the sweep left riding in it: this kernel only READS the
// spins (masks and seeds go to scratch; the growth kernel that follows flips), so the overlap blocks -- the first four
// waves of their workgroups -- see the state between the sweep and the move, as the reference measures it.
__global__ __launch_bounds__(1024) void k_houdayer_overlap(SimDims d, DevPtrs p, uint32_t n_houdayer) {
    extern __shared__ uint32_t lds[];
    if (blockIdx.x < n_houdayer) {
        houdayer_block(d, p, blockIdx.x, lds);
    } else if (d.N <= 4096) {
        overlap_block_1024(d, p, blockIdx.x - n_houdayer, lds);
    } else {
        if (threadIdx.x >= 256) return; // (a barrier waits for the surviving waves only)
        overlap_block(d, p, blockIdx.x - n_houdayer, lds);
    }
}

// grow every cluster of one tile word to its connected component inside the active graph, then flip
// with_energy (lattices with the schedule table, SimDims::hc): the workgroup also leaves the energies and
// magnetisations of its word's 32 slots after the move (mod.rs:749-756 recomputes them before the tempering event) --
// the flipped word is in the LDS anyway, so this replaces an energy-only launch of the sweep kernel.  A word without a
// seed changes nothing and keeps the sums the sweep left.
__device__ __forceinline__ void houdayer_grow_block(const SimDims &d, const DevPtrs &p, int with_energy, uint32_t bid, uint32_t *lds) {
    uint32_t *A = lds;             // [N] active mask of the word
    uint32_t *F = lds + d.N;       // [N] cluster membership
    uint32_t *changed = F + d.N;   // [3] (+ 1 spare)
    uint32_t *sw = changed + 4;    // [N + 1] the word after the move (with_energy)
    uint32_t *scratch = sw + d.N + 4; // [5][48]
    const uint32_t real = bid / d.W, w = bid % d.W;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    const uint32_t *Ag = p.hd_scratch + static_cast<size_t>(real) * 2 * tile_words + static_cast<size_t>(w) * d.N;
    const uint32_t *Fg = Ag + tile_words;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * tile_words + static_cast<size_t>(w) * d.N;
    const uint4 *rec = p.rec + static_cast<size_t>(real) * (d.N + 1);
    bool any = false;
    for (uint32_t i = tid; i < d.N; i += nthreads) {
        A[i] = Ag[i];
        const uint32_t f = Fg[i];
        F[i] = f;
        any = any || f != 0;
    }
    if (tid == 0) changed[0] = 0;
    __syncthreads();
    if (any) changed[0] = 1;
    __syncthreads();
    if (changed[0] == 0) return; // no seed in this word: nothing grows, nothing flips
    __syncthreads();
    if (tid == 0) changed[0] = 0;
    __syncthreads();
    // Growth is monotone (a bit is only ever set, and only where a path of active sites leads to the seed), so the
    // passes need no barrier between them: a pass may read its neighbours before or after their update, the fixed
    // point is the same.  Convergence is checked every four passes.
    // a thread visits the same (up to four) sites in every pass: their records are fetched once, not once per pass
    constexpr int kCached = 4;
    const bool cached = d.N <= kCached * nthreads;
    uint4 mine[kCached];
    // (in SITE order, from the neighbour table: consecutive lanes then read consecutive words of F -- in schedule order,
    // along the hyperplanes of the sweep, half of the kernel's LDS cycles were bank conflicts; the growth has no order)
#pragma unroll
    for (int k = 0; k < kCached; ++k) {
        const uint32_t site = tid + k * nthreads;
        mine[k] = make_uint4(0, 0, 0, 0);
        if (cached && site < d.N) {
            const NbrRow row(p.nbr_site, site);
            mine[k] = make_uint4((site * 4) | (row.at(0) * 4 << 16), (row.at(1) * 4) | (row.at(2) * 4 << 16),
                                 (row.at(3) * 4) | (row.at(4) * 4 << 16), row.at(5) * 4);
        }
    }
    uint32_t act[kCached]; // (the active masks of a thread's sites do not change: registers)
#pragma unroll
    for (int k = 0; k < kCached; ++k) act[k] = (cached && tid + k * nthreads < d.N) ? A[tid + k * nthreads] : 0u;
    auto grow_site = [&](const uint4 raw, uint32_t a, bool &grew) {
        const SiteRec r(raw);
        const uint32_t f = lds_word(F, r.site);
        uint32_t g = f;
#pragma unroll
        for (int k = 0; k < 6; ++k) g |= lds_word(F, r.nb[k]);
        g &= a;
        g |= f;
        if (g != f) {
            F[r.site >> 2] = g;
            grew = true;
        }
    };
    // (one barrier per convergence check, three flags in turn: the flag of check i + 1 is cleared before the barrier of
    // check i -- whoever clears it has passed the barrier of check i - 1, behind which every thread had read it last,
    // as the flag of check i - 2)
    if (tid < 3) changed[tid] = 0;
    __syncthreads();
    for (uint32_t iter = 0; iter <= d.N; ++iter) {
        bool grew = false;
        for (int pass = 0; pass < 4; ++pass) {
            if (cached) {
#pragma unroll
                for (int k = 0; k < kCached; ++k)
                    if (tid + k * nthreads < d.N) grow_site(mine[k], act[k], grew);
            } else {
                for (uint32_t pos = tid; pos < d.N; pos += nthreads) {
                    const uint4 raw = rec[pos];
                    grow_site(raw, lds_word(A, SiteRec(raw).site), grew);
                }
            }
        }
        if (grew) changed[iter % 3u] = 1;
        if (tid == 0) changed[(iter + 1u) % 3u] = 0;
        __syncthreads();
        if (changed[iter % 3u] == 0) break;
    }
    // ---- flip both replicas of every pair on its cluster
    if (!with_energy) {
        for (uint32_t i = tid; i < d.N; i += nthreads) {
            const uint32_t f = F[i];
            if (f) Sg[i] ^= f;
        }
        return;
    }
    for (uint32_t i = tid; i < d.N; i += nthreads) {
        const uint32_t f = F[i], v = Sg[i] ^ f;
        sw[i] = v;
        if (f) Sg[i] = v;
    }
    __syncthreads();
    if (tid >= 256) return; // (the sums are a 256-thread role; a barrier waits for the surviving waves only)
    const uint32_t *tab = p.hc_tab + static_cast<size_t>(real) * d.N;
    switch (d.nn) {
    case 1: hc_energy<1>(d, p, real, w, tab, lds_addr_of(sw), scratch, p.Esum, p.Msum); break;
    case 2: hc_energy<2>(d, p, real, w, tab, lds_addr_of(sw), scratch, p.Esum, p.Msum); break;
    default: hc_energy<3>(d, p, real, w, tab, lds_addr_of(sw), scratch, p.Esum, p.Msum); break;
    }
}

__global__ __launch_bounds__(1024) void k_houdayer_grow(SimDims d, DevPtrs p, int with_energy) {
    extern __shared__ uint32_t lds[];
    houdayer_grow_block(d, p, with_energy, blockIdx.x, lds);
}

// The growth launch with the sweep's measurement blocks riding in it (statistics of the swept state and of the
// overlap the seed launch measured; no tempering event).  They read the sums the sweep left from Esum_pre / Msum_pre,
// where the seed launch copied them: the growth workgroups of this very launch replace Esum / Msum.  One launch less
// on a chain of five per step.
__global__ __launch_bounds__(1024) void k_houdayer_grow_measure(SimDims d, DevPtrs p, int with_energy, uint32_t n_grow, PtParams pt,
                                                                int do_accum) {
    extern __shared__ uint32_t lds[];
    if (blockIdx.x < n_grow) {
        houdayer_grow_block(d, p, with_energy, blockIdx.x, lds);
    } else {
        DevPtrs q = p;
        q.Esum = p.Esum_pre;
        q.Msum = p.Msum_pre;
        measure_block<false>(d, q, pt, do_accum, 1, 0, nullptr, blockIdx.x - n_grow, lds);
    }
}

// The riders of the overlap moves that keep their launches to themselves (Joerg, CMR, every union-find form).
// k_overlap_pairing: the overlap of the swept state, and per realisation a block that (a) shuffles the replicas of every
// temperature (build_tasks, overlap.rs:35-56: what k_pairing does in a launch of its own) and (b) puts the sweep's sums
// aside in Esum_pre / Msum_pre for the statistics, which then ride in the launch that recomputes the energies AFTER the
// move (k_energy_measure) -- two launches less on the chain of a step.
__device__ __forceinline__ void pairing_presave_block(const SimDims &d, const DevPtrs &p, uint32_t real) {
    for (uint32_t k = threadIdx.x; k < d.n_sys; k += blockDim.x) {
        p.Esum_pre[static_cast<size_t>(real) * d.n_sys + k] = p.Esum[static_cast<size_t>(real) * d.n_sys + k];
        p.Msum_pre[static_cast<size_t>(real) * d.n_sys + k] = p.Msum[static_cast<size_t>(real) * d.n_sys + k];
    }
    const uint32_t t = threadIdx.x;
    if (t >= d.n_temps) return;
    uint64_t *st = p.pair_rng + (static_cast<size_t>(real) * d.n_temps * d.n_pairs + static_cast<size_t>(t) * d.n_pairs) * 4;
    DevRng rng;
    rng.s0 = st[0]; rng.s1 = st[1]; rng.s2 = st[2]; rng.s3 = st[3];
    uint32_t order[32]; // n_replicas <= 32 (checked on the host)
    for (uint32_t k = 0; k < d.n_replicas; ++k) order[k] = k;
    for (uint32_t i = d.n_replicas - 1; i >= 1; --i) { // reverse Fisher-Yates (SliceRandom::shuffle)
        const uint32_t j = dev_gen_range_u32(rng, i + 1);
        const uint32_t tmp = order[i];
        order[i] = order[j];
        order[j] = tmp;
    }
    uint32_t *slots = p.pair_slots + (static_cast<size_t>(real) * d.n_temps + t) * d.n_replicas;
    for (uint32_t k = 0; k < d.n_replicas; ++k) slots[k] = order[k] * d.n_temps + t;
    st[0] = rng.s0; st[1] = rng.s1; st[2] = rng.s2; st[3] = rng.s3;
}
// (c) with_bitmaps (the union-find forms): the per-slot site bitmaps of the swept state (k_slot_bitmaps: slot_bits[real][slot][MW],
// 64 sites x 32 slots per wave by 32 ballots), four chunks of 256 sites per block
__device__ __forceinline__ void slot_bitmaps_block(const SimDims &d, const DevPtrs &p, uint32_t bid) {
    const uint32_t n_chunks = (d.N + 255) / 256, n_quads = (n_chunks + 3) / 4;
    const uint32_t chunk = (bid % n_quads) * 4 + (threadIdx.x >> 8), w = (bid / n_quads) % d.W, real = bid / (n_quads * d.W);
    const uint32_t t256 = threadIdx.x & 255u, lane = t256 & 63u, i0 = chunk * 256 + (t256 & ~63u), i = i0 + lane;
    if (chunk >= n_chunks || i0 >= d.N) return;
    const uint32_t v = i < d.N ? p.S[(static_cast<size_t>(real) * d.W + w) * d.N + i] : 0u;
    uint32_t keep_lo = 0, keep_hi = 0; // lane k (< 32) ends up with the two words of slot k
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k) {
        const unsigned long long votes = __ballot((v >> k) & 1u);
        if (lane == k) {
            keep_lo = static_cast<uint32_t>(votes);
            keep_hi = static_cast<uint32_t>(votes >> 32);
        }
    }
    if (lane < 32) {
        uint32_t *row = p.slot_bits + (static_cast<size_t>(real) * d.SP + w * 32 + lane) * d.MW;
        const uint32_t word = i0 >> 5;
        row[word] = keep_lo;
        if (word + 1 < d.MW) row[word + 1] = keep_hi;
    }
}
__global__ __launch_bounds__(1024) void k_overlap_pairing(SimDims d, DevPtrs p, uint32_t n_overlap) {
    extern __shared__ uint32_t lds[];
    if (blockIdx.x >= n_overlap + d.n_real) {
        slot_bitmaps_block(d, p, blockIdx.x - n_overlap - d.n_real);
    } else if (blockIdx.x >= n_overlap) {
        pairing_presave_block(d, p, blockIdx.x - n_overlap);
    } else if (d.N <= 4096) {
        overlap_block_1024(d, p, blockIdx.x, lds);
    } else {
        if (threadIdx.x >= 256) return; // (a barrier waits for the surviving waves only)
        overlap_block(d, p, blockIdx.x, lds);
    }
}
// k_energy_measure: energies and magnetisations of the state the move left (mod.rs:749-756) by the sweep kernel's
// epilogue, and the statistics of the state BEFORE the move (mod.rs:543-595: sums from Esum_pre / Msum_pre, overlaps
// from the launch above) in the same launch.
template <int NN, bool DENSE>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(5))) void
k_energy_measure(SimDims d, DevPtrs p, uint32_t n_energy, PtParams pt, int do_accum, int accum_overlap) {
    extern __shared__ uint32_t lds[];
    if (blockIdx.x < n_energy) {
        sweep_hc_block<NN, false, false, DENSE>(d, p, blockIdx.x, 1, 0, nullptr, lds);
    } else {
        DevPtrs q = p;
        q.Esum = p.Esum_pre;
        q.Msum = p.Msum_pre;
        measure_block<false>(d, q, pt, do_accum, accum_overlap, 0, nullptr, blockIdx.x - n_energy, lds);
    }
}

// ---------------------------------------------------------------------------
// k_jorg: Joerg stochastic overlap-cluster move, single-cluster ("wolff") variant --
// clusters/overlap.rs:348-553 with dfs_cluster (utils.rs:125-149).  Bonds are
// activated with p = 1 - exp(-4 J s_i s_j / T) by draws of the pair's generator
// *in the order the depth-first search meets them*, so the search itself has to
// be replayed exactly: one lane per pair runs the reference's stack-based DFS
// (forward neighbour then backward neighbour per direction, LIFO stack), 48
// pairs per wave side by side.  The tile sits in LDS; the stacks are global
// scratch.  p depends only on the temperature for unit couplings and is
// computed by the host's expf (jorg_p), so host and device agree bit for bit.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t jorg_lds_words(const SimDims &d) {
    const size_t n_lanes = static_cast<size_t>(d.n_temps) * d.n_pairs;
    return static_cast<size_t>(d.N) * d.W + n_lanes * d.MW + 2 * n_lanes + d.n_sys + 4;
}

__global__ __launch_bounds__(1024) void k_jorg(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    const uint32_t n_lanes = d.n_temps * d.n_pairs;
    uint32_t *tile = lds;                                   // [W][N] spins
    uint32_t *member = lds + tile_words;                    // [n_lanes][MW] cluster bitmaps
    uint32_t *pair_slots = member + static_cast<size_t>(n_lanes) * d.MW; // [n_lanes][2]
    const uint32_t real = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * tile_words;
    const uint32_t *jsite = p.jsite + static_cast<size_t>(real) * d.N;
    uint64_t *prng = p.pair_rng + static_cast<size_t>(real) * n_lanes * 4;
    uint16_t *stack = p.dfs_stack + (static_cast<size_t>(real) * n_lanes + tid) * d.N;

    copy_to_lds(tile, Sg, static_cast<uint32_t>(tile_words), tid, nthreads);
    for (size_t i = tid; i < static_cast<size_t>(n_lanes) * d.MW; i += nthreads) member[i] = 0;
    __syncthreads();

    DevRng rng;
    rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    if (tid < n_lanes) {
        const uint32_t t = tid / d.n_pairs, g = tid % d.n_pairs;
        const uint64_t *st = prng + static_cast<size_t>(tid) * 4;
        rng.s0 = st[0];
        rng.s1 = st[1];
        rng.s2 = st[2];
        rng.s3 = st[3];
        if (g == 0) { // shuffle the replicas at temperature t (overlap.rs:44-49)
            uint32_t order[32];
            for (uint32_t k = 0; k < d.n_replicas; ++k) order[k] = k;
            for (uint32_t i = d.n_replicas - 1; i >= 1; --i) {
                const uint32_t j = dev_gen_range_u32(rng, i + 1);
                const uint32_t tmp = order[i];
                order[i] = order[j];
                order[j] = tmp;
            }
            for (uint32_t gg = 0; gg < d.n_pairs; ++gg) {
                pair_slots[2 * (t * d.n_pairs + gg) + 0] = order[2 * gg] * d.n_temps + t;
                pair_slots[2 * (t * d.n_pairs + gg) + 1] = order[2 * gg + 1] * d.n_temps + t;
            }
        }
    }
    __syncthreads();

    if (tid < n_lanes) {
        const uint32_t t = tid / d.n_pairs;
        const uint32_t sa = pair_slots[2 * tid], sb = pair_slots[2 * tid + 1];
        const uint32_t *wa = tile + (sa >> 5) * d.N, *wb = tile + (sb >> 5) * d.N;
        const uint32_t ba = sa & 31u, bb = sb & 31u;
        const float bond_p = p.jorg_p[t];
        uint32_t *mine = member + static_cast<size_t>(tid) * d.MW;
        auto spin_a = [&](uint32_t i) { return (wa[i] >> ba) & 1u; };
        auto active = [&](uint32_t i) { return (((wa[i] >> ba) ^ (wb[i] >> bb)) & 1u) != 0; };
        int64_t seed = -1;
        for (int probe = 0; probe < 64; ++probe) { // utils.rs:107-119
            const uint32_t i = static_cast<uint32_t>(dev_gen_range_u64(rng, d.N));
            if (active(i)) {
                seed = i;
                break;
            }
        }
        if (seed >= 0) {
            uint32_t top = 0;
            mine[seed >> 5] |= 1u << (seed & 31);
            stack[top++] = static_cast<uint16_t>(seed);
            while (top > 0) {
                const uint32_t site = stack[--top];
                const uint32_t jm = jsite[site];
                const NbrRow nbrs(p.nbr_site, site);
                for (uint32_t dir = 0; dir < 2 * d.nn; ++dir) { // f0, b0, f1, b1, ...
                    const uint32_t nb = nbrs.at(dir);
                    if ((mine[nb >> 5] >> (nb & 31u)) & 1u) continue;
                    if (!active(nb)) continue;
                    // inter = s_a(site) * s_a(nb) * J(bond) must be > 0
                    if (!((jm >> (12 + dir)) & 1u)) continue;
                    if ((spin_a(site) ^ spin_a(nb) ^ ((jm >> dir) & 1u)) != 0) continue;
                    const uint32_t draw = static_cast<uint32_t>(rng.next() >> 40);
                    bool bond;
                    if (d.unit) {
                        bond = __fmul_rn(static_cast<float>(draw), 1.0f / 16777216.0f) < bond_p;
                    } else { // p = 1 - exp(-4 |J| / T) of this bond (overlap.rs:430), owner = forward end
                        const uint32_t owner = (dir & 1u) ? nb : site;
                        const float j_abs = fabsf(p.Jf[(static_cast<size_t>(real) * d.N + owner) * d.nn + (dir >> 1)]);
                        bond = draw < jorg_threshold(j_abs, p.temps[t]);
                    }
                    if (bond) {
                        mine[nb >> 5] |= 1u << (nb & 31u);
                        stack[top++] = static_cast<uint16_t>(nb);
                    }
                }
            }
        }
        uint64_t *st = prng + static_cast<size_t>(tid) * 4;
        st[0] = rng.s0;
        st[1] = rng.s1;
        st[2] = rng.s2;
        st[3] = rng.s3;
    }
    __syncthreads();
    // flip both replicas of every pair on its cluster (pairs own disjoint bits of the tile)
    for (uint32_t item = tid; item < n_lanes * d.MW; item += nthreads) {
        const uint32_t lane = item / d.MW, mw = item % d.MW;
        uint32_t bits = member[item];
        const uint32_t sa = pair_slots[2 * lane], sb = pair_slots[2 * lane + 1];
        while (bits) {
            const uint32_t site = mw * 32 + (__ffs(bits) - 1);
            bits &= bits - 1;
            atomicXor(&tile[(sa >> 5) * d.N + site], 1u << (sa & 31u));
            atomicXor(&tile[(sb >> 5) * d.N + site], 1u << (sb & 31u));
        }
    }
    __syncthreads();
    for (size_t i = tid; i < tile_words; i += nthreads) Sg[i] = tile[i];
}

// ---------------------------------------------------------------------------
// Large lattices (a word's N sites do not fit the LDS, e.g. 64^3): the same
// algorithm with the spins left in global memory (L2-resident: 1 MB per word at
// 64^3).  k_sweep_big keeps one workgroup per (realisation, word) so that a
// workgroup barrier still orders the dependency levels; spin loads and stores
// are agent-scope relaxed atomics (served by L2, never a stale L1 line).
// Energy / magnetisation and the overlap dots are chunked over many workgroups
// and combined with integer atomics (exact, order-independent).
// Site records are 8 dwords here: site, f0, b0, f1, b1, f2, b2, jmeta.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sweep_big(SimDims d, DevPtrs p, int gibbs) {
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x;
    uint32_t *Sg = p.S + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint2 *planes = p.planes + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint32_t *planes_hi = p.planes_hi + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint4 *rec = p.rec + static_cast<size_t>(real) * (d.N + 1) * 2;
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    for (uint32_t pass = 0; pass < d.n_passes; ++pass) {
        const uint32_t pos = p.pass_start[pass] + tid;
        if (pos < p.pass_start[pass + 1]) {
            const BigRec r(rec[2 * pos], rec[2 * pos + 1]);
            uint32_t nb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) nb[k] = coherent_load(Sg + r.nb[k]);
            const uint32_t s = coherent_load(Sg + r.site);
            const uint2 lv = planes[r.site];
            const uint32_t acc = (gibbs ? accept_mask_gibbs(s, nb, r.jm, d.nn, lv.x, lv.y, planes_hi[r.site])
                                        : accept_mask(s, nb, r.jm, lv.x, lv.y)) & valid_bits;
            coherent_store(Sg + r.site, s ^ acc);
        }
        drain_global_stores(); // the next level reads these words through the L2
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_sweep_pipe: the large-lattice sweep spread over pipe_K workgroups per (realisation, word).
// The sites are cut into pipe_K slabs of consecutive indices (x-slabs of a cubic lattice).  A
// site needs the NEW value of its smaller neighbours and the OLD value of its larger ones; old
// values are read from S (not written during the sweep), new ones from S_next.  Slab k may run
// level l once slab k-1 has finished levels < l: by induction slab k-m has then finished levels
// <= l-m, which covers every smaller neighbour (the host checks level(j) + m <= level(i) for
// every such edge before enabling this kernel).  So the slabs form a software pipeline along the
// dependency levels: about n_levels + pipe_K steps instead of n_levels steps of one workgroup,
// with pipe_K times the workgroups.  Progress counters are epoch-based (never reset); a slab
// only ever waits for a workgroup with a smaller blockIdx (dispatched no later than itself), and
// the wait is bounded.  blockIdx is laid out so that all slabs of a word share an XCD (and its
// L2): block = k * n_tw_pad + tw with n_tw_pad a multiple of 8.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sweep_pipe(SimDims d, DevPtrs p, int gibbs, uint32_t epoch,
                                                    uint32_t n_tw_pad) {
    const uint32_t tw = blockIdx.x % n_tw_pad, k = blockIdx.x / n_tw_pad;
    if (tw >= d.n_real * d.W) return;
    const uint32_t real = tw / d.W, w = tw % d.W;
    const uint32_t tid = threadIdx.x;
    const uint32_t *Sold = p.S + static_cast<size_t>(tw) * d.N;
    uint32_t *Snew = p.S_next + static_cast<size_t>(tw) * d.N;
    const uint2 *planes = p.planes + static_cast<size_t>(tw) * d.N;
    const uint32_t *planes_hi = p.planes_hi + static_cast<size_t>(tw) * d.N;
    const uint4 *rec = p.rec + static_cast<size_t>(real) * (d.N + 1) * 2;
    uint32_t *progress = p.progress + static_cast<size_t>(tw) * d.pipe_K;
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint32_t base = epoch * (d.n_levels + 1);
    __shared__ uint32_t gave_up;
    if (tid == 0) gave_up = 0;
    __syncthreads();
    for (uint32_t level = 0; level < d.n_levels; ++level) {
        if (k > 0) {
            if (tid == 0) {
                uint32_t spins = 0;
                // wait until slab k-1 has completed `level` levels of this epoch (signed distance:
                // values of earlier epochs are smaller than base)
                // (relaxed: the data itself moves through agent-coherent accesses, and an acquire at
                // agent scope would invalidate the XCD's whole L2 at every level)
                while (static_cast<int32_t>(__hip_atomic_load(&progress[k - 1], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT) -
                                            (base + level)) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 26)) { // never reached unless a predecessor died: leave, and say so
                        gave_up = 1;            // (pa_sim_synchronize reports PA_ERR_DEVICE)
                        atomicExch(p.slab_error, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            if (gave_up) return;
        }
        const uint32_t begin = p.seg[level * (d.pipe_K + 1) + k], end = p.seg[level * (d.pipe_K + 1) + k + 1];
        for (uint32_t pos = begin + tid; pos < end; pos += 256) {
            const BigRec r(rec[2 * pos], rec[2 * pos + 1]);
            uint32_t nb[6];
#pragma unroll
            for (int q = 0; q < 6; ++q)
                nb[q] = r.nb[q] < r.site ? coherent_load(Snew + r.nb[q]) : Sold[r.nb[q]];
            const uint32_t s = Sold[r.site];
            const uint2 lv = planes[r.site];
            const uint32_t acc = (gibbs ? accept_mask_gibbs(s, nb, r.jm, d.nn, lv.x, lv.y, planes_hi[r.site])
                                        : accept_mask(s, nb, r.jm, lv.x, lv.y)) & valid_bits;
            coherent_store(Snew + r.site, s ^ acc);
        }
        drain_global_stores(); // every wave's stores are in the L2 before the counter moves
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(&progress[k], base + level + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------
// k_sweep_slab: the pipelined sweep with every slab's spins resident in the LDS.
//
// k_sweep_pipe pays three dependent L2 round trips per level (progress flag, neighbour words,
// store + drain).  Here a slab of whole hyperplanes (own = N / pipe_K sites) is loaded once and
// swept in place level by level like the on-chip kernel: neighbours inside the slab come from the
// LDS, and a level costs one workgroup barrier.  What crosses workgroups is kept off that path:
//   * inputs that do not depend on the spins -- the 16-byte site records (LDS word of the site and of
//     its six neighbours, coupling metadata), the acceptance planes k_rng filed in schedule order
//     and the export targets -- are streamed by a fifth wave with LDS-DMA loads into a ring
//     kSlabDepth levels deep; the four sweep waves never wait for them;
//   * a site whose new value another slab reads (the top plane; slab 0's first plane, through the
//     periodic wrap) also writes an EXCHANGE RECORD {value, tag, value ^ tag, ~tag}, tag = epoch
//     base + level + 1, at the position the reading slab will look for it.  A record is valid iff
//     its tag belongs to this sweep and the two check words agree, so value and "ready" travel in
//     one 16-byte access: no progress counter, no store drain, one round trip;
//   * the reading slab's sixth wave fetches the records of level l + 3 speculatively with LDS-DMA while
//     level l runs (into a staging ring; the site record of a reader holds the LDS word its staged
//     neighbour will occupy).  The same wave validates a group one level before its use; if the
//     producer was not that far ahead yet, the level's group is re-read until it is (bounded).
//     A slab therefore settles a few levels behind its predecessor and then runs at on-chip speed.
// The upper halo plane (and slab 0's lower one) carries OLD values and is loaded with the slab.
// Host-checked: every outside neighbour lies in one of the two halo planes, is either always newer
// or always older than the slab's sites that read it, a new value is read at exactly one level, and
// level(j) + m <= level(i) across m slabs.  Barriers are raw s_barrier + lgkmcnt(0): a
// __syncthreads() would drain the DMA rings (its fence waits vmcnt(0) while an LDS-DMA is pending).
// Counted vmcnt waits rely on vector-memory operations of a wave retiring in issue order.
// ---------------------------------------------------------------------------
// (kSlabDepth, kSlabStage: engine.hpp -- the host writes LDS word indices of staged records into the site records)
constexpr uint32_t kSlabThreads = 640; // sweep waves 0-3 (even levels), the loader wave (inputs), the fetcher wave (exchange records), sweep waves 6-9 (odd levels)

__device__ __forceinline__ void slab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS read the compiler does not see (an ordinary read would make it wait for every pending DMA)
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *ptr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v)
                 : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ptr)))
                 : "memory");
    return v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // a 128-bit register operand for inline asm
__device__ __forceinline__ uint4 lds_peek4(const uint32_t *ptr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v)
                 : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ptr)))
                 : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bool exchange_valid(const uint4 r, uint32_t base, uint32_t n_levels) {
    return (r.y - base - 1u) < n_levels && r.z == (r.x ^ r.y) && r.w == ~r.y;
}

template <bool GIBBS>
__global__ __launch_bounds__(kSlabThreads) void k_sweep_slab(SimDims d, DevPtrs p, uint32_t epoch, uint32_t n_tw_pad, int apply_swaps) {
    typedef __attribute__((address_space(1))) const void GlobalPtr;
    typedef __attribute__((address_space(3))) void LdsPtr;
    constexpr uint32_t D = kSlabDepth, R = kSlabStage;
    constexpr uint32_t G = GIBBS ? 8 : 7; // loader DMA instructions per level: 4 records, 2 planes, 1 export targets (+ 1 third plane)
    extern __shared__ uint32_t lds[];
    // (block-uniform values made scalar so that table reads indexed by them are scalar loads)
    const uint32_t tw = __builtin_amdgcn_readfirstlane(blockIdx.x % n_tw_pad);
    const uint32_t k = __builtin_amdgcn_readfirstlane(blockIdx.x / n_tw_pad);
    if (tw >= d.n_real * d.W) return;
    const uint32_t real = __builtin_amdgcn_readfirstlane(tw / d.W), w = __builtin_amdgcn_readfirstlane(tw % d.W);
    const uint32_t tid = threadIdx.x;
    const uint32_t K = d.pipe_K, P = d.plane, own = __builtin_amdgcn_readfirstlane(d.N / K);
    const uint32_t a = k * own;                                       // first site of the slab
    const uint32_t lo0 = __builtin_amdgcn_readfirstlane((a + d.N - P) % d.N);
    const uint32_t hi0 = __builtin_amdgcn_readfirstlane((a + own) % d.N); // halo planes below / above
    const uint32_t *Sold = p.S + static_cast<size_t>(tw) * d.N;
    uint32_t *Snew = p.S_next + static_cast<size_t>(tw) * d.N;
    const uint2 *planes = p.planes + static_cast<size_t>(tw) * d.planes_stride;
    const uint32_t *planes_hi = GIBBS ? p.planes_hi + static_cast<size_t>(tw) * d.planes_stride : nullptr;
    const uint4 *recS = p.recS + static_cast<size_t>(real) * d.planes_stride;
    uint4 *xchg = p.xchg + static_cast<size_t>(tw) * K * 2 * P;       // this word's exchange records, by reading slab
    const uint4 *xin = xchg + static_cast<size_t>(k) * 2 * P;         // the records this slab reads
    const uint32_t *halo_start_g = p.halo_start + static_cast<size_t>(k) * (d.n_levels + 2);
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint32_t base = epoch * (d.n_levels + 1);
    // one LDS array: spins (own, lower halo, upper halo), the rings, the staged records, control words
    uint32_t *ring_rec = lds + own + 2 * P;          // [D][256][4]
    uint32_t *ring_pl = ring_rec + D * 256 * 4;      // [D][256][2]
    uint32_t *ring_x = ring_pl + D * 256 * 2;        // [D][256] export targets
    uint32_t *stage = ring_x + D * 256;              // [R][128][4] exchange records of the levels ahead (= slab_stage_word())
    uint32_t *ctl = stage + R * 128 * 4;             // 1 gave up, 2 first level, 3 last level, 8 + s: group in stage slot s incomplete
    uint32_t *segs = ctl + 16;                       // [n_levels][2] (first padded position, sites) of this slab
    uint32_t *halo_start = segs + 2 * d.n_levels;    // [n_levels + 2] first staged entry of every level (copy of the table)
    uint32_t *ring_hi = halo_start + d.n_levels + 2; // [D][256] third acceptance plane (heat bath)
    // apply_swaps: a tempering event has run since the last sweep and its bit-column exchanges are still deferred
    // (swap_masks, written for every realisation by every event): they are applied to every word as it is loaded --
    // the input buffer is not written in this launch -- instead of by two launches in front of this one
    uint32_t mp[2] = {0, 0}, mc[2] = {0, 0}, mn[2] = {0, 0};
    if (apply_swaps) {
        const uint32_t *masks = p.swap_masks + static_cast<size_t>(real) * 2 * d.W;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mp[r] = w > 0 ? masks[r * d.W + w - 1] : 0u;
            mc[r] = masks[r * d.W + w];
            mn[r] = w + 1 < d.W ? masks[r * d.W + w + 1] : 0u;
        }
        if (k == 0 && tid == 0) p.applied_epoch[tw] = p.swap_epoch[real]; // (nothing in this launch reads it)
    }
    const bool any_swap = (mc[0] | mc[1] | (mp[0] >> 31) | (mp[1] >> 31)) != 0;
    const bool window = (((mp[0] | mp[1] | mc[0] | mc[1]) >> 31) != 0); // a pair straddles words
    auto load_word = [&](uint32_t site) -> uint32_t {
        const uint32_t v = Sold[site];
        if (!any_swap) return v;
        uint32_t prev = 0, next = 0;
        if (window) {
            if (w > 0) prev = Sold[static_cast<ptrdiff_t>(site) - static_cast<ptrdiff_t>(d.N)];
            if (w + 1 < d.W) next = Sold[static_cast<size_t>(site) + d.N];
        }
        return apply_swaps_window(prev, v, next, mp, mc, mn);
    };
    for (uint32_t i = tid; i < own; i += kSlabThreads) lds[i] = load_word(a + i);
    for (uint32_t i = tid; i < P; i += kSlabThreads) { // (only the planes read as OLD values are ever used)
        lds[own + i] = load_word(lo0 + i);
        lds[own + P + i] = load_word(hi0 + i);
    }
    for (uint32_t l = tid; l < d.n_levels; l += kSlabThreads) { // the per-level table reads become LDS reads
        const uint2 sg = p.slab_seg[l * K + k];
        segs[2 * l] = sg.x;
        segs[2 * l + 1] = sg.y;
    }
    for (uint32_t l = tid; l < d.n_levels + 2; l += kSlabThreads) halo_start[l] = halo_start_g[l];
    if (tid < 16) ctl[tid] = 0;
    if (tid == 0) ctl[2] = d.n_levels;
    __syncthreads();
    { // the slab's levels: first to last with a non-empty segment
        uint32_t first = d.n_levels, last = 0;
        for (uint32_t l = tid; l < d.n_levels; l += kSlabThreads)
            if (segs[2 * l + 1] != 0) {
                first = min(first, l);
                last = max(last, l);
            }
        if (first <= last) {
            atomicMin(&ctl[2], first);
            atomicMax(&ctl[3], last);
        }
    }
    __syncthreads();
    // (made scalar: loop bounds and everything derived from them stay in SGPRs)
    const uint32_t first = __builtin_amdgcn_readfirstlane(ctl[2]), last = __builtin_amdgcn_readfirstlane(ctl[3]);
    if (first > last) return; // a slab without sites
    const bool reads_others = __builtin_amdgcn_readfirstlane(halo_start[d.n_levels + 1]) != 0; // slab 0 stages nothing
    __syncthreads(); // the last ordinary barrier: from here on DMA loads stay in flight across barriers

    if (tid >= 256 && tid < 320) {
        // ================= loader wave: inputs of level L -> ring slot (L - first) % D
        const uint32_t lane = tid - 256;
        auto issue = [&](uint32_t L) {
            const uint32_t slot = (L - first) % D;
            const uint32_t begin = L <= last ? lds_peek(&segs[2 * L]) : 0u; // past the end: harmless, never read
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((GlobalPtr *)(recS + begin + q * 64 + lane),
                                                 (LdsPtr *)(ring_rec + (slot * 256 + q * 64) * 4), 16, 0, 0);
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds((GlobalPtr *)(planes + begin + q * 128 + 2 * lane),
                                                 (LdsPtr *)(ring_pl + (slot * 256 + q * 128) * 2), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((GlobalPtr *)(p.xportS + begin + 4 * lane), (LdsPtr *)(ring_x + slot * 256), 16, 0, 0);
            if (GIBBS)
                __builtin_amdgcn_global_load_lds((GlobalPtr *)(planes_hi + begin + 4 * lane),
                                                 (LdsPtr *)(ring_hi + slot * 256), 16, 0, 0);
        };
        for (uint32_t L = first; L < first + D - 1; ++L) issue(L);
        slab_barrier(); // matches the sweep waves' barrier after the fetcher's first groups
        __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 2) * G) & 0xF) | ((((D - 2) * G) >> 4) << 14)); // the first level's pieces
        slab_barrier(); // the sweep waves pick the first level up
        for (uint32_t level = first; level <= last; ++level) {
            // the pieces of `level + 1` have landed once only the D - 3 younger groups are outstanding: the sweep
            // waves pick up the records of the next level at the end of this one
            __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 3) * G) & 0xF) | ((((D - 3) * G) >> 4) << 14));
            slab_barrier(); // X(level)
            if (reads_others && lds_peek(&ctl[8 + level % R])) { // the sweep waves re-read the level's group: same barriers
                slab_barrier();
                if (lds_peek(&ctl[1])) return;
            }
            issue(level + D - 1); // into the slot read during level - 1 (those reads retired before X(level))
        }
        __builtin_amdgcn_s_waitcnt(0x0F70); // nothing of this wave may still be writing the LDS at exit
        return;
    }

    if (tid >= 320 && tid < 384) {
        // ================= fetcher wave: exchange records first read at level L -> staging slot L % R (128 entries)
        if (!reads_others) { // nothing to stage: only keep the barriers company
            slab_barrier();
            slab_barrier();
            for (uint32_t level = first; level <= last; ++level) slab_barrier();
            return;
        }
        const uint32_t lane = tid - 320;
        // halo_start[L], clamped (levels past the last read nothing)
        auto hs = [&](uint32_t L) -> uint32_t { return lds_peek(&halo_start[min(L, last + 1)]); };
        auto stage_fetch = [&](uint32_t L, uint32_t e0) {
            // (aux 16 = sc1: agent scope, past the CU's L1 -- the records are written while this kernel runs)
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds((GlobalPtr *)(xin + e0 + q * 64 + lane),
                                                 (LdsPtr *)(stage + ((L % R) * 128 + q * 64) * 4), 16, 0, 16);
        };
        // is the landed group of level L (n_h records) complete?  -> ctl[8 + L % R] (read by the other waves after X(L)) and
        // the return value; this wave reads what its own DMA wrote, ordered by its own counted wait.  Both halves are
        // requested before either is looked at: the wave's chain per level is one LDS round trip, not six -- every
        // wave of the workgroup meets at the level barrier, so this wave's chain was the level time of every slab that
        // reads another (480 against 400 ns per level, tools/stamps_slab.py)
        auto validate = [&](uint32_t L, uint32_t n_h) -> uint32_t {
            u32x4 r0, r1;
            const uint32_t a0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage + ((L % R) * 128 + lane) * 4));
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1)
                         : "v"(a0)
                         : "memory");
            const bool bad = (lane < n_h && !exchange_valid(make_uint4(r0.x, r0.y, r0.z, r0.w), base, d.n_levels)) ||
                             (64 + lane < n_h && !exchange_valid(make_uint4(r1.x, r1.y, r1.z, r1.w), base, d.n_levels));
            const uint32_t any = __ballot(bad) != 0ull ? 1u : 0u;
            if (lane == 0)
                asm volatile("ds_write_b32 %0, %1" ::"v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&ctl[8 + L % R]))), "v"(any)
                             : "memory");
            return any;
        };
        // Optional head start for the slab(s) read from (tuning knob): wait until the group of a level `lead` ahead
        // exists, then begin.
        if (d.slab_lead) {
            const uint32_t lead = d.slab_lead;
            uint32_t L = min(first + lead, last);
            while (L > first && lds_peek(&halo_start[L + 1]) == lds_peek(&halo_start[L])) --L; // a level that reads something
            const uint32_t e0 = lds_peek(&halo_start[L]), n_h = lds_peek(&halo_start[L + 1]) - e0;
            for (uint32_t spins = 0; spins < (1u << 20); ++spins) { // bounded: afterwards the ordinary re-read path takes over
                bool missing = false;
                for (uint32_t q = 0; q < 2; ++q)
                    if (q * 64 + lane < n_h) {
                        u32x4 raw;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(raw) : "v"(xin + e0 + q * 64 + lane) : "memory");
                        missing = missing || !exchange_valid(make_uint4(raw.x, raw.y, raw.z, raw.w), base, d.n_levels);
                    }
                if (__ballot(missing) == 0ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        for (uint32_t L = first; L < first + kSlabAhead; ++L) stage_fetch(L, L <= last ? hs(L) : 0u);
        __builtin_amdgcn_s_waitcnt(0x0F70); // the first groups are simply waited for
        uint32_t flag_cur = validate(first, hs(first + 1) - hs(first));
        // window of the table kept in registers: h[k] = halo_start[level + 1 + k]
        uint32_t h[kSlabAhead + 1];
#pragma unroll
        for (uint32_t k = 0; k <= kSlabAhead; ++k) h[k] = hs(first + 1 + k);
        slab_barrier();
        slab_barrier(); // the sweep waves pick the first level up
        for (uint32_t level = first; level <= last; ++level) {
            // the window's next entry: requested now, looked at after this level's work
            uint32_t h_new;
            asm volatile("ds_read_b32 %0, %1"
                         : "=v"(h_new)
                         : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&halo_start[min(level + 2 + kSlabAhead, last + 1)])))
                         : "memory");
            // the group of level + 1 has landed once only the kSlabAhead - 2 younger groups (two pieces each) are outstanding:
            // a fetch has that many levels to come back through this CU's memory queue, behind the loader's pieces
            __builtin_amdgcn_s_waitcnt(0x0F70 | (2u * (kSlabAhead - 2)));
            const uint32_t flag_next = validate(level + 1, h[1] - h[0]);
            slab_barrier(); // X(level)
            if (flag_cur) { // the sweep waves re-read the level's group: same barrier
                slab_barrier();
                if (lds_peek(&ctl[1])) return;
            }
            stage_fetch(level + kSlabAhead, level + kSlabAhead <= last ? h[kSlabAhead - 1] : 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h_new)::"memory");
#pragma unroll
            for (uint32_t k = 0; k < kSlabAhead; ++k) h[k] = h[k + 1];
            h[kSlabAhead] = h_new;
            flag_cur = flag_next;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        return;
    }

    // ================= sweep waves: two sets of four take the levels in turn (set A: waves 0-3, set B: waves 6-9).
    // A level is a chain -- barrier, neighbour words, accept, write -- and a lone wave issues an instruction every
    // ~4.7 cycles whatever it is, so everything else is done by the set that is NOT sweeping the level: it picks up the
    // next level's inputs (landed: the loader's counted wait; flag set by the fetcher before the barrier), unpacks the
    // seven LDS addresses and sends the previous level's global stores (new value, exchange record) on their way.
    const bool set_b = __builtin_amdgcn_readfirstlane(tid) >= 384u; // (wave-uniform, and known to be: the branches below are scalar)
    const uint32_t t = set_b ? tid - 384u : tid; // position inside a level
    slab_barrier(); // the fetcher's first groups are in the staging ring, the first one validated
    struct Inputs {
        uint32_t at, at4, nb[6], jm, hi, target, count, redo;
        uint2 lv;
    };
    auto pick_up = [&](uint32_t L) -> Inputs {
        Inputs in;
        in.count = L <= last ? segs[2 * L + 1] : 0u;
        in.redo = (reads_others && L <= last) ? ctl[8 + L % R] : 0u;
        const uint32_t slot = (L - first) % D;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        in.lv = make_uint2(0u, 0u);
        in.target = in.hi = 0;
        // (read whatever the count says -- the slot exists -- so that the count and the ring travel in one LDS round trip;
        // compute and settle look at `count`)
        r = *reinterpret_cast<const uint4 *>(ring_rec + (slot * 256 + t) * 4);
        in.lv = *reinterpret_cast<const uint2 *>(ring_pl + (slot * 256 + t) * 2);
        in.target = ring_x[slot * 256 + t];
        if (GIBBS) in.hi = ring_hi[slot * 256 + t];
        // (a neighbour in another slab is the value word of its staged exchange record: the host wrote that LDS word)
        const uint32_t lds0 = lds_addr_of(lds); // (LDS byte addresses, ready for ds_read)
        in.at = r.x & 0xFFFFu;
        in.nb[0] = lds0 + (r.x >> 16) * 4;
        in.nb[1] = lds0 + (r.y & 0xFFFFu) * 4;
        in.nb[2] = lds0 + (r.y >> 16) * 4;
        in.nb[3] = lds0 + (r.z & 0xFFFFu) * 4;
        in.nb[4] = lds0 + (r.z >> 16) * 4;
        in.nb[5] = lds0 + (r.w & 0xFFFFu) * 4;
        in.at4 = lds0 + in.at * 4;
        in.jm = r.w >> 16;
        // computed HERE, by the set that is not sweeping (left alone, the compiler sinks the unpacking to its first
        // use, behind the next barrier and onto the chain)
        asm volatile("" : "+v"(in.at), "+v"(in.at4), "+v"(in.jm), "+v"(in.lv.x), "+v"(in.lv.y), "+v"(in.hi), "+v"(in.target), "+v"(in.count));
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("" : "+v"(in.nb[q]));
        return in;
    };
    slab_barrier(); // the loader's pieces of the first level and the first flag are in place
    const bool dense3 = !GIBBS && d.dense_nn == 3;
    Inputs cur{};
    if (!set_b) cur = pick_up(first);
    // The loader's and the fetcher's code lies on the (lane-masked) way here, so the compiler believes LDS-DMA writes may
    // be pending and would put a vmcnt(0) wait before the first ring read INSIDE the loop -- where it waits for this
    // wave's own global stores, a memory round trip per level.  Said once here, it costs nothing: nothing is outstanding.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // the global stores of the level this set swept last, sent during the other set's level
    uint32_t owed_value = 0, owed_at = 0, owed_target = 0, owed_tag = 0;
    bool owed = false;
    auto settle = [&]() {
        if (owed) {
            // (written out: left to the compiler, the store would be waited for -- vmcnt(0) -- before the next LDS
            // read that reuses one of its registers; the next launch reads Snew)
            asm volatile("global_store_dword %0, %1, off" ::"v"(Snew + a + owed_at), "v"(owed_value) : "memory");
            if (owed_target) { // another slab reads this site: value and "ready" in one 16-byte record
                const u32x4 record = {owed_value, owed_tag, owed_value ^ owed_tag, ~owed_tag};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(xchg + (owed_target - 1)), "v"(record) : "memory");
            }
            owed = false;
        }
    };
    uint32_t redo_other = reads_others ? ctl[8 + first % R] : 0u; // the flag of the level the OTHER set sweeps next
    if (p.slab_stamps && tid == 0) {
        p.slab_stamps[static_cast<size_t>(blockIdx.x) * (d.n_levels + 4)] = first;
        p.slab_stamps[static_cast<size_t>(blockIdx.x) * (d.n_levels + 4) + 1] = last;
    }
    for (uint32_t level = first; level <= last; ++level) {
        slab_barrier(); // X(level): last level's spins are visible; the pieces of level + 1 have landed
        const bool mine = (((level - first) & 1u) != 0u) == set_b;
        const uint32_t sslot = level % R;
        // (registers only: the set that does not sweep this level asked for its flag while it swept the last one -- an
        // LDS round trip at this point would stand in front of its pick-up of the next level's inputs)
        const uint32_t redo = mine ? cur.redo : redo_other;
        if (p.slab_stamps && tid == 0) // (diagnostic runs only; bit 63: the level re-read its exchange records)
            p.slab_stamps[static_cast<size_t>(blockIdx.x) * (d.n_levels + 4) + 2 + level] =
                wall_clock64() | (static_cast<unsigned long long>(redo != 0) << 63);
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(redo) != 0, 0)) {
            // ---- the producer was not far enough ahead when this level's records were fetched: read them again
            const uint32_t e0 = halo_start[level], n_h = halo_start[level + 1] - e0;
            if (tid < n_h) { // (threads of set A)
                uint32_t spins = 0;
                for (;;) {
                    u32x4 raw;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(raw) : "v"(xin + e0 + tid) : "memory");
                    const uint4 r = make_uint4(raw.x, raw.y, raw.z, raw.w);
                    if (exchange_valid(r, base, d.n_levels)) {
                        *reinterpret_cast<uint4 *>(stage + (sslot * 128 + tid) * 4) = r;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { // never reached unless the producer died: leave
                        ctl[1] = 1;
                        break;
                    }
                }
            }
            slab_barrier();
            if (ctl[1]) {
                if (tid == 0) atomicExch(p.slab_error, 1u);
                return;
            }
        }
        if (mine) {
            // ---- this level's sites: reads -> accept -> write, nothing else before the next barrier
            if (reads_others) redo_other = ctl[8 + (level + 1) % R]; // (set by the fetcher before X(level); used after X(level + 1))
            if (t < cur.count) {
                uint32_t nb[6] = {lds_load(cur.nb[0]), lds_load(cur.nb[1]), lds_load(cur.nb[2]),
                                  lds_load(cur.nb[3]), lds_load(cur.nb[4]), lds_load(cur.nb[5])};
                const uint32_t sv = lds_load(cur.at4);
                // (every bond non-zero in three directions -- ferromagnets, +-J: no non-zero masks, no case split)
                const uint32_t acc = (GIBBS ? accept_mask_gibbs(sv, nb, cur.jm, d.nn, cur.lv.x, cur.lv.y, cur.hi)
                                      : dense3 ? accept_mask_dense<3>(sv, nb, cur.jm, cur.lv.x, cur.lv.y)
                                               : accept_mask(sv, nb, cur.jm, cur.lv.x, cur.lv.y)) & valid_bits;
                const uint32_t value = sv ^ acc;
                lds_store(cur.at4, value);
                owed = true;
                owed_value = value;
                owed_at = cur.at;
                owed_target = cur.target;
                owed_tag = base + level + 1;
            }
        } else {
            settle();
            cur = pick_up(level + 1);
        }
    }
    settle();
}

// chunk = 4096 schedule positions per workgroup; counts go to cnt[real][n_sys][3] (u32 atomics)
__device__ __forceinline__ void energy_big_block(const SimDims &d, const DevPtrs &p, uint32_t bid, uint32_t n_chunks) {
    __shared__ uint32_t scratch[5 * 48];
    const uint32_t chunk = bid % n_chunks;
    const uint32_t w = (bid / n_chunks) % d.W, real = bid / (n_chunks * d.W);
    const uint32_t tid = threadIdx.x;
    const uint32_t *Sg = p.S + (static_cast<size_t>(real) * d.W + w) * d.N;
    const uint32_t *jsite = p.jsite + static_cast<size_t>(real) * d.N;
    uint32_t c[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    uint32_t qa[16], qb[16], qc[16];
    // sites in index order (the sums do not care): the site's word, its coupling bits and its forward
    // neighbours' indices are coalesced streams, and on a hypercubic lattice so are the neighbour words
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t site = chunk * 4096 + k * 256 + tid;
        qa[k] = qb[k] = qc[k] = 0;
        if (site < d.N) {
            const uint4 fw = p.fwd_big[site];
            const uint32_t jm = jsite[site]; // neg[0..11] | nz[12..23] for f0,b0,f1,b1,f2,b2,..
            const uint32_t s = Sg[site];
            const uint32_t f0 = Sg[fw.x], f1 = Sg[fw.y], f2 = Sg[fw.z];
            const uint32_t u0 = (s ^ f0 ^ (0u - ((jm >> 0) & 1u))) & (0u - ((jm >> 12) & 1u));
            const uint32_t u1 = (s ^ f1 ^ (0u - ((jm >> 2) & 1u))) & (0u - ((jm >> 14) & 1u));
            const uint32_t u2 = (s ^ f2 ^ (0u - ((jm >> 4) & 1u))) & (0u - ((jm >> 16) & 1u));
            qa[k] = u0 ^ u1 ^ u2;
            qb[k] = maj3(u0, u1, u2);
            qc[k] = s;
        }
    }
    count16(qa, c[0]);
    count16(qb, c[1]);
    count16(qc, c[2]);
    uint32_t tot[3];
    block_bit_totals<3>(c, scratch, tot);
    if (tid < 32 && w * 32 + tid < d.n_sys) {
        uint32_t *cnt = p.big_counts + (static_cast<size_t>(real) * d.n_sys + w * 32 + tid) * 4;
        atomicAdd(&cnt[0], tot[0] + 2 * tot[1]); // unsatisfied forward bonds
        atomicAdd(&cnt[1], tot[2]);              // spins equal to -1
    }
}

__global__ __launch_bounds__(256) void k_energy_big(SimDims d, DevPtrs p, uint32_t n_chunks) {
    energy_big_block(d, p, blockIdx.x, n_chunks);
}

__device__ __forceinline__ void overlap_big_block(const SimDims &d, const DevPtrs &p, uint32_t bid, uint32_t n_chunks) {
    __shared__ uint32_t scratch[5 * 48];
    const uint32_t chunk = bid % n_chunks;
    const uint32_t w = (bid / n_chunks) % d.W, real = bid / (n_chunks * d.W);
    const uint32_t tid = threadIdx.x;
    const uint32_t q = d.n_temps >> 5, r = d.n_temps & 31u;
    const uint32_t wa = w + q, wb = w + q + 1;
    const uint32_t *S0 = p.S + static_cast<size_t>(real) * d.W * d.N;
    const uint32_t *A = S0 + static_cast<size_t>(w) * d.N;
    const uint32_t *B0 = wa < d.W ? S0 + static_cast<size_t>(wa) * d.N : nullptr;
    const uint32_t *B1 = wb < d.W ? S0 + static_cast<size_t>(wb) * d.N : nullptr;
    auto shifted = [&](uint32_t site) {
        const uint32_t b0 = B0 ? B0[site] : 0u, b1 = B1 ? B1[site] : 0u;
        return r ? ((b0 >> r) | (b1 << (32 - r))) : b0;
    };
    uint32_t c[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[s][j] = 0;
    uint32_t qa[16], qb[16], qc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t site = chunk * 4096 + k * 256 + tid; // index order, as in k_energy_big
        qa[k] = qb[k] = qc[k] = 0;
        if (site < d.N) {
            const uint4 fw = p.fwd_big[site];
            const uint32_t fwd[3] = {fw.x, fw.y, fw.z};
            const uint32_t y = A[site] ^ shifted(site);
            uint32_t z[3] = {0, 0, 0};
            for (uint32_t dir = 0; dir < d.nn; ++dir) z[dir] = y ^ A[fwd[dir]] ^ shifted(fwd[dir]);
            qa[k] = y;
            qb[k] = z[0] ^ z[1] ^ z[2];
            qc[k] = maj3(z[0], z[1], z[2]);
        }
    }
    count16(qa, c[0]);
    count16(qb, c[1]);
    count16(qc, c[2]);
    uint32_t tot[3];
    block_bit_totals<3>(c, scratch, tot);
    if (tid < 32 && w * 32 + tid < d.n_sys) {
        uint32_t *cnt = p.big_counts + (static_cast<size_t>(real) * d.n_sys + w * 32 + tid) * 4;
        atomicAdd(&cnt[2], tot[0]);              // sites where the pair disagrees
        atomicAdd(&cnt[3], tot[1] + 2 * tot[2]); // bonds with q_i q_k = -1
    }
}

__global__ __launch_bounds__(256) void k_overlap_big(SimDims d, DevPtrs p, uint32_t n_chunks) {
    overlap_big_block(d, p, blockIdx.x, n_chunks);
}

// Both in one launch (they read the same state and write different counters): the energies of the state a sweep left
// and its overlap measurement, when nothing comes between them in the protocol.
// The workgroup that finishes last for its realisation also turns the counts into Esum / Msum and the dots (what
// k_big_finalize does as a launch of its own elsewhere): every workgroup waits for its counter atomics to be
// acknowledged (they are performed at the device's coherence point), then draws a ticket.
__global__ __launch_bounds__(256) void k_energy_overlap_big(SimDims d, DevPtrs p, uint32_t n_chunks, uint32_t n_energy) {
    __shared__ uint32_t am_last;
    const uint32_t bid = blockIdx.x < n_energy ? blockIdx.x : blockIdx.x - n_energy;
    const uint32_t real = bid / (n_chunks * d.W);
    if (blockIdx.x < n_energy) energy_big_block(d, p, blockIdx.x, n_chunks);
    else overlap_big_block(d, p, blockIdx.x - n_energy, n_chunks);
    drain_global_stores(); // (the atomics without return count as stores)
    __syncthreads();
    if (threadIdx.x == 0) am_last = atomicAdd(&p.big_ticket[real], 1u) == 2u * n_chunks * d.W - 1u ? 1u : 0u;
    __syncthreads();
    if (!am_last) return;
    for (uint32_t k = threadIdx.x; k < d.n_sys; k += blockDim.x) {
        const size_t idx = static_cast<size_t>(real) * d.n_sys + k;
        uint32_t *cnt = p.big_counts + idx * 4;
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = atomicExch(&cnt[j], 0u); // (read where the additions were performed, and cleared)
        p.Esum[idx] = p.nz_bonds[real] - 2 * static_cast<int32_t>(v[0]);
        p.Msum[idx] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(v[1]);
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if ((rep & 1u) == 0 && rep + 1 < d.n_replicas) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(v[2]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) - 2 * static_cast<int32_t>(v[3]);
        }
    }
    if (threadIdx.x == 0) p.big_ticket[real] = 0;
}

// counts -> Esum / Msum (mode 0), dot_spin / dot_link (mode 1) or both (mode 2); one thread per slot
__global__ void k_big_finalize(SimDims d, DevPtrs p, int mode) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.n_real * d.n_sys) return;
    const uint32_t real = idx / d.n_sys, k = idx % d.n_sys;
    uint32_t *cnt = p.big_counts + static_cast<size_t>(idx) * 4;
    if (mode == 0 || mode == 2) {
        p.Esum[idx] = p.nz_bonds[real] - 2 * static_cast<int32_t>(cnt[0]);
        p.Msum[idx] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(cnt[1]);
        cnt[0] = 0;
        cnt[1] = 0;
    }
    if (mode != 0) {
        const uint32_t rep = k / d.n_temps, t = k % d.n_temps;
        if ((rep & 1u) == 0 && rep + 1 < d.n_replicas) {
            const size_t o = (static_cast<size_t>(real) * d.n_pairs + (rep >> 1)) * d.n_temps + t;
            p.dot_spin[o] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(cnt[2]);
            p.dot_link[o] = static_cast<int32_t>(d.N * d.nn) - 2 * static_cast<int32_t>(cnt[3]);
        }
        cnt[2] = 0;
        cnt[3] = 0;
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
static constexpr size_t kLdsPerCu = 160 * 1024;
static constexpr size_t kRngLds = 64 * 16 * 8 * 4;

size_t sweep_lds_bytes(const SimDims &d, bool gibbs) {
    const size_t n1 = static_cast<size_t>(d.N) + 1;
    return ((n1 + 1) / 2 * 2 + (gibbs ? 3 : 2) * n1 + 5 * 48) * 4;
}
size_t measure_lds_bytes(const SimDims &d) { return measure_lds_words(d) * 4; }
size_t sweep_hc_lds_bytes(const SimDims &d, bool gibbs) { return sweep_hc_lds_words(d, gibbs) * 4; }
static size_t overlap_lds_bytes(const SimDims &d) {
    return ((static_cast<size_t>(d.N) + 1) + 64 * 24 + 96) * 4; // (block_bit_totals_quad<3>: [17][48]; overlap_block_1024: [64][24] + [96])
}

hipError_t configure_kernels(const SimDims &) {
    for (const void *fn : {reinterpret_cast<const void *>(k_rng<false, false>),
                           reinterpret_cast<const void *>(k_rng<true, false>),
                           reinterpret_cast<const void *>(k_rng<false, true>),
                           reinterpret_cast<const void *>(k_sweep<false, 0, true>),
                           reinterpret_cast<const void *>(k_sweep<false, 0>),
                           reinterpret_cast<const void *>(k_sweep<false, 1>),
                           reinterpret_cast<const void *>(k_sweep<false, 2>),
                           reinterpret_cast<const void *>(k_sweep<false, 3>),
                           reinterpret_cast<const void *>(k_sweep<true, 0>),
                           reinterpret_cast<const void *>(k_sweep<true, 3>),
                           reinterpret_cast<const void *>(k_sweep_hc<1>),
                           reinterpret_cast<const void *>(k_sweep_hc<2>),
                           reinterpret_cast<const void *>(k_sweep_hc<3>),
                           reinterpret_cast<const void *>(k_sweep_hc<3, true>),
                           reinterpret_cast<const void *>(k_overlap),
                           reinterpret_cast<const void *>(k_overlap_1024),
                           reinterpret_cast<const void *>(k_overlap_wide),
                           reinterpret_cast<const void *>(k_sweep_wide),
                           reinterpret_cast<const void *>(k_rng_relay<false, false>),
                           reinterpret_cast<const void *>(k_rng_relay<false, true>),
                           reinterpret_cast<const void *>(k_rng_relay<true, false>),
                           reinterpret_cast<const void *>(k_rng_relay<true, true>),
                           reinterpret_cast<const void *>(k_rng_relay<true, 2>),
                           reinterpret_cast<const void *>(k_rng_relay<false, 2>),
                           reinterpret_cast<const void *>(k_rng_relay<false, 2, true>),
                           reinterpret_cast<const void *>(k_rng_overlap<false>),
                           reinterpret_cast<const void *>(k_rng_overlap<false, true>),
                           reinterpret_cast<const void *>(k_rng_relay<false, false, true>),
                           reinterpret_cast<const void *>(k_rng_relay<false, true, true>),
                           reinterpret_cast<const void *>(k_sweep_hc<1, false, true>),
                           reinterpret_cast<const void *>(k_sweep_hc<2, false, true>),
                           reinterpret_cast<const void *>(k_sweep_hc<3, false, true>),
                           reinterpret_cast<const void *>(k_sweep_hc<1, false, false, false>),
                           reinterpret_cast<const void *>(k_sweep_hc<1, false, true, false>),
                           reinterpret_cast<const void *>(k_sweep_hc<2, false, false, false>),
                           reinterpret_cast<const void *>(k_sweep_hc<2, false, true, false>),
                           reinterpret_cast<const void *>(k_sweep_hc<3, false, false, false>),
                           reinterpret_cast<const void *>(k_sweep_hc<3, false, true, false>),
                           reinterpret_cast<const void *>(k_rng_overlap<true>),
                           reinterpret_cast<const void *>(k_houdayer),
                           reinterpret_cast<const void *>(k_houdayer_overlap),
                           reinterpret_cast<const void *>(k_houdayer_grow),
                           reinterpret_cast<const void *>(k_houdayer_grow_measure),
                           reinterpret_cast<const void *>(k_jorg),
                           reinterpret_cast<const void *>(k_stream_sweep_flagged<3, true>),
                           reinterpret_cast<const void *>(k_overlap_pairing),
                           reinterpret_cast<const void *>(k_energy_measure<1, true>),
                           reinterpret_cast<const void *>(k_energy_measure<2, true>),
                           reinterpret_cast<const void *>(k_energy_measure<3, true>),
                           reinterpret_cast<const void *>(k_energy_measure<1, false>),
                           reinterpret_cast<const void *>(k_energy_measure<2, false>),
                           reinterpret_cast<const void *>(k_energy_measure<3, false>),
                           reinterpret_cast<const void *>(k_measure<false>),
                           reinterpret_cast<const void *>(k_measure<true>)}) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(kLdsPerCu));
        if (e != hipSuccess) return e;
    }
    for (const void *fn : {reinterpret_cast<const void *>(k_sweep_slab<false>), reinterpret_cast<const void *>(k_sweep_slab<true>)}) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsPerCu));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t sweep_wide_lds_bytes(const SimDims &d) { return sweep_wide_lds_words(d) * 4; }
bool takes_wide_route(const SimDims &d, bool gibbs) { return d.wide && !gibbs; }

hipError_t launch_rng(const SimDims &d, const DevPtrs &p, bool gibbs, hipStream_t stream) {
    if (takes_wide_route(d, gibbs)) { // the level planes of the Metropolis table, by the heat bath's seven-threshold code
        DevPtrs q = p;
        q.thr_gibbs = p.thr_wide;
        const uint32_t groups = ((d.n_real * d.W + 1) / 2 + 3) / 4;
        PA_LAUNCH((k_rng<false, true>), dim3(groups * d.B), dim3(256), kRngLds, stream, d, q);
        return hipGetLastError();
    }
    if (d.general) return launch_rng_f(d, p, gibbs, stream);
    if (reg_sweep_applies(d, gibbs) && !d.planes_perm) { // the sweep that follows reads the planes in its register order
        SimDims dr = d;
        dr.planes_perm = 1;
        return launch_rng(dr, p, gibbs, stream);
    }
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2;
    const uint32_t groups = (n_pairs_tw + 3) / 4;
    if (gibbs)
        PA_LAUNCH((k_rng<false, true>), dim3(groups * d.B), dim3(256), kRngLds, stream, d, p);
    else if (d.thr_preshifted)
        PA_LAUNCH((k_rng<true, false>), dim3(groups * d.B), dim3(256), kRngLds, stream, d, p);
    else
        PA_LAUNCH((k_rng<false, false>), dim3(groups * d.B), dim3(256), kRngLds, stream, d, p);
    return hipGetLastError();
}

// Random stream of the next Metropolis sweep fused with the overlap measurement of the current
// state (LDS kernels, unit couplings only).
hipError_t launch_rng_overlap(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool gibbs) {
    if (reg_sweep_applies(d, gibbs) && !d.planes_perm) {
        SimDims dr = d;
        dr.planes_perm = 1;
        return launch_rng_overlap(dr, p, stream, gibbs);
    }
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2;
    const uint32_t groups = (n_pairs_tw + 3) / 4;
    const uint32_t n_overlap = d.n_real * d.W;
    const size_t lds = std::max(kRngLds, overlap_lds_bytes(d));
    if (gibbs)
        PA_LAUNCH((k_rng_overlap<false, true>), dim3(n_overlap + groups * d.B), dim3(256), lds, stream, d, p, n_overlap);
    else if (d.thr_preshifted)
        PA_LAUNCH(k_rng_overlap<true>, dim3(n_overlap + groups * d.B), dim3(256), lds, stream, d, p, n_overlap);
    else
        PA_LAUNCH(k_rng_overlap<false>, dim3(n_overlap + groups * d.B), dim3(256), lds, stream, d, p, n_overlap);
    return hipGetLastError();
}

// A relayed measurement block needs: a workgroup of 256 threads (measure_block: n_sys <= 2 x 256, n_temps,
// n_replicas, 2 W <= 256), one replay lane per (realisation, ladder) of a random-stream workgroup's eight tile words,
// the on-chip table kernels.
// realisations a random-stream workgroup's 2 * rng_waves tile words can belong to
static uint32_t relay_reals_per_block(const SimDims &d) { return (2u * d.rng_waves - 1u) / d.W + 2u; }

bool rng_relay_supported(const SimDims &d) {
    // (large lattices: on the LDS-resident slab path, whose sweep launch exchanges the deferred bit columns as it loads)
    return !d.general && (!d.big || (d.pipe_K && d.pipe_lds)) && d.n_sys <= 512 && d.n_temps <= 256 && d.n_replicas <= 256 && 2 * d.W <= 256 &&
           relay_reals_per_block(d) * d.n_replicas <= 256 && measure_lds_bytes(d) <= 64 * 1024;
}

// full_ladder events are replayed too when a ladder's edges fit one mask word and the draws fit behind the jump table
bool rng_relay_ladder_supported(const SimDims &d) {
    const size_t tasks = static_cast<size_t>(relay_reals_per_block(d)) * d.n_replicas;
    return rng_relay_supported(d) && d.n_temps <= 32 && tasks * (d.n_temps - 1) <= 4 * 64 * d.rng_waves && // four items per lane
           tasks * (d.n_temps + 1) * 4 <= 16 * 1024;
}

hipError_t launch_rng_relay(const SimDims &d, const DevPtrs &p, bool with_overlap, const PtParams &pt, bool do_accum,
                            bool accum_overlap, bool do_pt, hipStream_t stream, bool gibbs) {
    if (reg_sweep_applies(d, gibbs) && !d.planes_perm) {
        SimDims dr = d;
        dr.planes_perm = 1;
        return launch_rng_relay(dr, p, with_overlap, pt, do_accum, accum_overlap, do_pt, stream, gibbs);
    }
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2;
    const uint32_t groups = (n_pairs_tw + d.rng_waves - 1) / d.rng_waves;
    const uint32_t n_overlap = with_overlap ? d.n_real * d.W : 0u;
    const uint32_t n_measure = (do_accum || accum_overlap || do_pt) ? d.n_real : 0u;
    const int replay = (n_measure && do_pt && d.n_temps >= 2) ? (pt.schedule != 0 ? 2 : 1) : 0; // 2: the whole ladder
    // behind the jump table: the replayed decisions (one word per ladder), or the ladders' draws and edge masks
    const size_t replay_words = replay == 2 ? static_cast<size_t>(relay_reals_per_block(d)) * d.n_replicas * (d.n_temps + 1) : 256;
    static const size_t lds_floor = [] { // (experiment knob: a larger reservation limits the workgroups per CU)
        const char *env = dev_env("PEAPODS_RNG_LDS");
        return env ? static_cast<size_t>(std::strtol(env, nullptr, 10)) : 0;
    }();
    const size_t lds = std::max({kRngLds + std::max<size_t>(1024, replay_words * 4), n_overlap ? overlap_lds_bytes(d) : 0,
                                 n_measure ? measure_lds_bytes(d) : 0, lds_floor});
    const dim3 grid(n_overlap + n_measure + groups * d.B), block(64 * d.rng_waves);
    const int a = do_accum ? 1 : 0, o = accum_overlap ? 1 : 0, t = do_pt ? 1 : 0;
#define PA_RELAY(P, R, G) PA_LAUNCH((k_rng_relay<P, R, G>), grid, block, lds, stream, d, p, n_overlap, n_measure, pt, a, o, t)
    if (gibbs) { // (heat-bath thresholds are compared unshifted, like k_rng<false, true>)
        if (replay == 2) PA_RELAY(false, 2, true);
        else if (replay == 1) PA_RELAY(false, 1, true);
        else PA_RELAY(false, 0, true);
    } else if (d.thr_preshifted) {
        if (replay == 2) PA_RELAY(true, 2, false);
        else if (replay == 1) PA_RELAY(true, 1, false);
        else PA_RELAY(true, 0, false);
    } else {
        if (replay == 2) PA_RELAY(false, 2, false);
        else if (replay == 1) PA_RELAY(false, 1, false);
        else PA_RELAY(false, 0, false);
    }
#undef PA_RELAY
    return hipGetLastError();
}

// (experiment, see k_stream_sweep_flagged) flags: one counter per stream workgroup of eight waves, never reset -- launch
// number `epoch` (from 1) waits for epoch * B
bool flagged_sweep_applies(const SimDims &d) {
    return d.hc && !d.general && !d.wide && d.thr_preshifted && d.dense_nn == d.nn && d.nn == 3 && sweep_hc_lds_bytes(d) >= kRngLds + 1024;
}
hipError_t launch_stream_sweep_flagged(const SimDims &d, const DevPtrs &p, uint32_t *flags, uint32_t epoch, uint32_t *timed_out,
                                       hipStream_t stream) {
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2, groups = (n_pairs_tw + 7) / 8;
    const uint32_t n_stream = groups * d.B;
    PA_LAUNCH((k_stream_sweep_flagged<3, true>), dim3(n_stream + d.n_real * d.W), dim3(512), sweep_hc_lds_bytes(d), stream, d, p,
              n_stream, flags, epoch * d.B, timed_out);
    return hipGetLastError();
}

// Files the sums a relayed tempering event exchanged (measure_block, pt.relay) -- launched behind that launch when the
// following sweep leaves the sums alone; realisations without an accepted swap have nothing staged.
__global__ __launch_bounds__(256) void k_adopt_relay_sums(SimDims d, DevPtrs p) {
    const uint32_t real = blockIdx.x;
    if (!p.relay_swapped[real]) return;
    for (uint32_t k = threadIdx.x; k < d.n_sys; k += blockDim.x) {
        p.Esum[static_cast<size_t>(real) * d.n_sys + k] = p.Esum_relay[static_cast<size_t>(real) * d.n_sys + k];
        p.Msum[static_cast<size_t>(real) * d.n_sys + k] = p.Msum_relay[static_cast<size_t>(real) * d.n_sys + k];
    }
}
hipError_t launch_adopt_relay_sums(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    PA_LAUNCH(k_adopt_relay_sums, dim3(d.n_real), dim3(256), 0, stream, d, p);
    return hipGetLastError();
}

// The overlap blocks and the measurement blocks of k_rng_relay without its random-stream workgroups (the tail of a
// run: the last sweep's overlap and its deferred measurement in one launch).  pt.relay must be 0: the measurement
// block then moves the generators itself.
hipError_t launch_measure_overlap(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, bool accum_overlap,
                                  bool do_pt, hipStream_t stream) {
    const uint32_t n_overlap = d.n_real * d.W, n_measure = d.n_real;
    const size_t lds = std::max(overlap_lds_bytes(d), measure_lds_bytes(d));
    const dim3 grid(n_overlap + n_measure), block(256);
    const int a = do_accum ? 1 : 0, o = accum_overlap ? 1 : 0, t = do_pt ? 1 : 0;
    if (d.thr_preshifted) PA_LAUNCH((k_rng_relay<true, false>), grid, block, lds, stream, d, p, n_overlap, n_measure, pt, a, o, t);
    else PA_LAUNCH((k_rng_relay<false, false>), grid, block, lds, stream, d, p, n_overlap, n_measure, pt, a, o, t);
    return hipGetLastError();
}

hipError_t launch_sweep(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                        bool gibbs, hipStream_t stream) {
    if (do_sweep && takes_wide_route(d, gibbs)) {
        PA_LAUNCH(k_sweep_wide, dim3(d.n_real * d.W), dim3(256), sweep_wide_lds_bytes(d), stream, d, p,
                           with_energy ? 1 : 0, 1);
        return hipGetLastError();
    }
    if (d.general) return launch_sweep_f(d, p, with_energy, do_sweep, stream);
    if (do_sweep && reg_sweep_applies(d, gibbs)) return launch_sweep_reg(d, p, with_energy, stream);
    if (d.big) {
        if (do_sweep)
            PA_LAUNCH(k_sweep_big, dim3(d.n_real * d.W), dim3(1024), 0, stream, d, p,
                               gibbs ? 1 : 0);
        if (with_energy) {
            const uint32_t n_chunks = (d.N + 4095) / 4096;
            PA_LAUNCH(k_energy_big, dim3(d.n_real * d.W * n_chunks), dim3(256), 0, stream, d,
                               p, n_chunks);
            PA_LAUNCH(k_big_finalize, dim3((d.n_real * d.n_sys + 255) / 256), dim3(256), 0,
                               stream, d, p, 0);
        }
        return hipGetLastError();
    }
    const dim3 grid(d.n_real * d.W), block(256);
    static const size_t pad = [] {
        const char *env = dev_env("PEAPODS_SWEEP_LDS");
        return env ? static_cast<size_t>(std::strtol(env, nullptr, 10)) : 0;
    }();
    const size_t lds = std::max(sweep_lds_bytes(d, gibbs && do_sweep), pad);
    unsigned long long *no_stamps = nullptr;
    const int e = with_energy ? 1 : 0, s = do_sweep ? 1 : 0;
    static const bool no_hc = dev_env("PEAPODS_NO_HC") != nullptr;
    if (gibbs && do_sweep) {
        if (d.hc && !no_hc && sweep_hc_lds_bytes(d, true) <= 160 * 1024) { // (83 KB at 16^3: one workgroup per CU)
            const dim3 block_hc(512);
            const size_t lds_g = sweep_hc_lds_bytes(d, true);
#define PA_HC(NN, G, D) PA_LAUNCH((k_sweep_hc<NN, false, G, D>), grid, block_hc, lds_g, stream, d, p, e, s, no_stamps)
            if (d.dense_nn == d.nn) {
                switch (d.nn) {
                case 1: PA_HC(1, true, true); break;
                case 2: PA_HC(2, true, true); break;
                default: PA_HC(3, true, true); break;
                }
            } else { // some bonds are zero: per-site non-zero masks and rule bits from the table
                switch (d.nn) {
                case 1: PA_HC(1, true, false); break;
                case 2: PA_HC(2, true, false); break;
                default: PA_HC(3, true, false); break;
                }
            }
#undef PA_HC
            return hipGetLastError();
        }
        PA_LAUNCH((k_sweep<false, 0, true>), grid, block, lds, stream, d, p, e, s, no_stamps);
        return hipGetLastError();
    }
    if (d.hc && !no_hc) { // implicit neighbours, table in the LDS
        // One sweep workgroup per CU while the launch has no more of them than the chip has CUs (a stream group of the
        // bench ensemble: 192): two on one CU share its LDS pipeline and SIMDs and stretch each other's passes (96
        // workgroups 17.5 us, 192 workgroups 23.5 us per launch; reserving 82 KB: 58.0 -> 57.4 us per step).
        const size_t alone = (do_sweep && grid.x <= 256) ? 82 * 1024 : 0;
        const size_t lds_hc = std::max({sweep_hc_lds_bytes(d), pad, alone});
        const dim3 block_hc(512); // two sets of four waves take the passes in turn
#define PA_HC(NN, D) PA_LAUNCH((k_sweep_hc<NN, false, false, D>), grid, block_hc, lds_hc, stream, d, p, e, s, no_stamps)
        if (d.dense_nn == d.nn) {
            switch (d.nn) {
            case 1: PA_HC(1, true); break;
            case 2: PA_HC(2, true); break;
            default: PA_HC(3, true); break;
            }
        } else {
            switch (d.nn) {
            case 1: PA_HC(1, false); break;
            case 2: PA_HC(2, false); break;
            default: PA_HC(3, false); break;
            }
        }
#undef PA_HC
        return hipGetLastError();
    }
    switch (d.dense_nn) {
    case 1: PA_LAUNCH((k_sweep<false, 1>), grid, block, lds, stream, d, p, e, s, no_stamps); break;
    case 2: PA_LAUNCH((k_sweep<false, 2>), grid, block, lds, stream, d, p, e, s, no_stamps); break;
    case 3: PA_LAUNCH((k_sweep<false, 3>), grid, block, lds, stream, d, p, e, s, no_stamps); break;
    default: PA_LAUNCH((k_sweep<false, 0>), grid, block, lds, stream, d, p, e, s, no_stamps); break;
    }
    return hipGetLastError();
}

// spins of the slab and its two halo planes, then the record / plane rings (sized for the heat bath) and control words
// spins of the slab and its two halo planes, the input rings (sized for the heat bath), the staged exchange records,
// control words and the slab's segment table
size_t sweep_slab_lds_bytes_for(uint32_t own_sites, uint32_t plane_sites, uint32_t n_levels) {
    return (static_cast<size_t>(own_sites) + 2 * static_cast<size_t>(plane_sites) + kSlabDepth * 256 * 8 + kSlabStage * 128 * 4 + 16 +
            3 * n_levels + 2) * 4;
}
size_t sweep_slab_lds_bytes(const SimDims &d) {
    return d.pipe_K ? sweep_slab_lds_bytes_for(d.N / d.pipe_K, d.plane, d.n_levels) : 0;
}

hipError_t launch_sweep_pipe(const SimDims &d, const DevPtrs &p, bool gibbs, uint32_t epoch,
                             hipStream_t stream, bool apply_swaps) {
    const uint32_t n_tw_pad = (d.n_real * d.W + 7u) & ~7u;
    // (apply_swaps: only the LDS-resident kernel exchanges the deferred bit columns itself; enqueue_sweep knows)
    if (d.pipe_lds && gibbs)
        PA_LAUNCH(k_sweep_slab<true>, dim3(n_tw_pad * d.pipe_K), dim3(kSlabThreads), sweep_slab_lds_bytes(d),
                           stream, d, p, epoch, n_tw_pad, apply_swaps ? 1 : 0);
    else if (d.pipe_lds)
        PA_LAUNCH(k_sweep_slab<false>, dim3(n_tw_pad * d.pipe_K), dim3(kSlabThreads), sweep_slab_lds_bytes(d),
                           stream, d, p, epoch, n_tw_pad, apply_swaps ? 1 : 0);
    else
        PA_LAUNCH(k_sweep_pipe, dim3(n_tw_pad * d.pipe_K), dim3(256), 0, stream, d, p,
                           gibbs ? 1 : 0, epoch, n_tw_pad);
    return hipGetLastError();
}

// Diagnostic build of k_sweep: workgroup 0 records shader-clock stamps at phase boundaries.
hipError_t launch_sweep_stamped(const SimDims &d, const DevPtrs &p, unsigned long long *stamps,
                                hipStream_t stream) {
    const dim3 grid(d.n_real * d.W), block(256);
    if (d.hc && d.dense_nn == 3) { // every workgroup stamps: stamps[grid][8]
        PA_LAUNCH((k_sweep_hc<3, true>), grid, dim3(512), sweep_hc_lds_bytes(d), stream, d, p, 1, 1, stamps);
        return hipGetLastError();
    }
    if (d.dense_nn == 3)
        PA_LAUNCH((k_sweep<true, 3>), grid, block, sweep_lds_bytes(d, false), stream, d, p,
                           1 | (dev_env("PEAPODS_DBG") ? std::atoi(dev_env("PEAPODS_DBG")) << 8 : 0), 1, stamps);
    else
        PA_LAUNCH((k_sweep<true, 0>), grid, block, sweep_lds_bytes(d, false), stream, d, p, 1, 1, stamps);
    return hipGetLastError();
}

hipError_t launch_apply_swaps(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    PA_LAUNCH(k_apply_swaps, dim3((d.N + 255) / 256, d.n_real), dim3(256), 0, stream, d, p);
    PA_LAUNCH(k_mark_applied, dim3((d.n_real * d.W + 255) / 256), dim3(256), 0, stream, d, p);
    return hipGetLastError();
}

hipError_t launch_energy_overlap_big(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    const uint32_t n_chunks = (d.N + 4095) / 4096, n_each = d.n_real * d.W * n_chunks;
    PA_LAUNCH(k_energy_overlap_big, dim3(2 * n_each), dim3(256), 0, stream, d, p, n_chunks, n_each);
    return hipGetLastError();
}

hipError_t launch_overlap(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    if (d.big) {
        const uint32_t n_chunks = (d.N + 4095) / 4096;
        PA_LAUNCH(k_overlap_big, dim3(d.n_real * d.W * n_chunks), dim3(256), 0, stream, d, p,
                           n_chunks);
        PA_LAUNCH(k_big_finalize, dim3((d.n_real * d.n_sys + 255) / 256), dim3(256), 0,
                           stream, d, p, 1);
        return hipGetLastError();
    }
    if (d.nn > 3) {
        PA_LAUNCH(k_overlap_wide, dim3(d.n_real * d.W), dim3(256),
                           ((static_cast<size_t>(d.N) + 1) + 5 * 80) * 4, stream, d, p);
        return hipGetLastError();
    }
    if (d.N <= 4096 && !dev_env("PEAPODS_OVERLAP_256"))
        PA_LAUNCH(k_overlap_1024, dim3(d.n_real * d.W), dim3(1024), overlap_lds_bytes(d), stream, d, p);
    else
        PA_LAUNCH(k_overlap, dim3(d.n_real * d.W), dim3(256), overlap_lds_bytes(d), stream, d, p);
    return hipGetLastError();
}

// Houdayer single-cluster move in two launches; with_overlap: the overlap blocks of k_overlap ride in the first
// (the caller exchanges dot_* and dot_*_prev after it and may launch the statistics between the two).
hipError_t launch_houdayer_seeds(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_overlap) {
    if (with_overlap) {
        const size_t lds = std::max(houdayer_lds_words(d) * 4, overlap_lds_bytes(d));
        PA_LAUNCH(k_houdayer_overlap, dim3(d.n_real + d.n_real * d.W), dim3(1024), lds, stream, d, p, d.n_real);
    } else {
        PA_LAUNCH(k_houdayer, dim3(d.n_real), dim3(1024), houdayer_lds_words(d) * 4, stream, d, p);
    }
    return hipGetLastError();
}
hipError_t launch_houdayer_grow(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_energy) {
    const bool e = with_energy && d.hc && d.dense_nn == d.nn;
    const size_t lds = (2 * static_cast<size_t>(d.N) + 4 + (e ? d.N + 4 + 5 * 48 : 0)) * 4;
    PA_LAUNCH(k_houdayer_grow, dim3(d.n_real * d.W), dim3(1024), lds, stream, d, p, e ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_houdayer_grow_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, hipStream_t stream,
                                        bool with_energy) {
    const bool e = with_energy && d.hc && d.dense_nn == d.nn;
    const size_t lds = std::max((2 * static_cast<size_t>(d.N) + 4 + (e ? d.N + 4 + 5 * 48 : 0)) * 4, measure_lds_bytes(d));
    PA_LAUNCH(k_houdayer_grow_measure, dim3(d.n_real * d.W + d.n_real), dim3(1024), lds, stream, d, p, e ? 1 : 0,
                       d.n_real * d.W, pt, do_accum ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_houdayer(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_energy) {
    PA_LAUNCH(k_houdayer, dim3(d.n_real), dim3(1024), houdayer_lds_words(d) * 4, stream, d, p);
    const bool e = with_energy && d.hc && d.dense_nn == d.nn;
    const size_t lds = (2 * static_cast<size_t>(d.N) + 4 + (e ? d.N + 4 + 5 * 48 : 0)) * 4;
    PA_LAUNCH(k_houdayer_grow, dim3(d.n_real * d.W), dim3(1024), lds, stream, d, p, e ? 1 : 0);
    return hipGetLastError();
}
bool houdayer_computes_energy(const SimDims &d) { return d.hc && d.dense_nn == d.nn; }
size_t houdayer_lds_bytes(const SimDims &d) { return houdayer_lds_words(d) * 4; }

bool move_riders_supported(const SimDims &d) { return d.hc && !d.big && !d.general && d.n_pairs > 0 && d.nn <= 3; }
hipError_t launch_overlap_pairing(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_bitmaps) {
    const uint32_t n_overlap = d.n_real * d.W;
    const uint32_t n_bitmaps = with_bitmaps ? d.n_real * d.W * (((d.N + 255) / 256 + 3) / 4) : 0u;
    PA_LAUNCH(k_overlap_pairing, dim3(n_overlap + d.n_real + n_bitmaps), dim3(1024), overlap_lds_bytes(d), stream, d, p, n_overlap);
    return hipGetLastError();
}
hipError_t launch_energy_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, bool accum_overlap,
                                 hipStream_t stream) {
    const uint32_t n_energy = d.n_real * d.W;
    const dim3 grid(n_energy + d.n_real), block(512);
    const size_t lds = std::max(sweep_hc_lds_bytes(d), measure_lds_bytes(d));
    const int a = do_accum ? 1 : 0, o = accum_overlap ? 1 : 0;
#define PA_EM(NN, D) PA_LAUNCH((k_energy_measure<NN, D>), grid, block, lds, stream, d, p, n_energy, pt, a, o)
    if (d.dense_nn == d.nn) {
        switch (d.nn) {
        case 1: PA_EM(1, true); break;
        case 2: PA_EM(2, true); break;
        default: PA_EM(3, true); break;
        }
    } else {
        switch (d.nn) {
        case 1: PA_EM(1, false); break;
        case 2: PA_EM(2, false); break;
        default: PA_EM(3, false); break;
        }
    }
#undef PA_EM
    return hipGetLastError();
}

hipError_t launch_jorg(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    PA_LAUNCH(k_jorg, dim3(d.n_real), dim3(1024), jorg_lds_words(d) * 4, stream, d, p);
    return hipGetLastError();
}
size_t jorg_lds_bytes(const SimDims &d) { return jorg_lds_words(d) * 4; }

hipError_t launch_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_overlap,
                          bool do_accum, bool accum_overlap, bool do_pt, hipStream_t stream) {
    PA_LAUNCH(k_measure<false>, dim3(d.n_real), dim3(1024), measure_lds_bytes(d), stream, d,
                       p, pt, do_overlap ? 1 : 0, do_accum ? 1 : 0, accum_overlap ? 1 : 0,
                       do_pt ? 1 : 0, static_cast<unsigned long long *>(nullptr));
    return hipGetLastError();
}

hipError_t launch_measure_stamped(const SimDims &d, const DevPtrs &p, const PtParams &pt,
                                  unsigned long long *stamps, hipStream_t stream) {
    PA_LAUNCH(k_measure<true>, dim3(d.n_real), dim3(1024), measure_lds_bytes(d), stream, d,
                       p, pt, 1, 1, 1, 1, stamps);
    return hipGetLastError();
}

} // namespace pa
