// k_sweep_reg: the Metropolis sweep of a 16 x 16 x 16 lattice with every bond non-zero (the +-J glass of BASELINE
// configs[2]/[3]) held entirely in registers -- one wave per tile word, no LDS, no barrier.
//
// What it replaces (reference): mcmc/sweep.rs:8-19 (local_field), :51-61 (sweep_sites, typewriter order),
// :170-185 (attempt_flip_lookup), spins/energy.rs:78-110 (energies and magnetisations of the swept state).
// Same decisions as k_sweep_hc: the dependency levels x + y + z = c of the typewriter order, the acceptance-level
// planes of the random-stream kernel, U + L >= 3 by bit-sliced adders.
//
// Layout.  lane = 16 (z / 4) + y, a lane's four z-values m = z mod 4 sit side by side in registers, and the register
// index is j = (x + y + z) mod 16 = the site's LEVEL mod 16:
//     R[m][j] of lane (y, zq)  =  spins (32 slots) of site x = (j - y - z) mod 16,  z = 4 zq + m.
// At level c every lane works on register j = c mod 16 (a uniform, static index once the level loop is unrolled by 16);
// its x-neighbours are the same lane's registers j -+ 1 (the periodic wrap included: x = 15 -> 0 lands in register j + 1
// too), its y-neighbours the lanes y -+ 1 of its row of sixteen (DPP row rotation = the periodic wrap), its z-neighbours
// the registers (m -+ 1, j -+ 1) of the same lane, or of the lane sixteen further for m = 0 / 3 (ds_bpermute, requested
// a level ahead: never waited for).  The dependent chain of a level is its vector instructions and nothing else.
//
// STATUS (round 4): exact (tests/test_gpu_parity.py::test_register_resident_sweep_matches_oracle) but NOT the default --
// measured 48.8 us per launch against k_sweep_hc's 24.2 on the bench ensemble (in-kernel: spins 6, levels 16.7, store 3-4,
// energy 5.6 us per wave, tools/stamps_reg.py).  A wave issues one vector instruction every ~5 cycles however independent
// they are, so what this kernel saves in instructions (10 K against 14.5 K per tile word) it loses by running them on one
// wave instead of eight.  Opt-in: PEAPODS_DEVELOPER=1 PEAPODS_REG=1.
//
// Cost: 64 + 12 VGPRs of state (the acceptance planes wait in 32 KB of LDS); ~33 instructions per site and level slot, 172 live slots of 4 x 48.  A workgroup of
// k_sweep_hc issues 14.5 K wave-instructions per tile word over eight waves, this kernel ~8.5 K over one -- beside the
// VALU-bound random stream of the other stream group that difference is the point -- and a workgroup without LDS fits a
// CU that random-stream workgroups have filled.
#include "engine.hpp"

#include "bitcount.hpp"
#include "device_util.hpp"

namespace pa {

uint32_t reg_slot_of_site(uint32_t site) { // 16^3: site = 256 x + 16 y + z
    const uint32_t x = site >> 8, y = (site >> 4) & 15u, z = site & 15u;
    return (((x + y + z) & 15u) * 4u + (z & 3u)) * 64u + (z >> 2) * 16u + y;
}

namespace {

template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int kFromBelow = 0x121; // row_ror:1  -- lane i receives lane i - 1 of its row (y - 1, periodic)
constexpr int kFromAbove = 0x12F; // row_ror:15 -- lane i receives lane i + 1 (y + 1)

__device__ __forceinline__ uint32_t from_lane(uint32_t byte_addr, uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(byte_addr), static_cast<int>(v)));
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t sign_mask(uint32_t word, int bit) {
    return static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(word), bit, 1));
}

// Buffer addressing (SGPR descriptor + 32-bit lane offset + scalar offset): with flat 64-bit addresses the compiler
// keeps one address pair per row of the acceptance planes in registers -- 128 VGPRs of nothing.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xFFFFFFFF, 0x00020000);
}

struct LdsWords2 { uint32_t x, y; };
__device__ __forceinline__ uint2 lds_pair(uint32_t lds_byte_addr) { // 8-byte aligned: one ds_read_b64
    const __attribute__((address_space(3))) LdsWords2 *q =
        reinterpret_cast<const __attribute__((address_space(3))) LdsWords2 *>(static_cast<uintptr_t>(lds_byte_addr));
    return make_uint2(q->x, q->y);
}

// flip <=> U + L >= 3 with U = unsatisfied bonds of the site's six, L = acceptance level of its draw (accept_mask_dense<3>)
__device__ __forceinline__ uint32_t accept_six(const uint32_t (&u)[6], uint32_t l0, uint32_t l1) {
    const uint32_t s1 = xor3(u[0], u[1], u[2]), c1 = maj3(u[0], u[1], u[2]);
    const uint32_t s2 = xor3(u[3], u[4], u[5]), c2 = maj3(u[3], u[4], u[5]);
    const uint32_t c3 = s1 & s2, U0 = s1 ^ s2;
    const uint32_t U1 = xor3(c1, c2, c3), U2 = maj3(c1, c2, c3);
    const uint32_t t0 = U0 ^ l0, k0 = U0 & l0;
    const uint32_t t1 = xor3(U1, l1, k0), k1 = maj3(U1, l1, k0);
    return (t0 & t1) | U2 | k1;
}

// bit-sliced accumulate: acc (7 planes, <= 64) += p (5 planes, <= 16)
__device__ __forceinline__ void add_planes(uint32_t (&acc)[7], const uint32_t (&p)[5]) {
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t a = acc[k];
        acc[k] = xor3(a, p[k], carry);
        carry = maj3(a, p[k], carry);
    }
#pragma unroll
    for (int k = 5; k < 7; ++k) {
        const uint32_t a = acc[k];
        acc[k] = a ^ carry;
        carry = a & carry;
    }
}

} // namespace

// (three such waves per SIMD at most: 168 registers -- beside three random-stream waves of 112 a SIMD still has room for one)
// STAMPS: diagnostic build (pa_debug_sweep_reg_stamps): lane 0 files s_memrealtime (100 MHz) at the phase boundaries
#define PA_REG_STAMP(k)                                                                            \
    if (STAMPS && threadIdx.x == 0) stamps[blockIdx.x * 8 + (k)] = wall_clock64();
template <bool STAMPS>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(3, 3))) void k_sweep_reg(SimDims d, DevPtrs p, int with_energy,
                                                                                                          unsigned long long *stamps) {
    PA_REG_STAMP(0)
    extern __shared__ uint32_t lds_planes[]; // [16 levels mod 16][4][64] uint2: the word's acceptance planes (32 KB)
    uint32_t real, w;
    tile_of_block(d, blockIdx.x, real, w);
    const uint32_t lane = threadIdx.x;
    const uint32_t y = lane & 15u, zq = lane >> 4, t0 = y + 4u * zq;
    const size_t tile = static_cast<size_t>(real) * d.W + w;
    const uint32_t *Sin = p.S + tile * d.N;
    const __amdgpu_buffer_rsrc_t s_in = buffer_of(Sin), s_out = buffer_of(p.S_next + tile * d.N);

    // (requested first: their latency hides behind the spins)
    const uint32_t epoch = p.swap_epoch[real];
    const uint32_t applied = p.applied_epoch[tile];
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
    uint32_t cw[4][3]; // coupling sign bits of the lane's 64 sites: word (m, direction), bit 2 j + (0 forward | 1 backward)
    {
        const uint32_t *cpl = p.reg_cpl + static_cast<size_t>(real) * (12 * 64);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int dir = 0; dir < 3; ++dir) cw[m][dir] = cpl[(m * 3 + dir) * 64 + lane];
    }
    const int32_t nz = p.nz_bonds[real];

    // the word's spins: register (m, j) <- site x = (j - y - z) mod 16
    uint32_t R[4][16];
    const uint32_t col = y * 16u + 4u * zq; // + m: the lane's columns; + 256 x: the site
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            R[m][j] = __builtin_amdgcn_raw_buffer_load_b32(s_in, ((((j - m - t0) & 15u) << 8) + col) * 4, m * 4, 0);
    // ... and its acceptance planes, straight into the LDS (32 pieces of 1 KB, no registers): the level loop reads a row of
    // 512 bytes per (level mod 16, m) from there.  (Read from global memory three levels ahead, the compiler's wait
    // counts collapsed to vmcnt(0) at the loop's back edge and wherever it regrouped the requests: 54 us per launch.)
    {
        typedef __attribute__((address_space(1))) const void GlobalPtr;
        typedef __attribute__((address_space(3))) void LdsPtr;
        const char *src = reinterpret_cast<const char *>(p.planes + tile * d.planes_stride) + lane * 16;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            __builtin_amdgcn_global_load_lds((GlobalPtr *)(src + k * 1024), (LdsPtr *)(lds_planes + k * 256), 16, 0, 0);
    }
    if (epoch != applied) {
        // tempering swaps accepted after the previous sweep: exchange the bit columns now (sweep_hc_block does the same).
        // A pair that straddles two words needs the old neighbouring words (the input buffer is not written in this
        // launch); where there is no such word its mask is zero and any readable address serves.
        const bool window = (((mp[0] | mp[1] | mc[0] | mc[1]) >> 31) != 0);
        const __amdgpu_buffer_rsrc_t s_prev = buffer_of(w > 0 ? Sin - d.N : Sin), s_next = buffer_of(w + 1 < d.W ? Sin + d.N : Sin);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t prev[16], next[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) prev[j] = next[j] = 0;
            if (window) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t at = ((((j - m - t0) & 15u) << 8) + col) * 4;
                    prev[j] = __builtin_amdgcn_raw_buffer_load_b32(s_prev, at, m * 4, 0);
                    next[j] = __builtin_amdgcn_raw_buffer_load_b32(s_next, at, m * 4, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) R[m][j] = apply_swaps_window(prev[j], R[m][j], next[j], mp, mc, mn);
            __builtin_amdgcn_sched_barrier(0); // (one column set's neighbouring words in flight, not all four)
        }
        if (lane == 0) p.applied_epoch[tile] = epoch;
    }
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint32_t addr_dn = ((lane - 16u) & 63u) << 2, addr_up = ((lane + 16u) & 63u) << 2; // lanes zq - 1 / zq + 1

    // ---- the 46 levels ----
    PA_REG_STAMP(1)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the planes have landed (one wave: no barrier needed)
    PA_REG_STAMP(2)
    uint32_t pl_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint32_t *)lds_planes)) + lane * 8;
    uint32_t up_next = from_lane(addr_up, R[0][1]);  // level 0: z + 1 of the lane's m = 3 column
    uint32_t dn_next = from_lane(addr_dn, R[3][15]); //          z - 1 of its m = 0 column
#pragma unroll 1
    for (uint32_t cb = 0; cb < 48; cb += 16) {
        // (the 384 sign masks of the body do not depend on cb: left alone, the compiler computes them all before the loop
        // and spills them; an opaque touch of their source words per round keeps each one next to its use)
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(cw[m][0]), "+v"(cw[m][1]), "+v"(cw[m][2]));
        asm volatile("" : "+v"(pl_addr)); // (likewise the 64 plane rows, which do not depend on cb either)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = cb + j;
            const int jm = (j + 15) & 15, jp = (j + 1) & 15;
            const uint32_t up_cur = up_next, dn_cur = dn_next;
            up_next = from_lane(addr_up, R[0][(j + 2) & 15]); // for level c + 1 (that register rests until level c + 2)
            const uint32_t cm = c - t0;
            auto site_op = [&](const int m, const uint32_t zm, const uint32_t zp) {
                if (c - static_cast<uint32_t>(m) > 42u) return; // (uniform) no lane has a site of this column set at this level
                const uint32_t s = R[m][j], xm = R[m][jm], xp = R[m][jp];
                const uint32_t ym = dpp_mov<kFromBelow>(xm), yp = dpp_mov<kFromAbove>(xp);
                uint32_t u[6];
                u[0] = xor3(s, xp, sign_mask(cw[m][0], 2 * j));
                u[1] = xor3(s, xm, sign_mask(cw[m][0], 2 * j + 1));
                u[2] = xor3(s, yp, sign_mask(cw[m][1], 2 * j));
                u[3] = xor3(s, ym, sign_mask(cw[m][1], 2 * j + 1));
                u[4] = xor3(s, zp, sign_mask(cw[m][2], 2 * j));
                u[5] = xor3(s, zm, sign_mask(cw[m][2], 2 * j + 1));
                const uint2 lv = lds_pair(pl_addr + (j * 4 + m) * 512);
                const uint32_t acc = accept_six(u, lv.x, lv.y);
                const uint32_t mask = (cm - static_cast<uint32_t>(m)) <= 15u ? valid_bits : 0u; // x = c - y - z on the lattice
                R[m][j] = s ^ (acc & mask);
            };
            site_op(2, R[1][jm], R[3][jp]);
            site_op(1, R[0][jm], R[2][jp]);
            site_op(3, R[2][jm], up_cur);
            dn_next = from_lane(addr_dn, R[3][j]); // for level c + 1
            site_op(0, dn_cur, R[1][jp]);
        }
        if (cb == 0) { PA_REG_STAMP(3) }
        if (cb == 16) { PA_REG_STAMP(4) }
    }
    PA_REG_STAMP(5)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(R[m][j], s_out, ((((j - m - t0) & 15u) << 8) + col) * 4, m * 4, 0);
    PA_REG_STAMP(6)
    if (!(with_energy & 1)) return;

    // ---- energy sum_{i,d} J s_i s_fwd and magnetisation sum_i s_i of the swept word (energy.rs:92-109) ----
    uint32_t cnt[3][7]; // bit-sliced per-slot counts over the lane's 64 sites: unsatisfied bonds (weights 1, 2), spins -1
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int k = 0; k < 7; ++k) cnt[s][k] = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        uint32_t qa[16], qb[16], qc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int jp = (j + 1) & 15;
            const uint32_t s = R[m][j], xp = R[m][jp];
            const uint32_t yp = dpp_mov<kFromAbove>(xp);
            const uint32_t zp = m < 3 ? R[(m + 1) & 3][jp] : from_lane(addr_up, R[0][jp]);
            const uint32_t u0 = xor3(s, xp, sign_mask(cw[m][0], 2 * j));
            const uint32_t u1 = xor3(s, yp, sign_mask(cw[m][1], 2 * j));
            const uint32_t u2 = xor3(s, zp, sign_mask(cw[m][2], 2 * j));
            qa[j] = xor3(u0, u1, u2);
            qb[j] = maj3(u0, u1, u2);
            qc[j] = s;
        }
        uint32_t p5[5];
        planes16(qa, p5);
        add_planes(cnt[0], p5);
        planes16(qb, p5);
        add_planes(cnt[1], p5);
        planes16(qc, p5);
        add_planes(cnt[2], p5);
        __builtin_amdgcn_sched_barrier(0); // (one column set's 48 words at a time)
    }
    const uint32_t r = lane & 3u;
    uint32_t tot[3][2][2]; // [set][jj][half]: 16-bit fields for the bit positions 2 r + jj (+ 16) and 2 r + jj + 8 (+ 24)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        uint32_t o9[9];
        quad_add_planes<7>(cnt[s], o9); // <= 256 per slot, alike in the quad's lanes
#pragma unroll
        for (uint32_t jj = 0; jj < 2; ++jj) {
            const uint32_t sh = 2 * r + jj;
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                lo += ((o9[k] >> sh) & 0x00010001u) << k;
                hi += ((o9[k] >> (sh + 8)) & 0x00010001u) << k;
            }
            // lanes of equal r: the four quads of the row, then the four rows
            lo += dpp_mov<0x124>(lo);
            hi += dpp_mov<0x124>(hi);
            lo += dpp_mov<0x128>(lo);
            hi += dpp_mov<0x128>(hi);
            lo += from_lane((lane ^ 16u) << 2, lo);
            hi += from_lane((lane ^ 16u) << 2, hi);
            lo += from_lane((lane ^ 32u) << 2, lo);
            hi += from_lane((lane ^ 32u) << 2, hi);
            tot[s][jj][0] = lo;
            tot[s][jj][1] = hi;
        }
    }
    uint32_t (*totals)[4] = reinterpret_cast<uint32_t (*)[4]>(lds_planes); // (the planes are dead)
    if (lane < 4) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int h = 0; h < 2; ++h) totals[(s * 2 + jj) * 2 + h][lane] = tot[s][jj][h];
    }
    __syncthreads();
    if (lane < 32 && w * 32 + lane < d.n_sys) {
        const uint32_t b = lane, rr = (b & 7u) >> 1, jj = b & 1u, q = b >> 3;
        uint32_t v[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const uint32_t word = totals[(s * 2 + jj) * 2 + (q & 1u)][rr];
            v[s] = (q >> 1) ? (word >> 16) : (word & 0xFFFFu);
        }
        const uint32_t slot = w * 32 + lane;
        const int32_t unsat = static_cast<int32_t>(v[0] + 2 * v[1]);
        p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = nz - 2 * unsat;
        p.Msum[static_cast<size_t>(real) * d.n_sys + slot] = static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(v[2]);
    }
    PA_REG_STAMP(7)
}
#undef PA_REG_STAMP

hipError_t launch_sweep_reg(const SimDims &d, const DevPtrs &p, bool with_energy, hipStream_t stream, unsigned long long *stamps) {
    if (stamps)
        PA_LAUNCH(k_sweep_reg<true>, dim3(d.n_real * d.W), dim3(64), 32 * 1024, stream, d, p, with_energy ? 1 : 0, stamps);
    else
        PA_LAUNCH(k_sweep_reg<false>, dim3(d.n_real * d.W), dim3(64), 32 * 1024, stream, d, p, with_energy ? 1 : 0, stamps);
    return hipGetLastError();
}

} // namespace pa
