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

__device__ __forceinline__ float dev_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -INFINITY;          // log(+-0)
        if (ix == 0x7f800000u) return x;              // log(inf)
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN;
        ix = __float_as_uint(__fmul_rn(x, 0x1p23f));  // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int32_t k = static_cast<int32_t>(tmp) >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = kLogfInvC[i], logc = kLogfLogC[i];
    const double z = static_cast<double>(__uint_as_float(iz));
    const double r = __dsub_rn(__dmul_rn(z, invc), 1.0);
    const double y0 = __dadd_rn(logc, __dmul_rn(static_cast<double>(k), 0x1.62e42fefa39efp-1));
    const double r2 = __dmul_rn(r, r);
    double y = __dadd_rn(__dmul_rn(0x1.5575b0be00b6ap-2, r), -0x1.ffffef20a4123p-2);
    y = __dadd_rn(__dmul_rn(-0x1.00ea348b88334p-2, r2), y);
    y = __dadd_rn(__dmul_rn(y, r2), __dadd_rn(y0, r));
    return static_cast<float>(y);
}

__device__ __forceinline__ float uniform_of_draw(uint32_t draw24) {
    return __fmul_rn(static_cast<float>(draw24), 0x1p-24f); // exact: gen::<f32>()
}
__device__ __forceinline__ float logit_arg(float u) { return __fdiv_rn(u, __fsub_rn(1.0f, u)); }

// out_ln[k] = ln(k 2^-24), out_logit[k] = ln(u / (1 - u)): compared with the host libm by the
// engine before the first general-coupling sweep.
__global__ void k_log_selftest(float *out_ln, float *out_logit) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << 24)) return;
    const float u = uniform_of_draw(k);
    out_ln[k] = dev_logf(u);
    out_logit[k] = dev_logf(logit_arg(u));
}

// s * j for a spin stored as bit `bit` of `word` (1 <=> -1): an exact sign flip.
__device__ __forceinline__ float signed_by_spin(float j, uint32_t word, uint32_t bit) {
    return __uint_as_float(__float_as_uint(j) ^ (((word >> bit) & 1u) << 31));
}

// ---------------------------------------------------------------------------
// k_rng_f: same work split as k_rng (4 waves share one site block and its jump table; lanes
// 0-31 / 32-63 are the 32 slots of two tile words).  Every lane walks the block's sites in
// order and stores the f32 threshold of its slot: thr_f[tile word][site][32 slots].
// ---------------------------------------------------------------------------
template <bool GIBBS>
__global__ __launch_bounds__(256) void k_rng_f(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t tab[]; // [64 * 16 * 8]
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
    float *out = p.thr_f + (static_cast<size_t>(tw_ok ? tw : 0) * d.N + begin) * 32 + bit;
    for (uint32_t j = 0; j < len; ++j) {
        const float u = uniform_of_draw(rng.next_top() >> 8);
        const float l = dev_logf(GIBBS ? logit_arg(u) : u);
        if (active) out[static_cast<size_t>(j) * 32] = __fmul_rn(half_t, l);
    }
    if (active && block == d.B - 1) {
        rng.store(st);
        store_cursor(p.cursor, real, d.SP, slot, st);
    }
}

// ---------------------------------------------------------------------------
// k_sweep_f: workgroup = (realisation, tile word).  A work item is (site of the current
// dependency level, slot); the 32 lanes of a half wave share a site, so its record, couplings
// and neighbour words are broadcast reads and the half wave's ballot is the site's flip word.
// !BIG: the word's spins and the realisation's couplings are staged in LDS.
//  BIG: both stay in global memory (spins through agent-scope atomics, as in k_sweep_big).
// ---------------------------------------------------------------------------
template <bool BIG>
__global__ __launch_bounds__(BIG ? 1024 : 256) void k_sweep_f(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    const uint32_t real = blockIdx.x / d.W, w = blockIdx.x % d.W;
    const uint32_t tid = threadIdx.x, n_threads = blockDim.x;
    const uint32_t tw = real * d.W + w;
    uint32_t *Sg = p.S + static_cast<size_t>(tw) * d.N;
    const float *Jg = p.Jf + static_cast<size_t>(real) * d.N * d.nn;
    const float *thr = p.thr_f + static_cast<size_t>(tw) * d.N * 32;
    const uint32_t n_pad = (d.N + 3u) & ~3u;
    uint32_t *sw = lds;                                   // [N]
    float *Jl = reinterpret_cast<float *>(lds + n_pad);   // [N * nn]
    if (!BIG) {
        copy_to_lds(sw, Sg, d.N, tid, n_threads);
        copy_to_lds(reinterpret_cast<uint32_t *>(Jl), reinterpret_cast<const uint32_t *>(Jg),
                    d.N * d.nn, tid, n_threads);
        __syncthreads();
    }
    const float *J = BIG ? Jg : Jl;
    const uint32_t live = d.n_sys - w * 32;
    const uint32_t valid_bits = live >= 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
    const uint32_t bit = tid & 31u, half = (tid >> 5) & 1u;
    const uint4 *rec = p.rec + static_cast<size_t>(real) * (d.N + 1) * (BIG ? 2 : 1);
    auto spin_word = [&](uint32_t site) -> uint32_t { return BIG ? coherent_load(Sg + site) : sw[site]; };

    for (uint32_t pass = 0; pass < d.n_passes; ++pass) {
        const uint32_t begin = p.pass_start[pass], end = p.pass_start[pass + 1];
        const uint32_t n_items = (end - begin) * 32;
        for (uint32_t item = tid; item < n_items; item += n_threads) {
            const uint32_t pos = begin + (item >> 5);
            uint32_t site, nb[6];
            if (BIG) {
                const BigRec r(rec[2 * pos], rec[2 * pos + 1]);
                site = r.site;
#pragma unroll
                for (int k = 0; k < 6; ++k) nb[k] = r.nb[k];
            } else {
                const SiteRec r(rec[pos]); // LDS byte offsets = 4 * index
                site = r.site >> 2;
#pragma unroll
                for (int k = 0; k < 6; ++k) nb[k] = r.nb[k] >> 2;
            }
            const uint32_t s = spin_word(site);
            float h = 0.0f;
            for (uint32_t dd = 0; dd < d.nn; ++dd) { // sweep.rs:11-17
                const uint32_t jf = nb[2 * dd], jb = nb[2 * dd + 1];
                h = __fadd_rn(h, signed_by_spin(J[site * d.nn + dd], spin_word(jf), bit));
                h = __fadd_rn(h, signed_by_spin(J[jb * d.nn + dd], spin_word(jb), bit));
            }
            // eng_change = -s_i * h (sweep.rs:43-44): s_i = +1 (bit clear) negates h
            const float e = signed_by_spin(h, ~s, bit);
            const bool flip = e >= thr[static_cast<size_t>(site) * 32 + bit];
            const unsigned long long votes = __ballot(flip);
            const uint32_t mask = (half ? static_cast<uint32_t>(votes >> 32) : static_cast<uint32_t>(votes)) & valid_bits;
            if (bit == 0 && mask) {
                if (BIG)
                    coherent_store(Sg + site, s ^ mask);
                else
                    sw[site] = s ^ mask;
            }
        }
        __syncthreads();
    }
    if (!BIG)
        for (uint32_t i = tid; i < d.N; i += n_threads) Sg[i] = sw[i];
}

// xw[tile word][site][4] = S_i ^ S_fwd(i, d) for d < nn (bit set <=> s_i s_j = -1).
__global__ __launch_bounds__(256) void k_bonds_f(SimDims d, DevPtrs p) {
    const uint32_t tw = blockIdx.y, real = tw / d.W;
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= d.N) return;
    const uint32_t *Sg = p.S + static_cast<size_t>(tw) * d.N;
    uint32_t site, f[3];
    if (d.big) {
        const uint4 *rec = p.rec + static_cast<size_t>(real) * (d.N + 1) * 2;
        const BigRec r(rec[2 * pos], rec[2 * pos + 1]);
        site = r.site; f[0] = r.nb[0]; f[1] = r.nb[2]; f[2] = r.nb[4];
    } else {
        const SiteRec r(p.rec[static_cast<size_t>(real) * (d.N + 1) + pos]);
        site = r.site >> 2; f[0] = r.nb[0] >> 2; f[1] = r.nb[2] >> 2; f[2] = r.nb[4] >> 2;
    }
    const uint32_t s = Sg[site];
    uint4 out = make_uint4(0, 0, 0, 0);
    out.x = s ^ Sg[f[0]];
    if (d.nn > 1) out.y = s ^ Sg[f[1]];
    if (d.nn > 2) out.z = s ^ Sg[f[2]];
    reinterpret_cast<uint4 *>(p.xw)[static_cast<size_t>(tw) * d.N + site] = out;
}

// One wave per tile word, lane = slot: the reference's sequential f32 sum over (site, direction)
// in typewriter order (energy.rs:92-107) and the magnetisation count.  Addresses are wave-uniform
// (scalar loads); only the additions depend on each other.
__global__ __launch_bounds__(64) void k_energy_f(SimDims d, DevPtrs p) {
    const uint32_t tw = blockIdx.x, real = tw / d.W, w = tw % d.W;
    const uint32_t bit = threadIdx.x & 31u;
    const uint4 *__restrict__ xw = reinterpret_cast<const uint4 *>(p.xw) + static_cast<size_t>(tw) * d.N;
    const float *__restrict__ J = p.Jf + static_cast<size_t>(real) * d.N * d.nn;
    const uint32_t *__restrict__ Sg = p.S + static_cast<size_t>(tw) * d.N;
    float total = 0.0f;
    uint32_t neg = 0;
    if (d.nn == 3) {
#pragma unroll 4
        for (uint32_t i = 0; i < d.N; ++i) {
            const uint4 x = xw[i];
            total = __fadd_rn(total, signed_by_spin(J[i * 3 + 0], x.x, bit));
            total = __fadd_rn(total, signed_by_spin(J[i * 3 + 1], x.y, bit));
            total = __fadd_rn(total, signed_by_spin(J[i * 3 + 2], x.z, bit));
            neg += (Sg[i] >> bit) & 1u;
        }
    } else {
        for (uint32_t i = 0; i < d.N; ++i) {
            const uint4 x = xw[i];
            const uint32_t xs[3] = {x.x, x.y, x.z};
            for (uint32_t dd = 0; dd < d.nn; ++dd)
                total = __fadd_rn(total, signed_by_spin(J[i * d.nn + dd], xs[dd], bit));
            neg += (Sg[i] >> bit) & 1u;
        }
    }
    const uint32_t slot = w * 32 + bit;
    if (threadIdx.x < 32 && slot < d.n_sys) {
        p.Esum[static_cast<size_t>(real) * d.n_sys + slot] = __float_as_int(total); // raw f32 bits
        p.Msum[static_cast<size_t>(real) * d.n_sys + slot] =
            static_cast<int32_t>(d.N) - 2 * static_cast<int32_t>(neg);
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
size_t sweep_f_lds_bytes(const SimDims &d) {
    return ((static_cast<size_t>(d.N) + 3) / 4 * 4 + static_cast<size_t>(d.N) * d.nn) * 4;
}

hipError_t configure_general_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_f<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_log_selftest(float *out_ln, float *out_logit, hipStream_t stream) {
    hipLaunchKernelGGL(k_log_selftest, dim3((1u << 24) / 256), dim3(256), 0, stream, out_ln, out_logit);
    return hipGetLastError();
}

hipError_t launch_rng_f(const SimDims &d, const DevPtrs &p, bool gibbs, hipStream_t stream) {
    const uint32_t n_pairs_tw = (d.n_real * d.W + 1) / 2;
    const uint32_t groups = (n_pairs_tw + 3) / 4;
    const size_t lds = 64 * 16 * 8 * 4;
    if (gibbs)
        hipLaunchKernelGGL(k_rng_f<true>, dim3(groups * d.B), dim3(256), lds, stream, d, p);
    else
        hipLaunchKernelGGL(k_rng_f<false>, dim3(groups * d.B), dim3(256), lds, stream, d, p);
    return hipGetLastError();
}

hipError_t launch_sweep_f(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                          hipStream_t stream) {
    if (do_sweep) {
        if (d.big)
            hipLaunchKernelGGL(k_sweep_f<true>, dim3(d.n_real * d.W), dim3(1024), 0, stream, d, p);
        else
            hipLaunchKernelGGL(k_sweep_f<false>, dim3(d.n_real * d.W), dim3(256), sweep_f_lds_bytes(d),
                               stream, d, p);
    }
    if (with_energy) {
        hipLaunchKernelGGL(k_bonds_f, dim3((d.N + 255) / 256, d.n_real * d.W), dim3(256), 0, stream, d, p);
        hipLaunchKernelGGL(k_energy_f, dim3(d.n_real * d.W), dim3(64), 0, stream, d, p);
    }
    return hipGetLastError();
}

} // namespace pa
