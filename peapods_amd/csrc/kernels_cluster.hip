// Overlap-cluster moves, Swendsen-Wang ("sw") inner variant (gfx950).
//
// Reference: clusters/overlap.rs houdayer_step (:153-339) and jorg_step (:348-553) with
// cluster_mode = sw, i.e. the union-find path: activate_bonds (clusters/utils.rs:225-241) visits
// every forward bond (site, dim) in order and asks should_bond, which consumes one draw of the
// pair's generator for every ELIGIBLE bond of the Joerg move (none for Houdayer); after
// uf_flatten_counts the sites are scanned in order and every cluster of more than one site
// receives a coin gen::<f32>() < 0.5 the first time one of its sites is met; clusters with
// coin 1 are flipped in both replicas.
//
// None of this depends on the order of the unions: the draw a bond sees is fixed by the number
// of eligible bonds before it in (site, dim) order, cluster membership is the connected
// components of the accepted bonds, and a cluster's coin is fixed by the rank of its smallest
// site among the multi-site clusters.  So the move runs as a parallel algorithm per pair:
//   eligibility bitmaps (ballots) -> prefix sums -> the pair's xoshiro stream generated in
//   blocks of 256 draws by jump-ahead -> accepted bonds -> connected components by min-label
//   hooking + pointer jumping in LDS -> cluster ranks -> coin draws -> flips (atomic XOR on the
//   two bit columns of the pair).
//
// The same machinery serves the Fortuin-Kasteleyn Swendsen-Wang update of single systems
// (clusters/fk.rs:102-196): eligible = satisfied non-zero bond, p = 1 - exp(-2 J s_i s_j / T), the
// system's own generator (its sweep cursor), and a coin for EVERY cluster, singletons included.
//
//   k_pairing      per realisation: the per-temperature shuffles of build_tasks (:35-56)
//   k_cluster_sw   workgroup = (realisation, temperature, pair) or (realisation, slot)
//   k_fk_wolff     FK single-cluster update: the reference's depth-first search, one lane per system
#include "engine.hpp"

#include <cstdlib>

#include "device_util.hpp"

namespace pa {

__device__ __forceinline__ uint32_t gen_range_u32(DevRng &rng, uint32_t range) {
    // rand 0.8.5 UniformInt<u32>::sample_single: widening multiply, zone rejection
    const uint32_t zone = (range << __clz(range)) - 1u;
    for (;;) {
        const uint32_t v = static_cast<uint32_t>(rng.next() >> 32);
        const uint64_t m = static_cast<uint64_t>(v) * range;
        if (static_cast<uint32_t>(m) <= zone) return static_cast<uint32_t>(m >> 32);
    }
}

// One thread per temperature: shuffle the replicas with pair generator t * n_pairs (reverse
// Fisher-Yates, SliceRandom::shuffle) and write the slots in the shuffled order
// (pair_slots[real][t][n_replicas]).
__global__ void k_pairing(SimDims d, DevPtrs p) {
    const uint32_t real = blockIdx.x, t = threadIdx.x;
    if (t >= d.n_temps) return;
    const uint32_t n_lanes = d.n_temps * d.n_pairs;
    uint64_t *st = p.pair_rng + (static_cast<size_t>(real) * n_lanes + static_cast<size_t>(t) * d.n_pairs) * 4;
    DevRng rng;
    rng.s0 = st[0]; rng.s1 = st[1]; rng.s2 = st[2]; rng.s3 = st[3];
    uint32_t order[32]; // n_replicas <= 32 (checked on the host)
    for (uint32_t k = 0; k < d.n_replicas; ++k) order[k] = k;
    for (uint32_t i = d.n_replicas - 1; i >= 1; --i) {
        const uint32_t j = gen_range_u32(rng, i + 1);
        const uint32_t tmp = order[i];
        order[i] = order[j];
        order[j] = tmp;
    }
    // slots of the shuffled order; groups of g consecutive entries form the tasks of temperature t
    uint32_t *slots = p.pair_slots + (static_cast<size_t>(real) * d.n_temps + t) * d.n_replicas;
    for (uint32_t k = 0; k < d.n_replicas; ++k) slots[k] = order[k] * d.n_temps + t;
    st[0] = rng.s0; st[1] = rng.s1; st[2] = rng.s2; st[3] = rng.s3;
}

// k_pair_stream: the draws every cluster task may consume, made before the cluster kernel runs.
// A task's bond and cluster draws are consecutive outputs of ITS generator (the pair's, or for FK
// the system's), at positions the cluster kernel knows only after counting eligible bonds -- but
// every task of a call uses the same positions 0 .. N z' + N at most, and entering a stream at
// position 64 u needs the same matrix T^(64 u) for every task.  So: one workgroup per (u, 256
// tasks) stages that 32 KB table in the LDS once, each lane jumps its task's state and steps 64
// times.  (Inside the cluster kernel the same jumps were gathers from L2-resident tables, two per
// 64 draws and lane: 60 % of the Joerg move's time.)  Unit couplings: four bit planes over the
// stream positions, [plane][2 words per unit] per task -- plane 0: draw < first threshold of the move
// (FK 1 - e^(-2/T), Joerg 1 - e^(-4/T), CMR blue 1 - r^2), plane 1: draw < 1 - e^(-2/T) (CMR red),
// planes 2 / 3: bits 30 / 31 of the 32-bit draw (bit 31 clear <=> coin "< 0.5"; both = rand's u8
// gen_range(0..=3)); planes 1 and 2 only for CMR.  Other couplings: the 24-bit draw itself.
// ---------------------------------------------------------------------------
// A workgroup serves kStreamRun consecutive units with ONE table: it enters the stream at unit u and steps through the
// following ones (1 / 2 / 4 / 8 / 16 / 32 units per workgroup: 49 / 46 / 44 / 54 / 55 / 97 us for Joerg at 12^3 x 6144
// pairs -- at four the table loads and the lone-wave stepping balance; a short stream takes one unit per workgroup).
static uint32_t stream_run() {
    static const uint32_t run = [] {
        const char *e = dev_env("PEAPODS_STREAM_RUN");
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 64 ? static_cast<uint32_t>(v) : 4u;
    }();
    return run;
}
// SECOND: the second-threshold and bit-30 planes as well (the two-phase CMR move reads all four).
template <bool RAW, bool SECOND>
__global__ __launch_bounds__(256) void k_pair_stream(SimDims d, DevPtrs p, uint32_t n_units, const uint32_t *first_thr, int fk, uint32_t kStreamRun) {
    extern __shared__ uint32_t tab[]; // [64][2][16][4] nibble tables of T^(64 u)
    const uint32_t n_runs = (n_units + kStreamRun - 1) / kStreamRun;
    const uint32_t u = (blockIdx.x % n_runs) * kStreamRun, chunk = blockIdx.x / n_runs;
    const uint32_t n_here = min(kStreamRun, n_units - u); // units this workgroup draws
    const uint32_t tid = threadIdx.x;
    const uint32_t per_real = fk ? d.n_sys : d.n_temps * d.n_pairs;
    const uint32_t task = chunk * 256 + tid, n_tasks = d.n_real * per_real;
    if (u > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.jump_unit + static_cast<size_t>(u) * (64 * 16 * 8));
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = tid; i < 64 * 16 * 2; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    if (task >= n_tasks) return;
    const uint32_t real = task / per_real, local = task % per_real;
    const uint32_t t = fk ? local % d.n_temps : local / d.n_pairs;
    uint32_t st[8];
    if (fk) {
        load_cursor(p.cursor, real, d.SP, local, st);
    } else {
        const uint64_t *state = p.pair_rng + static_cast<size_t>(task) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st[2 * k] = static_cast<uint32_t>(state[k]);
            st[2 * k + 1] = static_cast<uint32_t>(state[k] >> 32);
        }
    }
    if (u > 0) jump_state_lds(st, tab);
    Rng32 rng;
    rng.load(st);
    const size_t stride = static_cast<size_t>(cluster_stream_units(d)) * 64; // positions a task's stream has room for
    if (RAW) {
        uint32_t *out = p.stream_raw + static_cast<size_t>(task) * stride + u * 64;
        for (uint32_t j = 0; j < 64 * n_here; j += 4) {
            uint4 v;
            v.x = rng.next_top() >> 8;
            v.y = rng.next_top() >> 8;
            v.z = rng.next_top() >> 8;
            v.w = rng.next_top() >> 8;
            *reinterpret_cast<uint4 *>(out + j) = v;
        }
    } else {
        // Four bit planes over the stream positions (bit q & 31 of word q >> 5): draw < first threshold, draw < second,
        // bits 30 and 31 of the 32-bit draw.  A lane owns whole words, so a plane costs it two instructions per draw
        // (compare into the carry, add-with-carry shifts it in) or one (funnel shift for a bit of the draw), and the
        // cluster workgroups copy words instead of classifying a byte per draw.
        const uint32_t first = first_thr ? first_thr[t] : 0u;
        const uint32_t second = p.fk_thr[t];
        const uint32_t plane_words = static_cast<uint32_t>(stride / 32);
        uint32_t *out = reinterpret_cast<uint32_t *>(p.stream_bits + static_cast<size_t>(task) * stride) + u * 2;
        for (uint32_t w = 0; w < 2 * n_here; ++w) {
            uint32_t lt1 = 0, lt2 = 0, b30 = 0, b31 = 0; // the word's first draw ends up in bit 31: reversed below
#pragma unroll 8
            for (int j = 0; j < 32; ++j) {
                const uint32_t top = rng.next_top(), draw = top >> 8;
                asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(lt1) : "v"(draw), "v"(first) : "vcc");
                if (SECOND) {
                    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(lt2) : "v"(draw), "v"(second) : "vcc");
                    b30 = __builtin_amdgcn_alignbit(b30, top << 1, 31);
                }
                b31 = __builtin_amdgcn_alignbit(b31, top, 31);
            }
            out[w] = __brev(lt1);
            if (SECOND) {
                out[plane_words + w] = __brev(lt2);
                out[2 * plane_words + w] = __brev(b30);
            }
            out[3 * plane_words + w] = __brev(b31);
        }
    }
}

// k_slot_bitmaps: slot_bits[real][slot][MW] = the spins of one system as a bitmap over the sites (1 <=> -1).
// The cluster workgroups each need the bit columns of two (or `group`) systems; reading them out of the
// slot-sliced tile costs a full word per site and system (32x the bits needed, 50-60 us per call on 6144 pairs),
// so the tile is transposed once per call: 64 sites x 32 slots per wave by 32 ballots.
__global__ __launch_bounds__(256) void k_slot_bitmaps(SimDims d, DevPtrs p) {
    const uint32_t n_chunks = (d.N + 255) / 256;
    const uint32_t chunk = blockIdx.x % n_chunks, w = (blockIdx.x / n_chunks) % d.W, real = blockIdx.x / (n_chunks * d.W);
    const uint32_t lane = threadIdx.x & 63u, i0 = chunk * 256 + (threadIdx.x & ~63u), i = i0 + lane;
    if (i0 >= d.N) return;
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

constexpr uint32_t kSwThreads = 256;

// Exclusive prefix sum of v over the workgroup's threads (tid order); returns the exclusive
// prefix, *total receives the sum.  scratch: [kSwThreads / 64 + 1] dwords of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *scratch, uint32_t *total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= static_cast<uint32_t>(off)) incl += up;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
    for (uint32_t k = 0; k < kSwThreads / 64; ++k) {
        if (k < wave) base += scratch[k];
        sum += scratch[k];
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}

// The two per-site arrays of the union-find.  They live in the LDS; on a lattice whose task does not fit there
// (GLOBAL: 128 x 128 = 16384 sites need 128 KB for them alone) they live in an uncached scratch area of device memory,
// one slice per workgroup, and every access goes through the L2 (relaxed agent-scope loads / stores next to the same
// atomics the LDS form uses): the lock-free union mixes compare-and-swaps with plain reads of the same words, and a
// read served by the CU's L1 would not see another wave's swap.
template <bool GLOBAL> struct SiteWords {
    uint32_t *base;
    struct Ref {
        uint32_t *q;
        __device__ __forceinline__ operator uint32_t() const {
            return GLOBAL ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
        __device__ __forceinline__ const Ref &operator=(uint32_t v) const {
            if (GLOBAL) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *q = v;
            return *this;
        }
        __device__ __forceinline__ const Ref &operator=(const Ref &o) const { return *this = static_cast<uint32_t>(o); }
        __device__ __forceinline__ uint32_t *operator&() const { return q; }
    };
    __device__ __forceinline__ Ref operator[](size_t i) const { return Ref{base + i}; }
};

template <bool GLOBAL> struct SwLds {
    SiteWords<GLOBAL> label; // [N] component label (smallest site after convergence)
    SiteWords<GLOBAL> aux;   // [N] multi-site flag of roots
    uint32_t *act;     // [MW] active sites (the pair's spins differ)
    uint32_t *spa;     // [MW] spin of the pair's first system (1 <=> -1)
    uint32_t *spb;     // [MW] spin of the pair's second system
    uint32_t *bond;    // [nn][MW] eligible, later accepted, forward bonds per direction (at least two rows)
    uint32_t *blue;    // [nn][MW] CMR: accepted blue bonds, kept for the grey clusters
    uint32_t *acc;     // [nn][MW] accepted bonds of the current phase; then the coin maps of the flips
    uint32_t *thr;     // [N nn] non-unit couplings: threshold of every eligible bond by its draw rank
    uint32_t *wbase;   // [MW + 1] exclusive prefix (eligible bonds / drawing roots) per 32-site word
    uint32_t *roots;   // [MW] roots that draw
    uint32_t *flip_a;  // [MW] accumulated flips of the first system
    uint32_t *flip_b;  // [MW] accumulated flips of the second system
    uint32_t *bits;    // [n_bit_words] accept / coin bits by draw position
    uint32_t *bits2;   // [n_bit_words] second bit of the CMR grey draw
    uint32_t *wrapm;   // [3][MW] implicit neighbours: the site's forward neighbour lies through the periodic wrap
    uint32_t *scratch; // [8]
    uint32_t *misc;    // [16] 0 changed, 1 seed / root, 2 draws, 3-4 winding x/y, 5 large, 6 bonds, 8-11 top four
};

__host__ __device__ inline size_t cluster_sw_bit_words(const SimDims &d) {
    return (static_cast<size_t>(d.N) * d.nn + 255) / 256 * 8 + 16;
}
__host__ __device__ inline size_t cluster_sw_lds_words(const SimDims &d, bool global_labels = false) {
    // (bond / blue / acc hold one row per forward direction, at least two: the coin maps of the flips take two rows of acc)
    return (global_labels ? 0 : 2 * static_cast<size_t>(d.N)) + (11 + 3 * static_cast<size_t>(d.nn < 2 ? 2 : d.nn)) * d.MW + 1 +
           2 * cluster_sw_bit_words(d) + 24 + (d.unit ? 0 : static_cast<size_t>(d.N) * d.nn);
}

enum { kSwHoudayer = 0, kSwJorg = 1, kSwFk = 2, kSwCmr = 3 };

// group: replicas per task (2 except for the experimental "houdN" moves: N replicas whose spins
// sum to zero on the active sites, all of them flipped).  wolff: single-cluster variant of a
// deterministic-bond move (find_seed, utils.rs:107-119, then flip the seed's component).
// LEVEL = 0 is the lean build of the plain moves (no reporting, no single-cluster forms, pairs only): the reporting and
// seed-drawing code costs ~100 VGPRs and most of the occupancy.  The lean Houdayer and Joerg builds fit 64 VGPRs and
// 19.7 KB of LDS at 12^3 -- eight workgroups per CU, so the 6144 pairs of BASELINE configs[4] are three full rounds of
// 2048 -- and what a pair's workgroup executes is kept short by the word-parallel phases below.
// GLOBAL: the union-find's per-site arrays in device memory (large lattices; the full build only).
template <int MODE, int LEVEL, bool GLOBAL = false>
__global__ __launch_bounds__(kSwThreads, LEVEL == 0 ? ((MODE == kSwHoudayer || MODE == kSwJorg) ? 8 : 4) : (LEVEL == 1 ? 3 : 2)) void k_cluster_sw(
    SimDims d, DevPtrs p, uint32_t group, int wolff, ClusterAux aux) {
    constexpr bool EXTRAS = LEVEL > 0; // sizes / top four and the single-cluster forms
    // the lean build is for unit couplings (launch_sw_build): the per-bond thresholds and raw streams of other couplings stay out of it
    const bool unit = LEVEL == 0 || d.unit != 0;
    constexpr bool FULL = LEVEL > 1;   // + observation metrics, winding, snapshots
    constexpr bool JORG = MODE == kSwJorg, FK = MODE == kSwFk, CMR = MODE == kSwCmr;
    constexpr bool STOCHASTIC = JORG || FK || CMR; // bonds are drawn
    constexpr uint32_t N_PHASES = CMR ? 2 : 1;     // CMR: blue clusters, then grey = blue + red bonds
    extern __shared__ uint32_t lds[];
    SwLds<GLOBAL> L;
    {
        uint32_t *q = lds;
        if (GLOBAL) {
            L.label.base = p.sw_scratch + static_cast<size_t>(blockIdx.x) * 2 * d.N;
            L.aux.base = L.label.base + d.N;
        } else {
            L.label.base = q; q += d.N;
            L.aux.base = q; q += d.N;
        }
        L.act = q; q += d.MW;
        L.spa = q; q += d.MW;
        L.spb = q; q += d.MW;
        const uint32_t bond_rows = d.nn < 2 ? 2u : d.nn;
        L.bond = q; q += bond_rows * d.MW;
        L.blue = q; q += bond_rows * d.MW;
        L.acc = q; q += bond_rows * d.MW;
        L.wbase = q; q += d.MW + 1;
        L.roots = q; q += d.MW;
        L.flip_a = q; q += d.MW;
        L.flip_b = q; q += d.MW;
        L.bits = q; q += cluster_sw_bit_words(d);
        L.bits2 = q; q += cluster_sw_bit_words(d);
        L.wrapm = q; q += 3 * d.MW;
        L.scratch = q; q += 8;
        L.misc = q; q += 16;
        L.thr = q;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // diagnostic (pa_debug_cluster_stamps; the pointer is null otherwise): s_memrealtime of the first lane at the phase
    // boundaries -- 0 entry, 1 bitmaps, 2 eligible bonds, 3 bond draws, 12 labelled by runs, 4 united,
    // 5 flattened, 6 reported, 7 drawing roots, 8 cluster draws, 9 flips decided, 11 end
    auto stamp = [&](int k) {
        if (p.slab_stamps && threadIdx.x == 0) p.slab_stamps[static_cast<size_t>(blockIdx.x) * 16 + k] = wall_clock64();
    };
    stamp(0);
    const uint32_t n_groups = FK ? 1u : d.n_replicas / group;
    const uint32_t n_lanes = FK ? d.n_sys : d.n_temps * n_groups;
    const uint32_t real = blockIdx.x / n_lanes, task = blockIdx.x % n_lanes;
    const uint32_t t = FK ? task % d.n_temps : task / n_groups;
    const uint32_t g = FK ? 0u : task % n_groups;
    const uint32_t *members = p.pair_slots + (static_cast<size_t>(real) * d.n_temps + t) * d.n_replicas + g * group;
    const uint32_t sa = FK ? task : members[0];
    const uint32_t sb = FK ? task : members[1];
    uint32_t *Sa = p.S + (static_cast<size_t>(real) * d.W + (sa >> 5)) * d.N;
    uint32_t *Sb = p.S + (static_cast<size_t>(real) * d.W + (sb >> 5)) * d.N;
    const uint32_t ba = sa & 31u, bb = sb & 31u;
    const uint32_t *jsite = p.jsite + static_cast<size_t>(real) * d.N;
    // generator of the task: pair generator t * n_pairs + g (overlap.rs:197), or the system's cursor
    uint64_t *rng_state = FK ? nullptr : p.pair_rng + (static_cast<size_t>(real) * d.n_temps * d.n_pairs +
                                                       static_cast<size_t>(t) * d.n_pairs + g) * 4;
    uint32_t *Stile = p.S + static_cast<size_t>(real) * d.W * d.N; // all words (groups above two)
    // snapshots are those of the first realisation's first group at every temperature (mod.rs:302,683-705)
    const bool snap_here = FULL && !FK && (aux.flags & kAuxSnapshot) && real == 0 && g == 0;
    const bool observe = FULL && (aux.flags & kAuxObserve) != 0, metrics = FULL && (aux.flags & kAuxMetrics) != 0;
    const uint32_t n_rounds = (d.N + kSwThreads - 1) / kSwThreads; // site i = r * 256 + tid
    // write the ballot of a predicate over sites r*256 + tid into a bitmap (two words per wave)
    auto put_ballot = [&](uint32_t *bitmap, uint32_t r, bool pred) {
        const unsigned long long votes = __ballot(pred);
        const uint32_t word = (r * kSwThreads + (tid & ~63u)) >> 5;
        if (lane == 0 && word < d.MW) bitmap[word] = static_cast<uint32_t>(votes);
        if (lane == 32 && word + 1 < d.MW) bitmap[word + 1] = static_cast<uint32_t>(votes >> 32);
    };
    auto get_bit = [&](const uint32_t *bitmap, uint32_t i) -> uint32_t { return (bitmap[i >> 5] >> (i & 31u)) & 1u; };
    // Implicit neighbours (SimDims::hc, at most three directions): the forward neighbour of a site lies at the
    // direction's regular offset or, for the sites flagged in the wrap bitmap, at the wrapped one.  With that the
    // per-site phases that only combine bits of a site and its neighbours run on whole 32-site words of the bitmaps:
    // "the neighbour's bit" is the bitmap shifted by the offset, picked per site by the wrap bitmap.
    const bool implicit = d.hc != 0;
    auto offset_of = [&](uint32_t dd, bool wrapped) -> int32_t { // in sites (dd may differ between lanes)
        const int32_t reg = dd == 0 ? d.hc_d0[0] : (dd == 1 ? d.hc_d0[2] : d.hc_d0[4]);
        const int32_t ext = dd == 0 ? d.hc_dd[0] : (dd == 1 ? d.hc_dd[2] : d.hc_dd[4]);
        return (reg + (wrapped ? ext : 0)) >> 2;
    };
    auto shifted_words = [&](auto word_at, uint32_t m, int32_t off) -> uint32_t { // word m of: bit i = bit i + off of the bitmap
        const int32_t s = static_cast<int32_t>(32 * m) + off, w = s >> 5;
        const uint32_t lo = (w >= 0 && w < static_cast<int32_t>(d.MW)) ? word_at(static_cast<uint32_t>(w)) : 0u;
        const uint32_t hi = (w + 1 >= 0 && w + 1 < static_cast<int32_t>(d.MW)) ? word_at(static_cast<uint32_t>(w + 1)) : 0u;
        return __builtin_amdgcn_alignbit(hi, lo, static_cast<uint32_t>(s) & 31u);
    };
    auto shifted = [&](const uint32_t *bitmap, uint32_t m, int32_t off) -> uint32_t {
        return shifted_words([&](uint32_t w) { return bitmap[w]; }, m, off);
    };
    // the direction whose regular forward neighbour is the next site (3: none)
    const uint32_t run_dd = !implicit ? 3u : (d.hc_d0[0] == 4 ? 0u : (d.nn > 1 && d.hc_d0[2] == 4 ? 1u : (d.nn > 2 && d.hc_d0[4] == 4 ? 2u : 3u)));
    if (implicit)
        for (uint32_t m = tid; m < 3 * d.MW; m += kSwThreads) L.wrapm[m] = p.sw_wrap[m];

    // ---- A: active sites and the systems' spins, from the per-slot bitmaps (k_slot_bitmaps)
    {
        const uint32_t *bits = p.slot_bits + static_cast<size_t>(real) * d.SP * d.MW;
        const uint32_t tail = (d.N & 31u) ? ((1u << (d.N & 31u)) - 1u) : 0xFFFFFFFFu;
        for (uint32_t m = tid; m < d.MW; m += kSwThreads) {
            const uint32_t live = m + 1 == d.MW ? tail : 0xFFFFFFFFu;
            const uint32_t wa = bits[static_cast<size_t>(sa) * d.MW + m], wb = bits[static_cast<size_t>(sb) * d.MW + m];
            uint32_t active = (FK || CMR) ? live : (wa ^ wb);
            if (EXTRAS && MODE == kSwHoudayer && group > 2) { // the group's spins sum to zero (overlap.rs:199-205): group / 2 are -1
                uint32_t c[6] = {0, 0, 0, 0, 0, 0}; // bit-sliced count of the members' -1 spins per site
                for (uint32_t k = 0; k < group; ++k) {
                    uint32_t carry = bits[static_cast<size_t>(members[k]) * d.MW + m];
#pragma unroll
                    for (int bq = 0; bq < 6; ++bq) {
                        const uint32_t t2 = c[bq] & carry;
                        c[bq] ^= carry;
                        carry = t2;
                    }
                }
                active = live;
#pragma unroll
                for (int bq = 0; bq < 6; ++bq) active &= ((group / 2) >> bq) & 1u ? c[bq] : ~c[bq];
            }
            L.act[m] = active;
            L.spa[m] = wa;
            L.spb[m] = wb;
        }
        if (snap_here) { // the pair's spins before the move (overlap.rs:211-221)
            for (uint32_t i = tid; i < d.N; i += kSwThreads) {
                p.snap_spins[(static_cast<size_t>(t) * 2 + 0) * d.N + i] = ((bits[static_cast<size_t>(sa) * d.MW + (i >> 5)] >> (i & 31u)) & 1u) ? -1 : 1;
                p.snap_spins[(static_cast<size_t>(t) * 2 + 1) * d.N + i] = ((bits[static_cast<size_t>(sb) * d.MW + (i >> 5)] >> (i & 31u)) & 1u) ? -1 : 1;
            }
        }
    }
    if (snap_here && tid == 0) {
        p.snap_sids[t * 4 + 0] = p.system_ids[static_cast<size_t>(real) * d.n_sys + sa];
        p.snap_sids[t * 4 + 1] = p.system_ids[static_cast<size_t>(real) * d.n_sys + sb];
        p.snap_sids[t * 4 + 2] = 0; // cluster ids are rebuilt only if this call gets as far as writing them
    }
    for (uint32_t m = tid; m < d.MW; m += kSwThreads) {
        L.flip_a[m] = 0;
        L.flip_b[m] = 0;
    }
    __syncthreads();

    stamp(1);
    // number of set entries before (i, dd) in (site, dim) order, from the per-word prefix wbase
    auto entries_before = [&](uint32_t i, uint32_t dd) -> uint32_t {
        const uint32_t m = i >> 5, k = i & 31u, below = (1u << k) - 1u;
        uint32_t n = L.wbase[m];
        for (uint32_t x = 0; x < d.nn; ++x) {
            const uint32_t wv = L.bond[x * d.MW + m];
            n += __popc(wv & below) + ((x < dd) ? ((wv >> k) & 1u) : 0u);
        }
        return n;
    };
    // exclusive scan of a per-word count into wbase[0..MW]; returns the total
    auto scan_words = [&](auto count_of_word) -> uint32_t {
        uint32_t carry = 0;
        for (uint32_t m0 = 0; m0 < d.MW; m0 += kSwThreads) {
            const uint32_t m = m0 + tid;
            const uint32_t c = m < d.MW ? count_of_word(m) : 0u;
            uint32_t part = 0;
            const uint32_t ex = block_exclusive_scan(c, L.scratch, &part);
            if (m < d.MW) L.wbase[m] = carry + ex;
            carry += part;
        }
        if (tid == 0) L.wbase[d.MW] = carry;
        __syncthreads();
        return carry;
    };

    // the generator's state at entry (read by the one lane that needs it, when it needs it: nothing else in this
    // workgroup writes it before the end)
    auto load_base_state = [&](uint32_t base_state[8]) {
        if (FK) {
            load_cursor(p.cursor, real, d.SP, sa, base_state);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                base_state[2 * k] = static_cast<uint32_t>(rng_state[k]);
                base_state[2 * k + 1] = static_cast<uint32_t>(rng_state[k] >> 32);
            }
        }
    };
    // Bits of the draws at stream positions [start, start + count), read from the task's pre-drawn stream
    // (k_pair_stream): kind 0 / 1 = below the move's first / second threshold, 2 = coin (draw < 1/2), 3 = the two top
    // bits of the draw (rand's u8 gen_range(0..=3), overlap.rs:827); with thr_by_rank (couplings that are not
    // unit: every bond has its own probability) the raw draw is compared with the threshold of the bond of rank
    // q - start.  Position q lands at bit q - 256 * (start / 256) of bits / bits2.
    const size_t stream_stride = static_cast<size_t>(cluster_stream_units(d)) * 64;
    const size_t stream_task = FK ? static_cast<size_t>(real) * d.n_sys + task
                                  : (static_cast<size_t>(real) * d.n_temps + t) * d.n_pairs + g;
    // Prefetch: the plane words a later draw_bits(start, ...) will want, one per thread from the word of position
    // 256 * (start / 256) on, requested while other work hides the trip to memory (the count is not known yet, the
    // start is).  pre[0] = the plane of kinds 0 / 1 / 2, or bit 30 for kind 3; pre[1] = bit 31 for kind 3.
    struct PlaneWords { uint32_t w0, v[2]; };
    const uint32_t plane_words = static_cast<uint32_t>(stream_stride / 32);
    auto prefetch_bits = [&](uint32_t start, uint32_t kind) -> PlaneWords {
        PlaneWords pre{0xFFFFFFFFu, {0u, 0u}};
        if (!unit) return pre;
        pre.w0 = (start / 256) * 8;
        const uint32_t w = pre.w0 + tid;
        if (w < 2 * aux.stream_units) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(p.stream_bits + stream_task * stream_stride) + w;
            pre.v[0] = src[(kind == 2 ? 3u : (kind == 3 ? 2u : kind)) * plane_words];
            if (kind == 3) pre.v[1] = src[3 * plane_words];
        }
        return pre;
    };
    // One word of a plane made on the spot, by the lane that needs it: the jump to the word's unit through the tables in
    // device memory, then its 32 draws -- what k_pair_stream would have stored there (the same planes, the same bit order).
    // Slow (a lane's serial jump), and rare: the launcher sizes the stream for what moves take, not for what they could.
    auto late_plane_words = [&](uint32_t word, uint32_t kind, uint32_t &v0, uint32_t &v1) {
        uint32_t js[8];
        load_base_state(js);
        if (word / 2) jump_state_lds(js, p.jump_unit + static_cast<size_t>(word / 2) * (64 * 16 * 8));
        Rng32 rng;
        rng.load(js);
        if (word & 1u)
            for (int j = 0; j < 32; ++j) (void)rng.next_top();
        const uint32_t first = JORG ? p.jorg_thr[t] : (CMR ? p.cmr_thr[t] : (FK ? p.fk_thr[t] : 0u)), second = p.fk_thr[t];
        v0 = v1 = 0;
        for (uint32_t j = 0; j < 32; ++j) {
            const uint32_t top = rng.next_top(), draw = top >> 8;
            const uint32_t b = kind == 0 ? (draw < first ? 1u : 0u) : (kind == 1 ? (draw < second ? 1u : 0u) : (kind == 2 ? top >> 31 : (top >> 30) & 1u));
            v0 |= b << j;
            if (kind == 3) v1 |= (top >> 31) << j;
        }
    };
    auto draw_bits = [&](uint32_t start, uint32_t count, uint32_t kind, const uint32_t *thr_by_rank, const PlaneWords &pre) {
        if (count == 0) return;
        const uint32_t base_q = (start / 256) * 256, end = start + count;
        // the tables reach cluster_stream_units units; past them only after > 64 rejections of a seed draw (p < 2^-50
        // each): refuse, never guess.  (Streams of other couplings are always made in full.)
        if (end > (unit ? cluster_stream_units(d) : aux.stream_units) * 64u) {
            if (tid == 0) atomicOr(p.stream_error, 1u);
            return;
        }
        if (unit) { // the planes k_pair_stream left: a word copy (positions before `start` in the first word are never read)
            const uint32_t *planes = reinterpret_cast<const uint32_t *>(p.stream_bits + stream_task * stream_stride);
            const uint32_t w0 = base_q / 32, n_words = (end - base_q + 31) / 32;
            for (uint32_t w = tid; w < n_words; w += kSwThreads) {
                const uint32_t at = w0 + w;
                uint32_t v0, v1 = 0;
                if (at >= 2 * aux.stream_units) {
                    late_plane_words(at, kind, v0, v1); // k_pair_stream stopped short of this word (it makes what a move usually takes)
                } else if (pre.w0 == w0 && w == tid) { // the word is here already
                    v0 = pre.v[0];
                    v1 = pre.v[1];
                } else {
                    const uint32_t *src = planes + at;
                    v0 = src[(kind == 2 ? 3u : (kind == 3 ? 2u : kind)) * plane_words];
                    if (kind == 3) v1 = src[3 * plane_words];
                }
                L.bits[w] = kind == 2 ? ~v0 : v0;
                if (kind == 3) L.bits2[w] = v1;
            }
            return;
        }
        const uint32_t *raw = p.stream_raw + stream_task * stream_stride;
        for (uint32_t q0 = base_q; q0 < end; q0 += kSwThreads) {
            const uint32_t q = q0 + tid;
            const bool live = q >= start && q < end;
            bool b0 = false, b1 = false;
            if (live) {
                const uint32_t draw = raw[q];
                if (kind == 3) {
                    b0 = (draw >> 22) & 1u;
                    b1 = (draw >> 23) & 1u;
                } else {
                    b0 = draw < (kind == 2 ? (1u << 23) : thr_by_rank[q - start]);
                }
            }
            const unsigned long long v0 = __ballot(b0), v1 = __ballot(b1);
            const uint32_t wd = (q0 - base_q + (tid & ~63u)) >> 5;
            if (lane == 0) {
                L.bits[wd] = static_cast<uint32_t>(v0);
                L.bits2[wd] = static_cast<uint32_t>(v1);
            }
            if (lane == 32) {
                L.bits[wd + 1] = static_cast<uint32_t>(v0 >> 32);
                L.bits2[wd + 1] = static_cast<uint32_t>(v1 >> 32);
            }
        }
    };
    // root of x with path halving (only non-roots are rewritten, and a non-root never becomes a
    // root again, so this cannot disturb a concurrent compare-and-swap on a root)
    auto find_root = [&](uint32_t x) -> uint32_t {
        for (;;) {
            const uint32_t up = L.label[x];
            if (up == x) return x;
            const uint32_t upup = L.label[up];
            if (upup == up) return up;
            L.label[x] = upup;
            x = upup;
        }
    };
    auto root_of = [&](uint32_t x) -> uint32_t { // the same, reading only (once the unions are over)
        for (;;) {
            const uint32_t up = L.label[x];
            if (up == x) return x;
            x = up;
        }
    };
    // the task's generator `at` draws after its position at entry (one lane)
    auto rng_at = [&](uint32_t at) -> DevRng {
        uint32_t js[8];
        load_base_state(js);
        if (at / 64) jump_state_lds(js, p.jump_unit + static_cast<size_t>(at / 64) * (64 * 16 * 8));
        Rng32 r32;
        r32.load(js);
        for (uint32_t k = 0; k < (at & 63u); ++k) (void)r32.next_top();
        uint32_t st[8];
        r32.store(st);
        DevRng rng;
        rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
        rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
        rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
        rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
        return rng;
    };
    auto store_rng = [&](const DevRng &rng) {
        if (FK) {
            const uint32_t st[8] = {static_cast<uint32_t>(rng.s0), static_cast<uint32_t>(rng.s0 >> 32),
                                    static_cast<uint32_t>(rng.s1), static_cast<uint32_t>(rng.s1 >> 32),
                                    static_cast<uint32_t>(rng.s2), static_cast<uint32_t>(rng.s2 >> 32),
                                    static_cast<uint32_t>(rng.s3), static_cast<uint32_t>(rng.s3 >> 32)};
            store_cursor(p.cursor, real, d.SP, sa, st);
        } else {
            rng_state[0] = rng.s0; rng_state[1] = rng.s1; rng_state[2] = rng.s2; rng_state[3] = rng.s3;
        }
    };
    // rand 0.8.5 gen_range(0..N) on usize: widening multiply + zone rejection
    auto draw_site = [&](DevRng &rng) -> uint32_t {
        const uint64_t range = d.N, zone = (range << __clzll(range)) - 1ull;
        uint64_t v;
        do v = rng.next(); while (v * range > zone);
        return static_cast<uint32_t>(__umul64hi(v, range));
    };
    // block-wide maximum of a 64-bit key (all threads receive it)
    auto block_max = [&](unsigned long long key) -> unsigned long long {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long other = __shfl_xor(key, off);
            key = other > key ? other : key;
        }
        __syncthreads();
        if (lane == 0) {
            L.scratch[2 * (tid >> 6)] = static_cast<uint32_t>(key);
            L.scratch[2 * (tid >> 6) + 1] = static_cast<uint32_t>(key >> 32);
        }
        __syncthreads();
        unsigned long long best = 0;
        for (uint32_t w = 0; w < kSwThreads / 64; ++w) {
            const unsigned long long v = (static_cast<unsigned long long>(L.scratch[2 * w + 1]) << 32) | L.scratch[2 * w];
            best = v > best ? v : best;
        }
        return best;
    };
    // Sizes of the components in L.label and, for an observed graph, its metrics (utils.rs:317-369):
    // histogram rows by atomics (integer sums commute), the four largest clusters by four block-wide
    // maximum extractions, winding by propagating displacements from every root over the bonds
    // and testing every bond for a non-zero mismatch (any closed walk with a net displacement shows
    // up on one of its bonds; the reference's displacement union-find records the same fact).
    auto report = [&](const uint32_t *links, const uint32_t *links2) {
        const uint32_t slot_idx = FK ? task : t * d.n_pairs + g;
        for (uint32_t i = tid; i < d.N; i += kSwThreads) L.aux[i] = 0;
        if (tid < 16) L.misc[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < d.N; i += kSwThreads) atomicAdd(&L.aux[L.label[i]], 1u);
        __syncthreads();
        unsigned long long *row =
            reinterpret_cast<unsigned long long *>(FK ? p.cs_fk_csd + (static_cast<size_t>(real) * d.n_temps + t) * (d.N + 1)
                                                      : p.cs_ov_csd + ((static_cast<size_t>(real) * aux.n_modes + aux.mode_idx) * d.n_temps + t) * (d.N + 1));
        unsigned long long *orow = reinterpret_cast<unsigned long long *>(
            p.obs_acc + ((static_cast<size_t>(real) * 4 + aux.kind) * d.n_temps + t) * (kObsWords + d.N + 1));
        uint32_t mine[4] = {0, 0, 0, 0}; // ascending
        uint32_t n_large = 0;
        const uint32_t large_from = (d.N + 19) / 20; // ceil(0.05 N)
        uint32_t n_small[3] = {0, 0, 0}; // clusters of one, two, three sites: counted per lane, one atomic per wave below
        for (uint32_t i = tid; i < d.N; i += kSwThreads) {
            if (L.label[i] != i) continue;
            const uint32_t c = L.aux[i];
            if (c <= 3) {
                ++n_small[c - 1];
            } else {
                atomicAdd(&row[c], 1ull);
                if (metrics) atomicAdd(&orow[kObsWords + c], 1ull);
            }
            n_large += c >= large_from;
            if (c > mine[0]) {
                mine[0] = c;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (mine[a] > mine[a + 1]) {
                        const uint32_t tmp = mine[a];
                        mine[a] = mine[a + 1];
                        mine[a + 1] = tmp;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t sum = n_small[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
            if (lane == 0 && sum) {
                atomicAdd(&row[c + 1], static_cast<unsigned long long>(sum));
                if (metrics) atomicAdd(&orow[kObsWords + c + 1], static_cast<unsigned long long>(sum));
            }
        }
        uint32_t top[4];
        for (int k = 0; k < 4; ++k) { // largest remaining value; ties go to the smallest thread
            const unsigned long long key = (static_cast<unsigned long long>(mine[3]) << 32) | (0xFFFFFFFFu - tid);
            const unsigned long long best = block_max(key);
            top[k] = static_cast<uint32_t>(best >> 32);
            if (best == key && mine[3] != 0) {
                mine[3] = mine[2];
                mine[2] = mine[1];
                mine[1] = mine[0];
                mine[0] = 0;
            }
        }
        if (!FK && tid == 0)
            for (int k = 0; k < 4; ++k)
                p.cs_top4_raw[(static_cast<size_t>(real) * d.n_temps * d.n_pairs + slot_idx) * 4 + k] = top[k];
        if (!metrics) {
            __syncthreads();
            return;
        }
        uint32_t n_bonds = 0;
        for (uint32_t m = tid; m < d.MW; m += kSwThreads)
            for (uint32_t x = 0; x < d.nn; ++x)
                n_bonds += __popc(links[x * d.MW + m] | (links2 ? links2[x * d.MW + m] : 0u));
        atomicAdd(&L.misc[5], n_large);
        atomicAdd(&L.misc[6], n_bonds);
        __syncthreads();
        if (aux.square) {
            constexpr uint32_t kUnknown = 0xFFFFFFFFu, kOrigin = 0x80008000u;
            auto linked = [&](uint32_t dd, uint32_t i) -> bool {
                return get_bit(links + dd * d.MW, i) || (links2 && get_bit(links2 + dd * d.MW, i));
            };
            for (uint32_t i = tid; i < d.N; i += kSwThreads) L.aux[i] = L.label[i] == i ? kOrigin : kUnknown;
            __syncthreads();
            for (;;) {
                if (tid == 0) L.misc[0] = 0;
                __syncthreads();
                bool moved = false;
                for (uint32_t i = tid; i < d.N; i += kSwThreads) {
                    const uint32_t here = L.aux[i];
                    if (here == kUnknown) continue;
                    for (uint32_t dd = 0; dd < 2; ++dd) {
                        const uint32_t step = dd == 0 ? 0x10000u : 1u;
                        const uint32_t f = p.nbr_site[i * 12 + 2 * dd], b = p.nbr_site[i * 12 + 2 * dd + 1];
                        if (linked(dd, i) && L.aux[f] == kUnknown) {
                            L.aux[f] = here + step;
                            moved = true;
                        }
                        if (linked(dd, b) && L.aux[b] == kUnknown) {
                            L.aux[b] = here - step;
                            moved = true;
                        }
                    }
                }
                if (moved) L.misc[0] = 1;
                __syncthreads();
                const bool again = L.misc[0] != 0;
                __syncthreads();
                if (!again) break;
            }
            bool wx = false, wy = false;
            for (uint32_t i = tid; i < d.N; i += kSwThreads)
                for (uint32_t dd = 0; dd < 2; ++dd)
                    if (linked(dd, i)) {
                        const uint32_t diff = L.aux[i] + (dd == 0 ? 0x10000u : 1u) - L.aux[p.nbr_site[i * 12 + 2 * dd]];
                        // the halves cannot borrow from each other: |displacement| < 2^15
                        wx = wx || ((diff + 0x8000u) >> 16) != 0;
                        wy = wy || (diff & 0xFFFFu) != 0;
                    }
            if (wx) L.misc[3] = 1;
            if (wy) L.misc[4] = 1;
            __syncthreads();
        }
        if (tid == 0) {
            atomicAdd(&orow[0], 1ull);
            for (int k = 0; k < 4; ++k) atomicAdd(&orow[1 + k], static_cast<unsigned long long>(top[k]));
            atomicAdd(&orow[5], static_cast<unsigned long long>(L.misc[6]));
            const uint32_t wx = L.misc[3], wy = L.misc[4];
            atomicAdd(&orow[6], static_cast<unsigned long long>(wx));
            atomicAdd(&orow[7], static_cast<unsigned long long>(wy));
            atomicAdd(&orow[8], static_cast<unsigned long long>(wx | wy));
            atomicAdd(&orow[9], static_cast<unsigned long long>(wx & wy));
            atomicAdd(&orow[10], static_cast<unsigned long long>(L.misc[5]));
        }
        __syncthreads();
    };
    // accepted bonds of a snapshot, for the reference-order union-find replay (k_uf_replay)
    auto export_bonds = [&](uint32_t which, const uint32_t *links) {
        uint32_t *out = p.snap_bonds + (static_cast<size_t>(t) * 2 + which) * 6 * d.MW;
        for (uint32_t x = 0; x < d.nn; ++x)
            for (uint32_t m = tid; m < d.MW; m += kSwThreads) out[x * d.MW + m] = links[x * d.MW + m];
    };
    // one flip step: decide(i, fa, fb) per site; both systems' bitmaps and the accumulated flips follow
    auto flip_step = [&](auto decide) {
        for (uint32_t r0 = 0; r0 < n_rounds; r0 += 4) { // (reads of four rounds, then their ballots)
          bool fa4[4] = {false, false, false, false}, fb4[4] = {false, false, false, false};
#pragma unroll
          for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t i = (r0 + q) * kSwThreads + tid;
            if (r0 + q < n_rounds && i < d.N) decide(i, fa4[q], fb4[q]);
          }
#pragma unroll
          for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t r = r0 + q;
            if (r >= n_rounds) break;
            const unsigned long long va = __ballot(fa4[q]), vb = __ballot(fb4[q]);
            const uint32_t word = (r * kSwThreads + (tid & ~63u)) >> 5;
            if (lane == 0 && word < d.MW) {
                L.flip_a[word] ^= static_cast<uint32_t>(va);
                L.flip_b[word] ^= static_cast<uint32_t>(vb);
                L.spa[word] ^= static_cast<uint32_t>(va);
                L.spb[word] ^= static_cast<uint32_t>(vb);
            }
            if (lane == 32 && word + 1 < d.MW) {
                L.flip_a[word + 1] ^= static_cast<uint32_t>(va >> 32);
                L.flip_b[word + 1] ^= static_cast<uint32_t>(vb >> 32);
                L.spa[word + 1] ^= static_cast<uint32_t>(va >> 32);
                L.spb[word + 1] ^= static_cast<uint32_t>(vb >> 32);
            }
          }
        }
    };
    uint32_t pos = 0; // draws of the generator consumed so far
    bool flips_applied = false; // the flips went to the bit columns as they were decided
    bool rng_stored = false; // a single-cluster form has already written the generator back
    if (EXTRAS && CMR && wolff) { // the single-cluster form of the union-find path draws its seed first (overlap.rs:691-695)
        if (tid == 0) {
            DevRng rng = rng_at(0);
            const DevRng before = rng;
            L.misc[1] = draw_site(rng);
            uint32_t used = 0; // gen_range consumed `used` draws (rejections are rare but possible)
            DevRng walk = before;
            while (walk.s0 != rng.s0 || walk.s1 != rng.s1 || walk.s2 != rng.s2 || walk.s3 != rng.s3) {
                (void)walk.next();
                ++used;
            }
            L.misc[2] = used;
        }
        __syncthreads();
        pos = L.misc[2];
    }
    const uint32_t cmr_seed = (EXTRAS && CMR && wolff) ? L.misc[1] : 0u;
    __syncthreads();
    for (uint32_t phase = 0; phase < N_PHASES; ++phase) {
        const uint32_t bond_kind = (CMR && phase == 1) ? 1u : 0u, cluster_kind = (CMR && phase == 1) ? 3u : 2u;
        // the draws of a stochastic move's bonds start at `pos`; a deterministic move goes straight to the cluster draws
        const PlaneWords pre_bonds = prefetch_bits(pos, bond_kind);
        PlaneWords pre_clusters = prefetch_bits(pos, cluster_kind);
        // ---- B: eligible forward bonds.  Houdayer: both ends active (overlap.rs:228-231).
        // Joerg: both ends active and s_a(i) s_a(j) J > 0 (:417-429).  FK: s_i s_j J > 0
        // (fk.rs:113-120).  CMR: blue = satisfied in both systems (:588-596), red = satisfied in
        // exactly one (:779-792).
        if (implicit) { // one thread per (direction, word of 32 sites)
            const uint32_t *cpl = p.sw_cpl + static_cast<size_t>(real) * 6 * d.MW;
            for (uint32_t cell = tid; cell < d.nn * d.MW; cell += kSwThreads) {
                const uint32_t dd = cell / d.MW, m = cell - dd * d.MW;
                const int32_t reg = offset_of(dd, false), wrp = offset_of(dd, true);
                const uint32_t wm = L.wrapm[cell];
                auto at_neighbour = [&](const uint32_t *bitmap) -> uint32_t {
                    return (shifted(bitmap, m, reg) & ~wm) | (shifted(bitmap, m, wrp) & wm);
                };
                uint32_t e = L.act[m] & at_neighbour(L.act);
                if (STOCHASTIC) {
                    const uint32_t neg = cpl[cell], nz = cpl[3 * d.MW + cell];
                    const uint32_t a_sat = ~(L.spa[m] ^ at_neighbour(L.spa) ^ neg);
                    if (CMR) {
                        const uint32_t b_sat = ~(L.spb[m] ^ at_neighbour(L.spb) ^ neg);
                        e &= nz & (phase == 0 ? (a_sat & b_sat) : (a_sat ^ b_sat));
                    } else {
                        e &= nz & a_sat;
                    }
                }
                L.bond[cell] = e;
            }
        } else
        for (uint32_t r0 = 0; r0 < n_rounds; r0 += 4) { // neighbour / coupling loads of 4 rounds in flight
            uint32_t jm[4], nbj[4][6];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t i = (r0 + q) * kSwThreads + tid;
                jm[q] = 0;
#pragma unroll
                for (uint32_t dd = 0; dd < 6; ++dd) nbj[q][dd] = 0;
                if (r0 + q < n_rounds && i < d.N) {
                    if (STOCHASTIC) jm[q] = jsite[i];
#pragma unroll
                    for (uint32_t dd = 0; dd < 6; ++dd)
                        if (dd < d.nn) nbj[q][dd] = p.nbr_site[i * 12 + 2 * dd];
                }
            }
            // reads first, for all four rounds, then the ballots: the bitmaps they write share the LDS with the ones read
            uint32_t eligible[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t r = r0 + q, i = r * kSwThreads + tid;
                if (r >= n_rounds) break;
                const uint32_t ic = i < d.N ? i : 0u;
                const uint32_t live = i < d.N ? get_bit(L.act, ic) : 0u;
                const uint32_t sa_i = STOCHASTIC ? get_bit(L.spa, ic) : 0u, sb_i = CMR ? get_bit(L.spb, ic) : 0u;
#pragma unroll
                for (uint32_t dd = 0; dd < 6; ++dd) {
                    if (dd >= d.nn) continue;
                    const uint32_t j = nbj[q][dd];
                    uint32_t ok = live & get_bit(L.act, j);
                    if (STOCHASTIC) {
                        const uint32_t nz = (jm[q] >> (12 + 2 * dd)) & 1u, neg = (jm[q] >> (2 * dd)) & 1u;
                        const uint32_t a_sat = (sa_i ^ get_bit(L.spa, j) ^ neg) ^ 1u;
                        if (CMR) {
                            const uint32_t b_sat = (sb_i ^ get_bit(L.spb, j) ^ neg) ^ 1u;
                            ok &= nz & (phase == 0 ? (a_sat & b_sat) : (a_sat ^ b_sat));
                        } else {
                            ok &= nz & a_sat;
                        }
                    }
                    eligible[q] |= ok << dd;
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t r = r0 + q;
                if (r >= n_rounds) break;
                for (uint32_t dd = 0; dd < d.nn; ++dd) put_ballot(L.bond + dd * d.MW, r, (eligible[q] >> dd) & 1u);
            }
        }
        stamp(2);
        __syncthreads();
        // ---- C, D, E (stochastic bonds): the bond at rank k of the eligible ones, in (site, dim)
        // order, sees draw pos + k of the generator
        if (STOCHASTIC) {
            const uint32_t n_eligible = scan_words([&](uint32_t m) {
                uint32_t c = 0;
                for (uint32_t x = 0; x < d.nn; ++x) c += __popc(L.bond[x * d.MW + m]);
                return c;
            });
            // draw < ceil(p 2^24)  <=>  gen::<f32>() < p (unit couplings: classified by k_pair_stream)
            if (!unit) { // every bond has its own probability: thresholds by draw rank
                const float temp = p.temps[t];
                for (uint32_t i = tid; i < d.N; i += kSwThreads)
                    for (uint32_t dd = 0; dd < d.nn; ++dd)
                        if (get_bit(L.bond + dd * d.MW, i)) {
                            const float j_abs = fabsf(p.Jf[(static_cast<size_t>(real) * d.N + i) * d.nn + dd]);
                            L.thr[entries_before(i, dd)] = FK ? fk_threshold(j_abs, temp)
                                                        : (CMR ? (phase == 0 ? cmr_blue_threshold(j_abs, temp) : fk_threshold(j_abs, temp))
                                                               : jorg_threshold(j_abs, temp));
                        }
                __syncthreads();
            }
            draw_bits(pos, n_eligible, bond_kind, unit ? nullptr : L.thr, pre_bonds);
            // the cluster draws follow the bond draws: their words travel while the components are built
            pre_clusters = prefetch_bits(pos + n_eligible, cluster_kind);
            __syncthreads();
            const uint32_t rel = pos - (pos / 256) * 256;
            uint32_t *accepted = (CMR && phase == 0) ? L.blue : L.acc;
            if (implicit) { // one thread per (direction, word): its eligible bonds one by one, each at its rank in the word
                for (uint32_t cell = tid; cell < d.nn * d.MW; cell += kSwThreads) {
                    const uint32_t dd = cell / d.MW, m = cell - dd * d.MW;
                    const uint32_t e0 = L.bond[m], e1 = d.nn > 1 ? L.bond[d.MW + m] : 0u, e2 = d.nn > 2 ? L.bond[2 * d.MW + m] : 0u;
                    const uint32_t first = dd > 0 ? e0 : 0u, second = dd > 1 ? e1 : 0u; // the site's bonds drawn before this one
                    const uint32_t q0 = rel + L.wbase[m];
                    uint32_t todo = dd == 0 ? e0 : (dd == 1 ? e1 : e2), taken = 0;
                    while (todo) {
                        const uint32_t k = static_cast<uint32_t>(__ffs(static_cast<int>(todo))) - 1u, below = (1u << k) - 1u;
                        todo &= todo - 1u;
                        const uint32_t q = q0 + __popc(e0 & below) + __popc(e1 & below) + __popc(e2 & below) + ((first >> k) & 1u) +
                                           ((second >> k) & 1u);
                        taken |= ((L.bits[q >> 5] >> (q & 31u)) & 1u) << k;
                    }
                    accepted[cell] = taken;
                }
            } else
            for (uint32_t r0 = 0; r0 < n_rounds; r0 += 4) { // (reads of four rounds, then their ballots)
                uint32_t taken[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t qq = 0; qq < 4; ++qq) {
                    const uint32_t i = (r0 + qq) * kSwThreads + tid;
                    if (r0 + qq >= n_rounds) break;
                    if (i < d.N) { // the site's bonds take consecutive draws from the rank of its first one
                        const uint32_t m = i >> 5, k = i & 31u, below = (1u << k) - 1u;
                        uint32_t q = rel + L.wbase[m], mine = 0;
#pragma unroll
                        for (uint32_t x = 0; x < 6; ++x)
                            if (x < d.nn) {
                                const uint32_t wv = L.bond[x * d.MW + m];
                                q += __popc(wv & below);
                                mine |= ((wv >> k) & 1u) << x;
                            }
                        const uint32_t lo = L.bits[q >> 5], hi = L.bits[(q >> 5) + 1];
                        uint32_t window = __builtin_amdgcn_alignbit(hi, lo, q & 31u); // draws q, q + 1, ...
#pragma unroll
                        for (uint32_t x = 0; x < 6; ++x)
                            if (x < d.nn) {
                                const uint32_t has = (mine >> x) & 1u;
                                taken[qq] |= (has & window) << x;
                                window >>= has;
                            }
                    }
                }
#pragma unroll
                for (uint32_t qq = 0; qq < 4; ++qq) {
                    if (r0 + qq >= n_rounds) break;
                    for (uint32_t dd = 0; dd < d.nn; ++dd) put_ballot(accepted + dd * d.MW, r0 + qq, (taken[qq] >> dd) & 1u);
                }
            }
            __syncthreads();
            pos += n_eligible;
        }

        stamp(3);
        // ---- F: connected components.  Roots are hooked under the smaller root, so a component's
        // final root is its smallest site -- the order in which the reference meets the clusters.
        // bonds of this phase: the eligible ones of a deterministic move, else the accepted ones
        // (CMR keeps the blue bonds: grey clusters are blue + red)
        const uint32_t *links = !STOCHASTIC ? L.bond : ((CMR && phase == 0) ? L.blue : L.acc);
        auto bonded = [&](uint32_t dd, uint32_t i) -> bool {
            return get_bit(links + dd * d.MW, i) || (CMR && phase == 1 && get_bit(L.blue + dd * d.MW, i));
        };
        if (tid == 0) L.misc[12] = L.misc[13] = 0; // packed bonds, overflow
        if (run_dd < 3) {
            // Bonds to the next site need no union: a site starts under the first site of its run of such bonds (at most
            // 32 sites back: a longer run is a chain of parents, which the finds below follow like any other).
            auto run_word = [&](uint32_t m) -> uint32_t { // bit k: site 32 m + k is bonded to site 32 m + k + 1
                uint32_t wv = links[run_dd * d.MW + m];
                if (CMR && phase == 1) wv |= L.blue[run_dd * d.MW + m];
                return wv & ~L.wrapm[run_dd * d.MW + m];
            };
            for (uint32_t i = tid; i < d.N; i += kSwThreads) {
                const uint32_t m = i >> 5, k = i & 31u;
                const uint32_t prev = m ? run_word(m - 1) : 0u;
                const uint32_t before = k ? __builtin_amdgcn_alignbit(run_word(m), prev, k) : prev; // bit 31: bond (i - 1, i)
                L.label[i] = i - static_cast<uint32_t>(__clz(static_cast<int>(~before)));
            }
            // the run bonds as a bitmap of their own (in `roots`, free until the clusters draw): union_word below
            for (uint32_t m = tid; m < d.MW; m += kSwThreads) L.roots[m] = run_word(m);
        } else {
            for (uint32_t i = tid; i < d.N; i += kSwThreads) L.label[i] = i;
        }
        stamp(12);
        __syncthreads();
        // Lock-free union: every bond is visited once; the larger root is hooked under the smaller
        // one with a compare-and-swap, retried from the new roots when another lane got there first
        // (labels only ever decrease, so every chain ends and the loop terminates).
        // (the two finds of a union walk their chains side by side: two reads in flight per step instead of one)
        auto find_roots = [&](uint32_t &x, uint32_t &y) {
            for (;;) {
                const uint32_t ux = L.label[x], uy = L.label[y];
                if (ux == x && uy == y) return;
                const uint32_t uux = L.label[ux], uuy = L.label[uy];
                if (uux != ux) L.label[x] = uux; // path halving; x is not a root here (a root's label is itself)
                if (uuy != uy) L.label[y] = uuy;
                x = uux;
                y = uuy;
            }
        };
        auto unite = [&](uint32_t ra, uint32_t rb) {
            find_roots(ra, rb);
            while (ra != rb) {
                if (ra < rb) {
                    const uint32_t tmp = ra;
                    ra = rb;
                    rb = tmp;
                }
                const uint32_t old = atomicCAS(&L.label[ra], ra, rb); // ra > rb
                if (old == ra) break;
                ra = old;
                find_roots(ra, rb);
            }
        };
        // The bonds that need a union are few and scattered over the (site, direction) grid -- a union per grid cell runs
        // its divergent loops for a handful of lanes, 3 x rounds times over.  So the cells with a bond are first packed
        // into a list (site | neighbour << 16, in the `aux` words, which are free until the flatten below), one wave-level
        // counter bump per round, and the unions then run over the list with every lane busy.  A list that would not fit
        // (capacity N; a dense graph has up to N nn bonds) sets the overflow flag and the unions run over the grid.
        // (on implicit lattices straight from the bond words: one thread per (direction, word), offsets by a scan)
        // The bonds of a (direction, word) cell that need a union.  In the run direction only the bonds through the wrap
        // are left.  In the others a bond at site i adds nothing when the previous site has the same bond and run bonds
        // join the two sites on both sides (a square of bonds: any three connect its corners) -- of a stretch of such
        // bonds between two runs only the first stays, which in a compact cluster is most of them gone.
        auto union_word = [&](uint32_t cell) -> uint32_t {
            auto bonds_of = [&](uint32_t c) -> uint32_t { return links[c] | ((CMR && phase == 1) ? L.blue[c] : 0u); };
            const uint32_t wv = bonds_of(cell);
            if (run_dd >= 3) return wv;
            const uint32_t dd = cell / d.MW, m = cell - dd * d.MW;
            if (dd == run_dd) return wv & L.wrapm[cell];
            const int32_t reg = offset_of(dd, false), wrp = offset_of(dd, true);
            auto squares = [&](uint32_t c, uint32_t mm) -> uint32_t { // bit j: bond at j, run bonds (j, j + 1) here and across
                const uint32_t wm = L.wrapm[c];
                return bonds_of(c) & L.roots[mm] & ((shifted(L.roots, mm, reg) & ~wm) | (shifted(L.roots, mm, wrp) & wm));
            };
            const uint32_t carry_in = m ? squares(cell - 1, m - 1) >> 31 : 0u;
            return wv & ~((squares(cell, m) << 1) | carry_in);
        };
        auto site_bonds = [&](uint32_t i, uint32_t nb[6]) -> uint32_t { // (table lattices) bit dd: bond (i, nb[dd]) needs a union
            uint32_t todo = 0;
            if (i >= d.N) return 0u;
#pragma unroll
            for (uint32_t dd = 0; dd < 6; ++dd)
                if (dd < d.nn && bonded(dd, i)) {
                    todo |= 1u << dd;
                    nb[dd] = p.nbr_site[i * 12 + 2 * dd];
                }
            return todo;
        };
        if (implicit) {
            // pack the cells [c_lo, c_hi) and unite over the list; false (nothing united) when the list would not fit
            auto union_pass = [&](uint32_t c_lo, uint32_t c_hi) -> bool {
                uint32_t carry = 0; // (every thread sees the same totals)
                for (uint32_t c0 = c_lo; c0 < c_hi; c0 += kSwThreads) {
                    const uint32_t cell = c0 + tid;
                    uint32_t wv = cell < c_hi ? union_word(cell) : 0u, part = 0;
                    uint32_t at = carry + block_exclusive_scan(static_cast<uint32_t>(__popc(wv)), L.scratch, &part);
                    carry += part;
                    if (wv && at + __popc(wv) <= d.N) {
                        const uint32_t dd = cell / d.MW, m = cell - dd * d.MW, wm = L.wrapm[cell];
                        const int32_t reg = offset_of(dd, false), wrp = offset_of(dd, true);
                        while (wv) {
                            const uint32_t k = static_cast<uint32_t>(__ffs(static_cast<int>(wv))) - 1u, i = 32 * m + k;
                            wv &= wv - 1u;
                            L.aux[at++] = i | (static_cast<uint32_t>(static_cast<int32_t>(i) + (((wm >> k) & 1u) ? wrp : reg)) << 16);
                        }
                    }
                }
                __syncthreads();
                if (carry > d.N) return false;
                for (uint32_t e = tid; e < carry; e += kSwThreads) {
                    const uint32_t pair = L.aux[e];
                    unite(pair & 0xFFFFu, pair >> 16);
                }
                return true;
            };
            // all directions at once when their bonds fit the list, else one direction at a time (at most N bonds each)
            if (!union_pass(0, d.nn * d.MW))
                for (uint32_t dd = 0; dd < d.nn; ++dd) {
                    union_pass(dd * d.MW, (dd + 1) * d.MW);
                    __syncthreads();
                }
        } else
        for (uint32_t r = 0; r < n_rounds; ++r) {
            const uint32_t i = r * kSwThreads + tid;
            uint32_t nb[6] = {0, 0, 0, 0, 0, 0};
            const uint32_t todo = site_bonds(i, nb);
            unsigned long long votes[6];
            uint32_t total = 0;
#pragma unroll
            for (uint32_t dd = 0; dd < 6; ++dd) {
                votes[dd] = dd < d.nn ? __ballot((todo >> dd) & 1u) : 0ull;
                total += static_cast<uint32_t>(__popcll(votes[dd]));
            }
            if (total == 0) continue; // (wave-uniform)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&L.misc[12], total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + total > d.N) {
                if (lane == 0) L.misc[13] = 1;
                continue;
            }
#pragma unroll
            for (uint32_t dd = 0; dd < 6; ++dd) {
                if (dd >= d.nn) continue;
                const uint32_t before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(votes[dd] >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(votes[dd]), 0u));
                if ((todo >> dd) & 1u) L.aux[base + before] = i | (nb[dd] << 16);
                base += static_cast<uint32_t>(__popcll(votes[dd]));
            }
        }
        __syncthreads();
        if (implicit) {
        } else if (L.misc[13] == 0) {
            const uint32_t n_list = L.misc[12];
            for (uint32_t e = tid; e < n_list; e += kSwThreads) {
                const uint32_t pair = L.aux[e];
                unite(pair & 0xFFFFu, pair >> 16);
            }
        } else {
            for (uint32_t r = 0; r < n_rounds; ++r) {
                const uint32_t i = r * kSwThreads + tid;
                uint32_t nb[6] = {0, 0, 0, 0, 0, 0};
                const uint32_t todo = site_bonds(i, nb);
#pragma unroll
                for (uint32_t dd = 0; dd < 6; ++dd)
                    if ((todo >> dd) & 1u) unite(i, nb[dd]);
            }
        }
        stamp(4);
        __syncthreads();
        if (implicit) {
            // flatten in place: with finds that only read, a label is written by its own thread alone, and whoever reads
            // it on the way meets the old parent or the root.  The roots go to a bitmap (`roots`; the run bonds are done).
            for (uint32_t r = 0; r < n_rounds; ++r) {
                const uint32_t i = r * kSwThreads + tid;
                bool is_root = false;
                if (i < d.N) {
                    const uint32_t x = root_of(i);
                    L.label[i] = x;
                    is_root = x == i;
                }
                put_ballot(L.roots, r, is_root);
            }
        } else {
            for (uint32_t i = tid; i < d.N; i += kSwThreads) L.aux[i] = find_root(i); // flatten (reads only)
            __syncthreads();
            for (uint32_t i = tid; i < d.N; i += kSwThreads) L.label[i] = L.aux[i];
        }
        __syncthreads();

        stamp(5);
        // ---- what the call reports (mod.rs:442-455,604-626): sizes of this phase's clusters -- for
        // CMR the blue ones (overlap.rs:725-742) -- and the accepted bonds of a snapshot
        const uint32_t *links2 = (CMR && phase == 1) ? L.blue : nullptr;
        const bool seed_first = EXTRAS && wolff && (MODE == kSwHoudayer || JORG); // reports only once a seed is found
        const bool reports = EXTRAS && (aux.flags & (kAuxSizes | kAuxMetrics)) != 0 && phase == 0;
        if (reports && !seed_first) report(links, links2);
        if (snap_here) {
            export_bonds(phase, links);
            if (!seed_first && tid == 0) p.snap_sids[t * 4 + 2] = 1;
        }
        if (observe) break; // the graph is measured, nothing moves (fk.rs:140-142, overlap.rs:292-294,743-745)

        if (wolff && seed_first) {
            // single-cluster form of a pair move on the union-find path: bonds first, then up to 64
            // probes for an active seed (utils.rs:107-119), flip its component (overlap.rs:245-272,446-472)
            if (tid == 0) {
                DevRng rng = rng_at(pos);
                uint32_t root = 0xFFFFFFFFu;
                for (int probe = 0; probe < 64 && root == 0xFFFFFFFFu; ++probe) {
                    const uint32_t i = draw_site(rng);
                    if (get_bit(L.act, i)) root = L.label[i];
                }
                L.misc[1] = root;
                store_rng(rng);
            }
            __syncthreads();
            const uint32_t root = L.misc[1];
            __syncthreads();
            if (root == 0xFFFFFFFFu) return; // no seed: nothing flips and nothing is reported
            for (uint32_t i = tid; i < d.N; i += kSwThreads)
                if (L.label[i] == root) {
                    if (EXTRAS && MODE == kSwHoudayer && group > 2) {
                        for (uint32_t k = 0; k < group; ++k)
                            atomicXor(&Stile[static_cast<size_t>(members[k] >> 5) * d.N + i], 1u << (members[k] & 31u));
                    } else {
                        atomicXor(&Sa[i], 1u << ba);
                        atomicXor(&Sb[i], 1u << bb);
                    }
                }
            if (reports) report(links, links2);
            if (snap_here && tid == 0) p.snap_sids[t * 4 + 2] = 1;
            return;
        }
        if (EXTRAS && wolff && FK) { // fk.rs:144-151: the seed is drawn after the bonds
            if (tid == 0) {
                DevRng rng = rng_at(pos);
                L.misc[1] = L.label[draw_site(rng)];
                store_rng(rng);
            }
            __syncthreads();
            const uint32_t root = L.misc[1];
            for (uint32_t i = tid; i < d.N; i += kSwThreads)
                if (L.label[i] == root) atomicXor(&Sa[i], 1u << ba);
            return;
        }
        if (EXTRAS && wolff && CMR) { // overlap.rs:748-755 (blue: both systems), :806-820 (grey: k in 1..=3)
            const uint32_t root = L.label[cmr_seed];
            uint32_t k = 3;
            if (phase == 1) {
                if (tid == 0) {
                    DevRng rng = rng_at(pos);
                    for (;;) { // rand 0.8.5 u8 gen_range(1..=3): u32 draw, zone = u32::MAX - (2^32 - 3) % 3
                        const uint32_t v = static_cast<uint32_t>(rng.next() >> 32);
                        const uint64_t m = static_cast<uint64_t>(v) * 3u;
                        if (static_cast<uint32_t>(m) <= 0xFFFFFFFFu - ((0xFFFFFFFFu - 3u + 1u) % 3u)) {
                            L.misc[7] = 1u + static_cast<uint32_t>(m >> 32);
                            break;
                        }
                    }
                    store_rng(rng);
                }
                __syncthreads();
                k = L.misc[7];
                rng_stored = true;
            }
            flip_step([&](uint32_t i, bool &fa, bool &fb) {
                const bool in = L.label[i] == root;
                fa = in && (k & 1u);
                fb = in && (k & 2u);
            });
            __syncthreads();
            continue;
        }
        stamp(6);
        // ---- G: clusters that draw -- more than one site for the overlap moves (overlap.rs:312-318,
        // :757-762, :822-828), every cluster for FK (fk.rs:182-192) -- ranked by their smallest site
        uint32_t n_clusters;
        if (implicit) {
            // a root's cluster has more than one site exactly when a bond touches the root: its own forward bonds, or the
            // forward bonds of its backward neighbours (the bond words shifted the other way) -- whole words again
            auto touched = [&](uint32_t m) -> uint32_t {
                auto bonds_of = [&](uint32_t c) -> uint32_t { return links[c] | ((CMR && phase == 1) ? L.blue[c] : 0u); };
                uint32_t any = 0;
#pragma unroll
                for (uint32_t dd = 0; dd < 3; ++dd)
                    if (dd < d.nn) {
                        auto plain = [&](uint32_t w) { return bonds_of(dd * d.MW + w) & ~L.wrapm[dd * d.MW + w]; };
                        auto wrapped = [&](uint32_t w) { return bonds_of(dd * d.MW + w) & L.wrapm[dd * d.MW + w]; };
                        // (an extent of one makes a site its own neighbour: such a bond joins nothing)
                        const int32_t reg = offset_of(dd, false), wrp = offset_of(dd, true);
                        if (reg != 0) any |= plain(m) | shifted_words(plain, m, -reg);
                        if (wrp != 0) any |= wrapped(m) | shifted_words(wrapped, m, -wrp);
                    }
                return any;
            };
            stamp(7);
            n_clusters = scan_words([&](uint32_t m) {
                const uint32_t drawing = FK ? L.roots[m] : (L.roots[m] & touched(m));
                L.roots[m] = drawing;
                return static_cast<uint32_t>(__popc(drawing));
            });
        } else {
            for (uint32_t i = tid; i < d.N; i += kSwThreads) L.aux[i] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < d.N; i += kSwThreads)
                if (L.label[i] != i) L.aux[L.label[i]] = 1;
            __syncthreads();
            for (uint32_t r = 0; r < n_rounds; ++r) {
                const uint32_t i = r * kSwThreads + tid;
                put_ballot(L.roots, r, i < d.N && L.label[i] == i && (FK || L.aux[i] != 0));
            }
            __syncthreads();
            stamp(7);
            n_clusters = scan_words([&](uint32_t m) { return static_cast<uint32_t>(__popc(L.roots[m])); });
        }
        // ---- H: the cluster draws continue the stream after the bond draws: a coin
        // gen::<f32>() < 0.5, or for the grey clusters of CMR gen_range(0..=3) = the top two bits
        const bool grey = CMR && phase == 1;
        draw_bits(pos, n_clusters, cluster_kind, nullptr, pre_clusters);
        __syncthreads();
        const uint32_t rel = pos - (pos / 256) * 256;
        stamp(8);
        // ---- I: flips (both systems for a coin; bit 0 / bit 1 of k for the two systems of a grey cluster)
        if (implicit) {
            // the draws of a word's roots are consecutive bits of the stream: deposit them at the roots' positions, once
            // per word, and a site's flip is one bit of that map at its label (`acc`: the bonds are done with)
            for (uint32_t m = tid; m < d.MW; m += kSwThreads) {
                uint32_t at = L.roots[m], coin_a = 0, coin_b = 0;
                const uint32_t q = rel + L.wbase[m];
                uint32_t draws_a = __builtin_amdgcn_alignbit(L.bits[(q >> 5) + 1], L.bits[q >> 5], q & 31u);
                uint32_t draws_b = grey ? __builtin_amdgcn_alignbit(L.bits2[(q >> 5) + 1], L.bits2[q >> 5], q & 31u) : 0u;
                while (at) {
                    const uint32_t k = static_cast<uint32_t>(__ffs(static_cast<int>(at))) - 1u;
                    at &= at - 1u;
                    coin_a |= (draws_a & 1u) << k;
                    coin_b |= (draws_b & 1u) << k;
                    draws_a >>= 1;
                    draws_b >>= 1;
                }
                L.acc[m] = coin_a;
                L.acc[d.MW + m] = grey ? coin_b : coin_a;
            }
            __syncthreads();
            if (N_PHASES == 1 && !(EXTRAS && MODE == kSwHoudayer && group > 2)) {
                // nothing reads the systems' bitmaps after a single-phase move: the flips go straight to the bit columns
                for (uint32_t i = tid; i < d.N; i += kSwThreads) {
                    const uint32_t root = L.label[i];
                    if (get_bit(L.acc, root)) {
                        atomicXor(&Sa[i], 1u << ba);
                        if (!FK) atomicXor(&Sb[i], 1u << bb);
                    }
                }
                flips_applied = true;
            } else
            flip_step([&](uint32_t i, bool &fa, bool &fb) {
                const uint32_t root = L.label[i];
                fa = get_bit(L.acc, root) != 0;
                fb = get_bit(L.acc + d.MW, root) != 0;
            });
        } else
        flip_step([&](uint32_t i, bool &fa, bool &fb) {
            const uint32_t root = L.label[i];
            if (FK || L.aux[root]) {
                const uint32_t q = rel + L.wbase[root >> 5] + __popc(L.roots[root >> 5] & ((1u << (root & 31u)) - 1u));
                const uint32_t b0 = (L.bits[q >> 5] >> (q & 31u)) & 1u;
                if (grey) {
                    fa = b0 != 0;
                    fb = ((L.bits2[q >> 5] >> (q & 31u)) & 1u) != 0;
                } else {
                    fa = fb = b0 != 0;
                }
            }
        });
        pos += n_clusters;
        __syncthreads();
    }
    stamp(9);
    // ---- the generator ends `pos` draws after where it started
    if (tid < 64 && !rng_stored) { // (the first wave: its lanes share the jump, lane 0 walks the rest and stores)
        uint32_t js[8];
        load_base_state(js);
        if (pos / 64) jump_state_wave(js, p.jump_unit + static_cast<size_t>(pos / 64) * (64 * 16 * 8));
        // (the tables of T^1 .. T^63 follow those of the units)
        if (pos & 63u) jump_state_wave(js, p.jump_unit + static_cast<size_t>(cluster_stream_units(d) + (pos & 63u)) * (64 * 16 * 8));
        if (tid == 0) {
            uint32_t st[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) st[k] = js[k];
            if (FK) {
                store_cursor(p.cursor, real, d.SP, sa, st);
            } else {
                for (int q = 0; q < 4; ++q) rng_state[q] = (static_cast<uint64_t>(st[2 * q + 1]) << 32) | st[2 * q];
            }
        }
    }
    // ---- apply the accumulated flips to the bit columns of the systems
    if (!flips_applied)
    for (uint32_t i = tid; i < d.N; i += kSwThreads) {
        if (EXTRAS && MODE == kSwHoudayer && group > 2) { // every replica of the group flips with the cluster
            if (get_bit(L.flip_a, i))
                for (uint32_t k = 0; k < group; ++k)
                    atomicXor(&Stile[static_cast<size_t>(members[k] >> 5) * d.N + i], 1u << (members[k] & 31u));
            continue;
        }
        if (get_bit(L.flip_a, i)) atomicXor(&Sa[i], 1u << ba);
        if (!FK && get_bit(L.flip_b, i)) atomicXor(&Sb[i], 1u << bb);
    }
    stamp(11);
}

// k_top4_accumulate: top4_accum[mode][t][k] += top4 / N over the pairs of a call, in pair order
// (simulation/mod.rs:725-735) -- one thread per (temperature, rank), so every f64 sum sees its
// terms in the reference's order.
__global__ void k_top4_accumulate(SimDims d, DevPtrs p, ClusterAux aux) {
    const uint32_t real = blockIdx.x;
    for (uint32_t tk = threadIdx.x; tk < d.n_temps * 4; tk += blockDim.x) {
        const uint32_t t = tk / 4, k = tk % 4;
        double *acc = p.cs_top4_acc + ((static_cast<size_t>(real) * aux.n_modes + aux.mode_idx) * d.n_temps + t) * 4 + k;
        double sum = *acc;
        for (uint32_t pr = 0; pr < d.n_pairs; ++pr)
            sum += static_cast<double>(p.cs_top4_raw[(static_cast<size_t>(real) * d.n_temps * d.n_pairs + t * d.n_pairs + pr) * 4 + k]) /
                   static_cast<double>(d.N);
        *acc = sum;
    }
}

// k_uf_replay: cluster ids of a snapshot in the reference's union-find convention.  The parallel
// components above name a cluster by its smallest site; the reference's ids are the roots its
// union-by-rank / path-halving forest ends with (utils.rs:68-91,222-238), which depend on the order
// of the unions.  One lane per temperature replays that order over the exported bonds.
__global__ __launch_bounds__(64) void k_uf_replay(SimDims d, DevPtrs p, int cmr) {
    extern __shared__ uint32_t lds[];
    uint32_t *parent = lds;                                        // [N]
    uint8_t *rank = reinterpret_cast<uint8_t *>(lds + d.N);        // [N]
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    if (p.snap_sids[t * 4 + 2] == 0) return; // nothing new was written for this slot: the ids stay
    for (uint32_t i = tid; i < d.N; i += 64) {
        parent[i] = i;
        rank[i] = 0;
    }
    __syncthreads();
    auto find = [&](uint32_t x) -> uint32_t {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    for (int which = 0; which < (cmr ? 2 : 1); ++which) {
        const uint32_t *bonds = p.snap_bonds + (static_cast<size_t>(t) * 2 + which) * 6 * d.MW;
        if (tid == 0) {
            for (uint32_t i = 0; i < d.N; ++i)
                for (uint32_t dd = 0; dd < d.nn; ++dd) {
                    if (!((bonds[dd * d.MW + (i >> 5)] >> (i & 31u)) & 1u)) continue;
                    const uint32_t rx = find(i), ry = find(p.nbr_site[i * 12 + 2 * dd]);
                    if (rx == ry) continue;
                    if (rank[rx] < rank[ry]) {
                        parent[rx] = ry;
                    } else {
                        parent[ry] = rx;
                        if (rank[rx] == rank[ry]) rank[rx] += 1;
                    }
                }
            for (uint32_t i = 0; i < d.N; ++i) parent[i] = find(i); // uf_flatten, utils.rs:289-293
        }
        __syncthreads();
        uint32_t *out = (cmr && which == 0) ? p.snap_blue : p.snap_ids;
        for (uint32_t i = tid; i < d.N; i += 64) out[static_cast<size_t>(t) * d.N + i] = parent[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_wolff_wave: the single-cluster moves whose bonds are drawn in the order a depth-first search
// meets them (Joerg, overlap.rs:510-545; FK, fk.rs:45-86), one WAVE per pair / system.  The search
// has to be replayed draw for draw, so it stays sequential per task -- but its steps need not be
// slow: the neighbour table, the coupling bits, the task's spin / activity / membership bitmaps
// and its stack all live in the LDS, lanes 0..11 test the twelve neighbour slots of the popped
// site at once (one ballot = the eligible bonds, in the reference's order f0, b0, f1, b1, ...),
// and only the accept draws walk through them one by one, the generator held uniformly by the
// wave.  A site costs a few LDS round trips instead of several global ones per neighbour
// (k_jorg / k_fk_wolff, one lane per task with stacks in global memory, remain for lattices
// whose tables do not fit: Joerg 2.1 ms -> 0.2 ms per call on 6144 pairs of 12^3).
// ---------------------------------------------------------------------------
// tasks (waves) per workgroup share the lattice tables.  A lone wave issues an instruction every five cycles or so and
// waits out two LDS round trips per site, so the throughput is the number of searches in flight: the workgroup size is
// the one that lets every task of the launch be resident at once if the LDS allows it (12^3, 6144 pairs: 12 waves and
// 78 KB per workgroup, two workgroups per CU, instead of 16 waves and 93 KB in two rounds)
__host__ __device__ inline uint32_t wolff_wave_bitmaps(int mode) { return mode == kSwFk ? 2u : (mode == kSwCmr ? 4u : 3u); }
// neighbour rows are padded to 8 (or 16) 16-bit slots: the row of a site is a shift away, not a multiplication
__host__ __device__ inline uint32_t wolff_wave_row_shift(const SimDims &d) { return 2 * d.nn <= 8 ? 3u : 4u; }
__host__ __device__ inline size_t wolff_wave_lds_words(const SimDims &d, uint32_t tasks, int mode) {
    return (static_cast<size_t>(d.N) << (wolff_wave_row_shift(d) - 1)) + d.N +
           tasks * (static_cast<size_t>(wolff_wave_bitmaps(mode)) * d.MW + (d.N + 1) / 2) + 16;
}
__host__ inline uint32_t wolff_wave_tasks(const SimDims &d, int mode) {
    const uint32_t n_tasks = mode == kSwFk ? d.n_sys : d.n_temps * d.n_pairs;
    if (const char *env = dev_env("PEAPODS_WAVE_TASKS")) {
        uint32_t tasks = static_cast<uint32_t>(std::max(1l, std::min(16l, std::strtol(env, nullptr, 10))));
        while (tasks > 1 && wolff_wave_lds_words(d, tasks, mode) * 4 > 160 * 1024) tasks /= 2;
        return tasks;
    }
    uint32_t best = 1;
    uint64_t best_rounds = ~0ull;
    for (uint32_t tasks : {16u, 12u, 8u, 6u, 4u, 2u, 1u}) {
        const size_t bytes = wolff_wave_lds_words(d, tasks, mode) * 4;
        if (bytes > 160 * 1024) continue;
        const uint64_t resident = std::min<uint64_t>(160 * 1024 / ((bytes + 511) & ~size_t(511)), 32 / tasks);
        const uint64_t groups = static_cast<uint64_t>(d.n_real) * ((n_tasks + tasks - 1) / tasks);
        const uint64_t rounds = (groups + 256 * resident - 1) / (256 * resident);
        if (rounds < best_rounds) {
            best_rounds = rounds;
            best = tasks;
        }
    }
    return best;
}

// a value every lane holds alike, told to the compiler (the generator of a search then lives in scalar registers)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32)); // (the builtin returns int)
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// UNIT: every coupling is +-1 or 0 (SimDims::unit) -- the per-bond probabilities (expf) stay out of the search loop's code
template <int MODE, bool UNIT>
__global__ __launch_bounds__(1024) void k_wolff_wave(SimDims d, DevPtrs p) {
    const uint32_t kWaveTasks = blockDim.x / 64;
    constexpr bool FK = MODE == kSwFk, CMR = MODE == kSwCmr;
    extern __shared__ uint32_t lds[];
    const uint32_t row_shift = wolff_wave_row_shift(d);
    uint16_t *nbr = reinterpret_cast<uint16_t *>(lds); // [N][8 or 16] neighbours by site: f0, b0, f1, b1, ... (padded with the site)
    uint32_t *jbits = lds + (static_cast<size_t>(d.N) << (row_shift - 1)); // [N] neg[0..11] | nz[12..23] of the twelve bond slots
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per_task = wolff_wave_bitmaps(MODE) * d.MW + (d.N + 1) / 2;
    uint32_t *mine = jbits + d.N + wave * per_task;
    // the task's bitmaps: spins of the first system, the cluster; Joerg: the active sites; CMR: spins of the second
    // system, the blue cluster (FK and CMR grow over every site: no activity map)
    uint32_t *spa = mine, *member = mine + d.MW;
    uint32_t *act = (FK || CMR) ? nullptr : mine + 2 * d.MW;
    uint32_t *spb = CMR ? mine + 2 * d.MW : nullptr, *blue = CMR ? mine + 3 * d.MW : nullptr;
    uint16_t *stack = reinterpret_cast<uint16_t *>(mine + wolff_wave_bitmaps(MODE) * d.MW);
    const uint32_t n_tasks = FK ? d.n_sys : d.n_temps * d.n_pairs;
    const uint32_t blocks_per_real = (n_tasks + kWaveTasks - 1) / kWaveTasks;
    const uint32_t real = blockIdx.x / blocks_per_real;
    const uint32_t task = __builtin_amdgcn_readfirstlane((blockIdx.x % blocks_per_real) * kWaveTasks + wave);
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.nbr_site); // [N][12] u16: the first 2 nn slots are used
        const uint32_t row_words = 1u << (row_shift - 1);
        for (uint32_t i = tid; i < d.N * row_words; i += blockDim.x) {
            const uint32_t site = i >> (row_shift - 1), pair = i & (row_words - 1u);
            lds[i] = pair < d.nn ? src[site * 6 + pair] : site * 0x10001u;
        }
        const uint32_t *js = p.jsite + static_cast<size_t>(real) * d.N;
        for (uint32_t i = tid; i < d.N; i += blockDim.x) jbits[i] = js[i];
    }
    __syncthreads();
    if (task >= n_tasks) return; // (no workgroup barrier below)
    const uint32_t t = FK ? task % d.n_temps : task / d.n_pairs;
    const uint32_t g = FK ? 0u : task % d.n_pairs;
    const uint32_t *members = p.pair_slots + (static_cast<size_t>(real) * d.n_temps + t) * d.n_replicas + g * 2;
    const uint32_t sa = FK ? task : members[0], sb = FK ? task : members[1];
    uint32_t *Sa = p.S + (static_cast<size_t>(real) * d.W + (sa >> 5)) * d.N;
    uint32_t *Sb = p.S + (static_cast<size_t>(real) * d.W + (sb >> 5)) * d.N;
    const uint32_t ba = sa & 31u, bb = sb & 31u;
    // ---- the task's bitmaps: active sites (Joerg: the pair's spins differ), spins of the first system, cluster
    for (uint32_t i0 = 0; i0 < d.N; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t va = i < d.N ? (Sa[i] >> ba) & 1u : 0u, vb = i < d.N ? (Sb[i] >> bb) & 1u : 0u;
        const unsigned long long ma = __ballot((FK || CMR) ? i < d.N : (va ^ vb) != 0), ms = __ballot(va != 0);
        const unsigned long long mb = __ballot(vb != 0);
        const uint32_t word = i0 >> 5;
        if (lane == 0) {
            if (!(FK || CMR)) act[word] = static_cast<uint32_t>(ma);
            spa[word] = static_cast<uint32_t>(ms);
            if (CMR) spb[word] = static_cast<uint32_t>(mb);
            member[word] = 0;
        }
        if (lane == 32 && word + 1 < d.MW) {
            if (!(FK || CMR)) act[word + 1] = static_cast<uint32_t>(ma >> 32);
            spa[word + 1] = static_cast<uint32_t>(ms >> 32);
            if (CMR) spb[word + 1] = static_cast<uint32_t>(mb >> 32);
            member[word + 1] = 0;
        }
    }
    auto bit = [&](const uint32_t *bitmap, uint32_t i) -> uint32_t { return (bitmap[i >> 5] >> (i & 31u)) & 1u; };
    // ---- the task's generator, held identically by every lane
    DevRng rng;
    uint64_t *pair_state = FK ? nullptr : p.pair_rng + (static_cast<size_t>(real) * d.n_temps * d.n_pairs + task) * 4;
    if (FK) {
        uint32_t st[8];
        load_cursor(p.cursor, real, d.SP, sa, st);
        rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
        rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
        rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
        rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
    } else {
        rng.s0 = pair_state[0]; rng.s1 = pair_state[1]; rng.s2 = pair_state[2]; rng.s3 = pair_state[3];
    }
    auto draw_site = [&]() -> uint32_t { // rand 0.8.5 gen_range(0..N) on usize: widening multiply + zone rejection
        const uint64_t range = d.N, zone = (range << __clzll(range)) - 1ull;
        uint64_t v;
        do v = rng.next(); while (v * range > zone);
        return static_cast<uint32_t>(__umul64hi(v, range));
    };
    // ---- seed: FK any site (fk.rs:54); Joerg up to 64 probes for an active one (utils.rs:107-119)
    uint32_t seed = 0xFFFFFFFFu;
    if (FK || CMR) { // (CMR: overlap.rs:847)
        seed = draw_site();
    } else {
        for (int probe = 0; probe < 64 && seed == 0xFFFFFFFFu; ++probe) {
            const uint32_t i = draw_site();
            if (bit(act, i)) seed = i;
        }
    }
    if (seed != 0xFFFFFFFFu) {
        const float temp = UNIT ? 0.0f : p.temps[t];
        rng.s0 = uniform64(rng.s0); rng.s1 = uniform64(rng.s1); rng.s2 = uniform64(rng.s2); rng.s3 = uniform64(rng.s3);
        // The search's draws come from the VECTOR unit (every lane steps the same state; lane 0's value is read back):
        // with six searches per SIMD the scalar unit, which serves a SIMD once in four cycles, is what the launch waits
        // for (~80 K scalar instructions per search against 27 K vector ones, profiles/r03_pmc_c4_jorg.csv), and a
        // generator step is ~20 of the ~35 scalar instructions a draw costs.
        constexpr bool kVectorRng = true;
        Rng32 vrng;
        auto to_vector = [&] {
            uint32_t st[8] = {static_cast<uint32_t>(rng.s0), static_cast<uint32_t>(rng.s0 >> 32), static_cast<uint32_t>(rng.s1),
                              static_cast<uint32_t>(rng.s1 >> 32), static_cast<uint32_t>(rng.s2), static_cast<uint32_t>(rng.s2 >> 32),
                              static_cast<uint32_t>(rng.s3), static_cast<uint32_t>(rng.s3 >> 32)};
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(st[k])); // (kept in vector registers: no scalar copy to fall back on)
            vrng.load(st);
        };
        auto to_scalar = [&] {
            uint32_t st[8];
            vrng.store(st);
#pragma unroll
            for (int k = 0; k < 8; ++k) st[k] = __builtin_amdgcn_readfirstlane(st[k]);
            rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
            rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
            rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
            rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
        };
        uint32_t top = 0;
        uint32_t next = 0; // the entry on top of the stack, kept in a register: most pops never read the LDS
        bool next_known = false;
        // depth-first search from whatever is on the stack (dfs_cluster, utils.rs:125-149).  phase 0: Joerg / FK
        // bonds, or CMR's blue bonds (both systems satisfied, 1 - r^2); phase 1: CMR's red bonds (exactly one
        // system satisfied, 1 - r), which a flip of both systems on the blue cluster leaves as they are
        auto search = [&](uint32_t phase) {
            const uint32_t slot_mask = (1u << (2 * d.nn)) - 1u; // the row's real slots
            const uint32_t threshold = __builtin_amdgcn_readfirstlane(FK ? p.fk_thr[t] : (CMR ? (phase == 0 ? p.cmr_thr[t] : p.fk_thr[t]) : p.jorg_thr[t]));
            while (top > 0) {
                --top;
                const uint32_t site = next_known ? next : __builtin_amdgcn_readfirstlane(stack[top]);
                next_known = false;
                const uint32_t jm = jbits[site];
                // lanes 0..11: is the bond of slot `lane` eligible?  Every bitmap word is requested before any is
                // looked at (no short-circuit: each would be another LDS round trip on the search's critical path)
                // (every lane reads: lanes past the row fold onto it, their votes are masked off below -- no exec juggling)
                const uint32_t sl = lane & ((1u << row_shift) - 1u);
                const uint32_t nb = nbr[(site << row_shift) + sl];
                const uint32_t w_site = spa[site >> 5], w_nb = spa[nb >> 5], w_mem = member[nb >> 5];
                const uint32_t w_act = (FK || CMR) ? 0xFFFFFFFFu : act[nb >> 5];
                const uint32_t w_site_b = CMR ? spb[site >> 5] : 0u, w_nb_b = CMR ? spb[nb >> 5] : 0u;
                const uint32_t neg = (jm >> sl) & 1u, nzb = (jm >> (12 + sl)) & 1u;
                const uint32_t a_unsat = ((w_site >> (site & 31u)) ^ (w_nb >> (nb & 31u)) ^ neg) & 1u;
                uint32_t wanted = (a_unsat ^ 1u) & (w_act >> (nb & 31u));
                if (CMR) {
                    const uint32_t b_unsat = ((w_site_b >> (site & 31u)) ^ (w_nb_b >> (nb & 31u)) ^ neg) & 1u;
                    wanted = phase == 0 ? ((a_unsat | b_unsat) ^ 1u) : (a_unsat ^ b_unsat);
                }
                const uint32_t eligible = nzb & ~(w_mem >> (nb & 31u)) & wanted & 1u;
                uint32_t todo = static_cast<uint32_t>(__ballot(eligible != 0)) & slot_mask;
                while (todo) {
                    const uint32_t q = __ffs(todo) - 1;
                    todo &= todo - 1;
                    const uint32_t target = __builtin_amdgcn_readlane(nb, q);
                    const uint32_t draw = kVectorRng ? __builtin_amdgcn_readfirstlane(vrng.next_top() >> 8)
                                                     : __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(rng.next() >> 40));
                    uint32_t limit = threshold; // draw < ceil(p 2^24)  <=>  gen::<f32>() < p
                    if (!UNIT) { // this bond's own probability; the coupling belongs to the bond's forward end
                        const uint32_t owner = (q & 1u) ? target : site;
                        const float j_abs = fabsf(p.Jf[(static_cast<size_t>(real) * d.N + owner) * d.nn + (q >> 1)]);
                        limit = FK ? fk_threshold(j_abs, temp)
                                   : (CMR ? (phase == 0 ? cmr_blue_threshold(j_abs, temp) : fk_threshold(j_abs, temp))
                                          : jorg_threshold(j_abs, temp));
                        // the same in every lane: said so, the search's control flow and the generator stay on the scalar unit
                        limit = __builtin_amdgcn_readfirstlane(limit);
                    }
                    if (draw < limit) {
                        if (lane == 0) { // (an LDS atomic without return: nothing to wait for; LDS operations of a wave stay in order)
                            atomicOr(&member[target >> 5], 1u << (target & 31u));
                            stack[top] = static_cast<uint16_t>(target);
                        }
                        ++top;
                        next = target;
                        next_known = true;
                        // another slot of this site leading to the same neighbour now finds it in the cluster
                        todo &= ~static_cast<uint32_t>(__ballot(nb == target)); // (bits past the row are clear already)
                    }
                }
            }
        };
        if (lane == 0) {
            member[seed >> 5] |= 1u << (seed & 31u);
            stack[0] = static_cast<uint16_t>(seed);
        }
        top = 1;
        next = seed;
        next_known = true;
        if (kVectorRng) to_vector();
        search(0);
        uint32_t k = 3; // which system(s) the final cluster flips: both, except for CMR's grey cluster
        if (CMR) {
            // the blue cluster has flipped in both systems (overlap.rs:889-895); the search goes on from every blue
            // site, pushed in ascending order (:897-902), over the red bonds; k in 1..=3 then picks the systems
            for (uint32_t m = lane; m < d.MW; m += 64) blue[m] = member[m];
            for (uint32_t i0 = 0; i0 < d.N; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool in = i < d.N && bit(member, i);
                const unsigned long long votes = __ballot(in);
                if (in) stack[top + __popcll(votes & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(i);
                top += static_cast<uint32_t>(__popcll(votes));
            }
            next_known = false;
            search(1);
            if (kVectorRng) to_scalar();
            for (;;) { // rand 0.8.5 u8 gen_range(1..=3): u32 draw, zone = u32::MAX - (2^32 - 3) % 3
                const uint32_t v = static_cast<uint32_t>(rng.next() >> 32);
                const uint64_t m = static_cast<uint64_t>(v) * 3u;
                if (static_cast<uint32_t>(m) <= 0xFFFFFFFFu - ((0xFFFFFFFFu - 3u + 1u) % 3u)) {
                    k = 1u + static_cast<uint32_t>(m >> 32);
                    break;
                }
            }
        }
        if (kVectorRng && !CMR) to_scalar();
        // ---- flip the cluster (the pair's / system's bit columns; other tasks own other bits of these words)
        for (uint32_t m = lane; m < d.MW; m += 64) {
            const uint32_t grey = member[m], was_blue = CMR ? blue[m] : 0u;
            uint32_t flip_a = ((k & 1u) ? grey : 0u) ^ was_blue, flip_b = ((k & 2u) ? grey : 0u) ^ was_blue;
            while (flip_a) {
                const uint32_t site = m * 32 + (__ffs(flip_a) - 1);
                flip_a &= flip_a - 1;
                atomicXor(&Sa[site], 1u << ba);
            }
            while (!FK && flip_b) {
                const uint32_t site = m * 32 + (__ffs(flip_b) - 1);
                flip_b &= flip_b - 1;
                atomicXor(&Sb[site], 1u << bb);
            }
        }
    }
    if (lane == 0) {
        if (FK) {
            const uint32_t st[8] = {static_cast<uint32_t>(rng.s0), static_cast<uint32_t>(rng.s0 >> 32),
                                    static_cast<uint32_t>(rng.s1), static_cast<uint32_t>(rng.s1 >> 32),
                                    static_cast<uint32_t>(rng.s2), static_cast<uint32_t>(rng.s2 >> 32),
                                    static_cast<uint32_t>(rng.s3), static_cast<uint32_t>(rng.s3 >> 32)};
            store_cursor(p.cursor, real, d.SP, sa, st);
        } else {
            pair_state[0] = rng.s0; pair_state[1] = rng.s1; pair_state[2] = rng.s2; pair_state[3] = rng.s3;
        }
    }
}

// ---------------------------------------------------------------------------
// k_fk_wolff: FK single-cluster update (fk.rs:42-86): seed = gen_range(0..N) of the system's
// generator, then dfs_cluster (utils.rs:125-149) with bonds drawn in the order the search meets
// them, so the search is replayed draw for draw: one lane per system, the tile in LDS, stacks in
// global scratch.  p = 1 - exp(-2/T) for a satisfied unit bond (host expf -> integer threshold).
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t fk_wolff_lds_words(const SimDims &d) {
    return static_cast<size_t>(d.N) * d.W + static_cast<size_t>(d.n_sys) * d.MW + 4;
}

__global__ __launch_bounds__(1024) void k_fk_wolff(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    uint32_t *tile = lds;                 // [W][N] spins
    uint32_t *member = lds + tile_words;  // [n_sys][MW] cluster bitmaps
    const uint32_t real = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * tile_words;
    const uint32_t *jsite = p.jsite + static_cast<size_t>(real) * d.N;

    copy_to_lds(tile, Sg, static_cast<uint32_t>(tile_words), tid, nthreads);
    for (size_t i = tid; i < static_cast<size_t>(d.n_sys) * d.MW; i += nthreads) member[i] = 0;
    __syncthreads();

    for (uint32_t slot = tid; slot < d.n_sys; slot += nthreads) {
        uint16_t *stack = p.dfs_stack + (static_cast<size_t>(real) * d.n_sys + slot) * d.N;
        uint32_t st[8];
        load_cursor(p.cursor, real, d.SP, slot, st);
        DevRng rng;
        rng.s0 = (static_cast<uint64_t>(st[1]) << 32) | st[0];
        rng.s1 = (static_cast<uint64_t>(st[3]) << 32) | st[2];
        rng.s2 = (static_cast<uint64_t>(st[5]) << 32) | st[4];
        rng.s3 = (static_cast<uint64_t>(st[7]) << 32) | st[6];
        const uint32_t *wa = tile + (slot >> 5) * d.N;
        const uint32_t ba = slot & 31u;
        const uint32_t threshold = p.fk_thr[slot % d.n_temps];
        uint32_t *mine = member + static_cast<size_t>(slot) * d.MW;
        // rand 0.8.5 gen_range(0..N) on usize: widening multiply + zone rejection
        uint32_t seed;
        {
            const uint64_t range = d.N;
            const uint64_t zone = (range << __clzll(range)) - 1ull;
            for (;;) {
                const uint64_t v = rng.next();
                if (v * range <= zone) {
                    seed = static_cast<uint32_t>(__umul64hi(v, range));
                    break;
                }
            }
        }
        uint32_t top = 0;
        mine[seed >> 5] |= 1u << (seed & 31u);
        stack[top++] = static_cast<uint16_t>(seed);
        while (top > 0) {
            const uint32_t site = stack[--top];
            const uint32_t jm = jsite[site];
            const NbrRow nbrs(p.nbr_site, site);
            const uint32_t s_site = (wa[site] >> ba) & 1u;
            for (uint32_t dir = 0; dir < 2 * d.nn; ++dir) { // f0, b0, f1, b1, ...
                const uint32_t nb = nbrs.at(dir);
                if ((mine[nb >> 5] >> (nb & 31u)) & 1u) continue;
                if (!((jm >> (12 + dir)) & 1u)) continue;                      // J = 0
                if ((s_site ^ ((wa[nb] >> ba) & 1u) ^ ((jm >> dir) & 1u)) != 0) continue; // s s J <= 0
                const uint32_t draw = static_cast<uint32_t>(rng.next() >> 40);
                uint32_t thr = threshold;
                if (!d.unit) { // p = 1 - exp(-2 |J| / T) of this bond (fk.rs:74), owner = forward end
                    const uint32_t owner = (dir & 1u) ? nb : site;
                    thr = fk_threshold(fabsf(p.Jf[(static_cast<size_t>(real) * d.N + owner) * d.nn + (dir >> 1)]),
                                       p.temps[slot % d.n_temps]);
                }
                if (draw < thr) {
                    mine[nb >> 5] |= 1u << (nb & 31u);
                    stack[top++] = static_cast<uint16_t>(nb);
                }
            }
        }
        st[0] = static_cast<uint32_t>(rng.s0); st[1] = static_cast<uint32_t>(rng.s0 >> 32);
        st[2] = static_cast<uint32_t>(rng.s1); st[3] = static_cast<uint32_t>(rng.s1 >> 32);
        st[4] = static_cast<uint32_t>(rng.s2); st[5] = static_cast<uint32_t>(rng.s2 >> 32);
        st[6] = static_cast<uint32_t>(rng.s3); st[7] = static_cast<uint32_t>(rng.s3 >> 32);
        store_cursor(p.cursor, real, d.SP, slot, st);
    }
    __syncthreads();
    for (uint32_t item = tid; item < d.n_sys * d.MW; item += nthreads) {
        const uint32_t slot = item / d.MW, mw = item % d.MW;
        uint32_t bits = member[item];
        while (bits) {
            const uint32_t site = mw * 32 + (__ffs(bits) - 1);
            bits &= bits - 1;
            atomicXor(&tile[(slot >> 5) * d.N + site], 1u << (slot & 31u));
        }
    }
    __syncthreads();
    for (size_t i = tid; i < tile_words; i += nthreads) Sg[i] = tile[i];
}

// ---------------------------------------------------------------------------
// k_cmr_wolff: CMR move, single-cluster variant without statistics (overlap.rs:845-965): seed =
// gen_range(0..N); depth-first search over doubly satisfied bonds (p = 1 - r^2) -> flip both
// systems; the search continues from every blue site over singly satisfied bonds (p = 1 - r) on
// the flipped spins; k = gen_range(1..=3) decides which system(s) flip on the grey cluster.
// Draws come in search order, so one lane per pair replays the search (as k_jorg does).
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t cmr_wolff_lds_words(const SimDims &d) {
    const size_t n_lanes = static_cast<size_t>(d.n_temps) * d.n_pairs;
    return static_cast<size_t>(d.N) * d.W + n_lanes * d.MW + 4;
}

__global__ __launch_bounds__(1024) void k_cmr_wolff(SimDims d, DevPtrs p) {
    extern __shared__ uint32_t lds[];
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    const uint32_t n_lanes = d.n_temps * d.n_pairs;
    uint32_t *tile = lds;                  // [W][N] spins
    uint32_t *member = lds + tile_words;   // [n_lanes][MW] cluster bitmaps
    const uint32_t real = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *Sg = p.S + static_cast<size_t>(real) * tile_words;
    const uint32_t *jsite = p.jsite + static_cast<size_t>(real) * d.N;

    copy_to_lds(tile, Sg, static_cast<uint32_t>(tile_words), tid, nthreads);
    for (size_t i = tid; i < static_cast<size_t>(n_lanes) * d.MW; i += nthreads) member[i] = 0;
    __syncthreads();

    for (uint32_t task = tid; task < n_lanes; task += nthreads) {
        uint16_t *stack = p.dfs_stack + (static_cast<size_t>(real) * n_lanes + task) * d.N;
        uint64_t *st = p.pair_rng + (static_cast<size_t>(real) * n_lanes + task) * 4;
        DevRng rng;
        rng.s0 = st[0]; rng.s1 = st[1]; rng.s2 = st[2]; rng.s3 = st[3];
        const uint32_t *members = p.pair_slots + (static_cast<size_t>(real) * d.n_temps + task / d.n_pairs) * d.n_replicas +
                                  2 * (task % d.n_pairs);
        const uint32_t sa = members[0], sb = members[1];
        uint32_t *wa = tile + (sa >> 5) * d.N, *wb = tile + (sb >> 5) * d.N;
        const uint32_t ba = sa & 31u, bb = sb & 31u;
        const uint32_t t = task / d.n_pairs;
        const uint32_t thr_blue = p.cmr_thr[t], thr_red = p.fk_thr[t];
        uint32_t *mine = member + static_cast<size_t>(task) * d.MW;
        auto in_cluster = [&](uint32_t i) { return (mine[i >> 5] >> (i & 31u)) & 1u; };
        // satisfied in system a / b: s_i s_j J > 0 for the bond in direction slot `dir` of `site`
        auto sat = [&](const uint32_t *wv, uint32_t bit, uint32_t site, uint32_t nb, uint32_t jm, uint32_t dir) {
            return (((wv[site] >> bit) ^ (wv[nb] >> bit) ^ (jm >> dir)) & 1u) == 0;
        };
        uint32_t seed;
        {
            const uint64_t range = d.N; // rand 0.8.5 gen_range(0..N) on usize
            const uint64_t zone = (range << __clzll(range)) - 1ull;
            for (;;) {
                const uint64_t v = rng.next();
                if (v * range <= zone) {
                    seed = static_cast<uint32_t>(__umul64hi(v, range));
                    break;
                }
            }
        }
        uint32_t top = 0;
        mine[seed >> 5] |= 1u << (seed & 31u);
        stack[top++] = static_cast<uint16_t>(seed);
        for (int phase = 0; phase < 2; ++phase) {
            while (top > 0) {
                const uint32_t site = stack[--top];
                const uint32_t jm = jsite[site];
                const NbrRow nbrs(p.nbr_site, site);
                for (uint32_t dir = 0; dir < 2 * d.nn; ++dir) { // f0, b0, f1, b1, ...
                    const uint32_t nb = nbrs.at(dir);
                    if (in_cluster(nb)) continue;
                    if (!((jm >> (12 + dir)) & 1u)) continue; // J = 0: neither system is satisfied
                    const bool a_sat = sat(wa, ba, site, nb, jm, dir), b_sat = sat(wb, bb, site, nb, jm, dir);
                    if (phase == 0 ? !(a_sat && b_sat) : (a_sat == b_sat)) continue;
                    const uint32_t draw = static_cast<uint32_t>(rng.next() >> 40);
                    uint32_t thr = phase == 0 ? thr_blue : thr_red;
                    if (!d.unit) { // r = exp(-2 |J| / T): blue 1 - r r, red 1 - r (overlap.rs:899-901, 926-929)
                        const uint32_t owner = (dir & 1u) ? nb : site;
                        const float j_abs = fabsf(p.Jf[(static_cast<size_t>(real) * d.N + owner) * d.nn + (dir >> 1)]);
                        thr = phase == 0 ? cmr_blue_threshold(j_abs, p.temps[t]) : fk_threshold(j_abs, p.temps[t]);
                    }
                    if (draw < thr) {
                        mine[nb >> 5] |= 1u << (nb & 31u);
                        stack[top++] = static_cast<uint16_t>(nb);
                    }
                }
            }
            if (phase == 0) {
                // flip the blue cluster in both systems, then restart the search from all its sites
                for (uint32_t mw = 0; mw < d.MW; ++mw) {
                    uint32_t bits = mine[mw];
                    while (bits) {
                        const uint32_t site = mw * 32 + (__ffs(bits) - 1);
                        bits &= bits - 1;
                        atomicXor(&wa[site], 1u << ba);
                        atomicXor(&wb[site], 1u << bb);
                        stack[top++] = static_cast<uint16_t>(site);
                    }
                }
            }
        }
        uint32_t k; // rand 0.8.5 u8 gen_range(1..=3): u32 draw, zone = u32::MAX - (2^32 - 3) % 3
        for (;;) {
            const uint32_t v = static_cast<uint32_t>(rng.next() >> 32);
            const uint64_t m = static_cast<uint64_t>(v) * 3u;
            if (static_cast<uint32_t>(m) <= 0xFFFFFFFFu - ((0xFFFFFFFFu - 3u + 1u) % 3u)) {
                k = 1u + static_cast<uint32_t>(m >> 32);
                break;
            }
        }
        for (uint32_t mw = 0; mw < d.MW; ++mw) {
            uint32_t bits = mine[mw];
            while (bits) {
                const uint32_t site = mw * 32 + (__ffs(bits) - 1);
                bits &= bits - 1;
                if (k & 1u) atomicXor(&wa[site], 1u << ba);
                if (k & 2u) atomicXor(&wb[site], 1u << bb);
            }
        }
        st[0] = rng.s0; st[1] = rng.s1; st[2] = rng.s2; st[3] = rng.s3;
    }
    __syncthreads();
    for (size_t i = tid; i < tile_words; i += nthreads) Sg[i] = tile[i];
}

size_t cmr_wolff_lds_bytes(const SimDims &d) { return cmr_wolff_lds_words(d) * 4; }

hipError_t launch_cmr_wolff(const SimDims &d, const DevPtrs &p, hipStream_t stream) {
    PA_LAUNCH(k_pairing, dim3(d.n_real), dim3(std::max(64u, (d.n_temps + 63u) & ~63u)), 0, stream, d, p);
    PA_LAUNCH(k_cmr_wolff, dim3(d.n_real), dim3(1024), cmr_wolff_lds_bytes(d), stream, d, p);
    return hipGetLastError();
}

// out[k] = exp(-k 2^-17), k < 2^24 ([-128, 0] in steps of 7.6e-6): compared with the host libm by
// the engine before the first cluster move with non-unit couplings.
__global__ void k_exp_selftest(float *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (1u << 24)) out[k] = dev_expf(-__fmul_rn(static_cast<float>(k), 0x1p-17f));
}

hipError_t launch_exp_selftest(float *out, hipStream_t stream) {
    PA_LAUNCH(k_exp_selftest, dim3((1u << 24) / 256), dim3(256), 0, stream, out);
    return hipGetLastError();
}

size_t cluster_sw_lds_bytes(const SimDims &d) { return cluster_sw_lds_words(d) * 4; }
size_t cluster_sw_lds_bytes_global_labels(const SimDims &d) { return cluster_sw_lds_words(d, true) * 4; }

hipError_t configure_cluster_kernels() {
    for (const void *fn : {reinterpret_cast<const void *>(k_cluster_sw<kSwHoudayer, 0>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwJorg, 0>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwFk, 0>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwCmr, 0>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwHoudayer, 2, true>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwJorg, 2, true>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwFk, 2, true>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwCmr, 2, true>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwHoudayer, 1>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwJorg, 1>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwFk, 1>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwCmr, 1>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwHoudayer, 2>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwJorg, 2>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwFk, 2>),
                           reinterpret_cast<const void *>(k_cluster_sw<kSwCmr, 2>),
                           reinterpret_cast<const void *>(k_fk_wolff),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwJorg, true>),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwFk, true>),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwCmr, true>),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwJorg, false>),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwFk, false>),
                           reinterpret_cast<const void *>(k_wolff_wave<kSwCmr, false>),
                           reinterpret_cast<const void *>(k_cmr_wolff),
                           reinterpret_cast<const void *>(k_uf_replay)}) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Which build of k_cluster_sw a call needs: 0 the plain move, 1 + sizes / top four and the single-cluster
// union-find forms, 2 + observation metrics, winding and snapshots (230 VGPRs).
static int cluster_sw_level(uint32_t flags, bool single_form) {
    if (flags & ~static_cast<uint32_t>(kAuxSizes)) return 2;
    return (flags != 0 || single_form) ? 1 : 0;
}
template <int MODE>
static void launch_sw_build(int level, dim3 grid, hipStream_t stream, const SimDims &d, const DevPtrs &p, uint32_t group,
                            int single, const ClusterAux &aux) {
    if (cluster_sw_lds_bytes(d) > 160 * 1024) { // the per-site arrays in device memory (p.sw_scratch: ensure_cluster_sw)
        PA_LAUNCH((k_cluster_sw<MODE, 2, true>), grid, dim3(kSwThreads), cluster_sw_lds_words(d, true) * 4, stream, d, p,
                           group, single, aux);
        return;
    }
    const size_t lds = cluster_sw_lds_bytes(d);
    if (level == 0 && !d.unit) level = 1; // (the lean build assumes unit couplings)
    if (level == 0)
        PA_LAUNCH((k_cluster_sw<MODE, 0>), grid, dim3(kSwThreads), lds, stream, d, p, group, single, aux);
    else if (level == 1)
        PA_LAUNCH((k_cluster_sw<MODE, 1>), grid, dim3(kSwThreads), lds, stream, d, p, group, single, aux);
    else
        PA_LAUNCH((k_cluster_sw<MODE, 2>), grid, dim3(kSwThreads), lds, stream, d, p, group, single, aux);
}

hipError_t launch_cluster_sw(const SimDims &d, const DevPtrs &p, int oc_mode, uint32_t group, bool wolff,
                             const ClusterAux &aux_in, hipStream_t stream, bool paired) {
    if (!paired) {
        PA_LAUNCH(k_pairing, dim3(d.n_real), dim3(std::max(64u, (d.n_temps + 63u) & ~63u)), 0, stream, d, p);
        PA_LAUNCH(k_slot_bitmaps, dim3(d.n_real * d.W * ((d.N + 255) / 256)), dim3(256), 0, stream, d, p);
    }
    ClusterAux aux = aux_in;
    { // the generators' draws: bonds (stochastic moves) + one per cluster (twice for CMR) + the seed
        const uint32_t draws = (oc_mode == 1 ? 0u : d.N * d.nn) + (oc_mode == 3 ? 2u : 1u) * d.N + 64u;
        aux.stream_units = std::min(cluster_stream_units(d), (draws + 63u) / 64u + 1u);
        if (d.unit) {
            // The Houdayer move draws a coin per multi-site cluster: far fewer than the one per site it could take, so the
            // stream is made for N / 8 of them, and a pair that runs past it makes the words it lacks itself (k_cluster_sw,
            // late_plane_words): exact either way.  The moves that draw bonds keep their full streams: two replicas on
            // opposite sides of the global flip differ almost everywhere, and such a pair does draw most of its bonds
            // (Joerg at 12^3 with 45 of 110 units: the stream 38 -> 26 us, the cluster kernel 101 -> 108).
            uint32_t usual = oc_mode == 1 ? d.N / 8 + 64u : draws;
            if (const char *e = dev_env("PEAPODS_STREAM_UNITS")) usual = static_cast<uint32_t>(std::max(1, atoi(e))) * 64u - 64u;
            aux.stream_units = std::min(aux.stream_units, (usual + 63u) / 64u + 1u);
        }
        const uint32_t *first = oc_mode == 2 ? p.jorg_thr : (oc_mode == 3 ? p.cmr_thr : nullptr);
        const uint32_t kStreamRun = aux.stream_units <= 16 ? 1u : stream_run(); // (a short stream: every unit its own workgroup)
        const dim3 sgrid(((aux.stream_units + kStreamRun - 1) / kStreamRun) * ((d.n_real * d.n_temps * d.n_pairs + 255u) / 256u));
        if (!d.unit)
            PA_LAUNCH((k_pair_stream<true, false>), sgrid, dim3(256), 32 * 1024, stream, d, p, aux.stream_units, first, 0, kStreamRun);
        else if (oc_mode == 3) // CMR: both thresholds and the two-bit draws of the grey clusters
            PA_LAUNCH((k_pair_stream<false, true>), sgrid, dim3(256), 32 * 1024, stream, d, p, aux.stream_units, first, 0, kStreamRun);
        else
            PA_LAUNCH((k_pair_stream<false, false>), sgrid, dim3(256), 32 * 1024, stream, d, p, aux.stream_units, first, 0, kStreamRun);
    }
    const dim3 grid(d.n_real * d.n_temps * (d.n_replicas / group));
    // PA_OC_HOUDAYER = 1 has its single-cluster form in every build
    const int level = cluster_sw_level(aux.flags, wolff || group > 2);
    if (oc_mode == 2) // PA_OC_JORG
        launch_sw_build<kSwJorg>(level, grid, stream, d, p, 2u, wolff ? 1 : 0, aux);
    else if (oc_mode == 3) // PA_OC_CMR
        launch_sw_build<kSwCmr>(level, grid, stream, d, p, 2u, wolff ? 1 : 0, aux);
    else
        launch_sw_build<kSwHoudayer>(level, grid, stream, d, p, group, wolff ? 1 : 0, aux);
    return hipGetLastError();
}

size_t wolff_wave_lds_bytes(const SimDims &d, int mode) { return wolff_wave_lds_words(d, wolff_wave_tasks(d, mode), mode) * 4; }
size_t wolff_wave_lds_bytes(const SimDims &d) { return wolff_wave_lds_bytes(d, kSwCmr); } // the largest of the three forms

// Joerg single-cluster move, one wave per pair (after the pairing shuffle)
hipError_t launch_jorg_wave(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool paired) {
    if (!paired) PA_LAUNCH(k_pairing, dim3(d.n_real), dim3(std::max(64u, (d.n_temps + 63u) & ~63u)), 0, stream, d, p);
    const uint32_t n_tasks = d.n_temps * d.n_pairs, per_block = wolff_wave_tasks(d, kSwJorg);
    const dim3 grid(d.n_real * ((n_tasks + per_block - 1) / per_block)), block(64 * per_block);
    if (d.unit) PA_LAUNCH((k_wolff_wave<kSwJorg, true>), grid, block, wolff_wave_lds_bytes(d, kSwJorg), stream, d, p);
    else PA_LAUNCH((k_wolff_wave<kSwJorg, false>), grid, block, wolff_wave_lds_bytes(d, kSwJorg), stream, d, p);
    return hipGetLastError();
}

hipError_t launch_cmr_wave(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool paired) {
    if (!paired) PA_LAUNCH(k_pairing, dim3(d.n_real), dim3(std::max(64u, (d.n_temps + 63u) & ~63u)), 0, stream, d, p);
    const uint32_t n_tasks = d.n_temps * d.n_pairs, per_block = wolff_wave_tasks(d, kSwCmr);
    const dim3 grid(d.n_real * ((n_tasks + per_block - 1) / per_block)), block(64 * per_block);
    if (d.unit) PA_LAUNCH((k_wolff_wave<kSwCmr, true>), grid, block, wolff_wave_lds_bytes(d, kSwCmr), stream, d, p);
    else PA_LAUNCH((k_wolff_wave<kSwCmr, false>), grid, block, wolff_wave_lds_bytes(d, kSwCmr), stream, d, p);
    return hipGetLastError();
}

hipError_t launch_top4_accumulate(const SimDims &d, const DevPtrs &p, const ClusterAux &aux, hipStream_t stream) {
    PA_LAUNCH(k_top4_accumulate, dim3(d.n_real), dim3(128), 0, stream, d, p, aux);
    return hipGetLastError();
}

hipError_t launch_uf_replay(const SimDims &d, const DevPtrs &p, bool cmr, hipStream_t stream) {
    const size_t lds = static_cast<size_t>(d.N) * 5 + 16;
    PA_LAUNCH(k_uf_replay, dim3(d.n_temps), dim3(64), lds, stream, d, p, cmr ? 1 : 0);
    return hipGetLastError();
}

size_t fk_wolff_lds_bytes(const SimDims &d) { return fk_wolff_lds_words(d) * 4; }

// The depth-first single-cluster kernel serves calls that report nothing; a call that collects
// sizes takes the union-find path, which draws every bond first and the seed afterwards (fk.rs:45).
hipError_t launch_fk(const SimDims &d, const DevPtrs &p, bool wolff, const ClusterAux &aux_in, hipStream_t stream) {
    ClusterAux aux = aux_in;
    if (wolff && aux.flags == 0) { // depth-first single-cluster form: one wave per system if the tables fit the LDS
        const uint32_t per_block = wolff_wave_tasks(d, kSwFk);
        if (wolff_wave_lds_bytes(d, kSwFk) <= 160 * 1024 && d.N <= 65535) {
            const dim3 grid(d.n_real * ((d.n_sys + per_block - 1) / per_block)), block(64 * per_block);
            if (d.unit) PA_LAUNCH((k_wolff_wave<kSwFk, true>), grid, block, wolff_wave_lds_bytes(d, kSwFk), stream, d, p);
            else PA_LAUNCH((k_wolff_wave<kSwFk, false>), grid, block, wolff_wave_lds_bytes(d, kSwFk), stream, d, p);
        } else {
            PA_LAUNCH(k_fk_wolff, dim3(d.n_real), dim3(1024), fk_wolff_lds_bytes(d), stream, d, p);
        }
        return hipGetLastError();
    }
    // union-find forms (they read the per-slot bitmaps and the systems' pre-drawn streams)
    PA_LAUNCH(k_slot_bitmaps, dim3(d.n_real * d.W * ((d.N + 255) / 256)), dim3(256), 0, stream, d, p);
    {
        aux.stream_units = std::min(cluster_stream_units(d), (d.N * d.nn + d.N + 64u + 63u) / 64u + 1u);
        const uint32_t kStreamRun = stream_run();
        const dim3 sgrid(((aux.stream_units + kStreamRun - 1) / kStreamRun) * ((d.n_real * d.n_sys + 255u) / 256u));
        if (d.unit)
            PA_LAUNCH((k_pair_stream<false, false>), sgrid, dim3(256), 32 * 1024, stream, d, p, aux.stream_units, p.fk_thr, 1, kStreamRun);
        else
            PA_LAUNCH((k_pair_stream<true, false>), sgrid, dim3(256), 32 * 1024, stream, d, p, aux.stream_units, p.fk_thr, 1, kStreamRun);
    }
    launch_sw_build<kSwFk>(cluster_sw_level(aux.flags, wolff), dim3(d.n_real * d.n_sys), stream, d, p, 1u, wolff ? 1 : 0, aux);
    return hipGetLastError();
}

} // namespace pa
