// Device-visible layout of one simulation (pa_sim) and the kernel launchers.
//
// Representation ("slot-sliced multispin coding"):
//   * One TILE per disorder realisation.  Bit k of a tile word belongs to
//     temperature SLOT k = replica * n_temps + t; a tile has W = ceil(n_sys/32)
//     words per lattice site.  S[real][w][site] holds the spins (bit = 1 <=> -1)
//     of 32 slots at one site, so every bitwise instruction advances 32 systems
//     that share the realisation's couplings.
//   * The reference's typewriter order is kept exactly: sites are grouped into
//     dependency levels (host_util.hpp) and levels run in order.
//   * The reference's per-system xoshiro256** stream is kept exactly: each
//     (slot, block-of-sites) lane enters its block by GF(2) jump-ahead from the
//     slot's cursor, draws sequentially, and a wave ballot transposes the 32
//     per-slot decisions of one site into the tile word the sweep consumes.
//   * Parallel tempering swaps bit columns (and cursors); system_ids/slot_of
//     keep the reference's slot->system permutation for every getter.
#pragma once

#include <cstdint>
#include <hip/hip_runtime.h>

namespace pa {

// Launch tracer (diagnostic, pa_sim_profile_step): every kernel launch of the library goes through PA_LAUNCH, which files
// an event in front of the launch while a trace is open on the calling thread -- a no-op otherwise (one thread-local
// load).  The name is the kernel expression as written at the launch site.
void trace_mark(const char *name, hipStream_t stream);
extern thread_local bool g_trace_on;
#define PA_LAUNCH(kernel, grid, block, lds, stream, ...)                                           \
    do {                                                                                           \
        if (::pa::g_trace_on) ::pa::trace_mark(#kernel, stream);                                   \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                          \
    } while (0)

// Developer switches (host_util.cpp): getenv(name) when PEAPODS_DEVELOPER=1 is set too, else null.
const char *dev_env(const char *name);

struct SimDims {
    uint32_t N;          // sites
    uint32_t W;          // words per site
    uint32_t SP;         // W * 32 (padded slot count)
    uint32_t n_sys;      // n_replicas * n_temps
    uint32_t n_temps;
    uint32_t n_replicas;
    uint32_t n_pairs;    // n_replicas / 2
    uint32_t n_real;
    uint32_t nn;         // forward directions z' (<= 3)
    uint32_t B;          // RNG blocks per sweep
    uint32_t BS;         // sites per RNG block
    uint32_t n_passes;   // schedule passes: dependency levels cut into chunks of <= 256 sites
    uint32_t MW;         // ceil(N / 32): words of the per-site parity bitmap
    uint32_t has_odd;    // 1 if any site has an odd number of non-zero bonds
    uint32_t n_edges;    // max(n_temps - 1, 1) (allocation stride of the PT edge counters)
    uint32_t thr_preshifted; // 1 if every threshold is < 2^24, so thr[6..12) = thr[0..6) << 8 is exact
    uint32_t big;        // 1: lattice too large for the LDS kernels (spins stay in global memory)
    uint32_t dense_nn;   // = nn when every bond of every realisation is non-zero, else 0
    uint32_t pipe_K;     // large unit-coupling lattices: slabs of the pipelined sweep (0 = not used)
    uint32_t n_levels;   // dependency levels of the schedule
    uint32_t unit;       // 1: every coupling is -1, 0 or +1 and every T/2 > 0 (cluster moves use integer thresholds)
    uint32_t general;    // 1: arbitrary f32 couplings / temperatures -> the reference's logarithmic rule
                         //    (kernels_general.hip); Esum then holds raw f32 bits of the energy sum
    uint32_t pipe_lds;   // 1: the slabs of the pipelined sweep keep their spins in the LDS (k_sweep_slab)
    uint32_t plane;      // sites of one hyperplane of the slowest dimension (halo window of a slab)
    uint32_t slab_lead;  // k_sweep_slab: levels a slab starts behind the slab it reads from (tuning knob PEAPODS_SLAB_LEAD)
    uint32_t planes_stride; // entries of planes / planes_hi per (realisation, word): N, or the padded
                         //    schedule length when the slab kernel reads them in schedule order
    // Implicit neighbours (k_sweep_hc): every one of the 2 nn neighbour slots f0,b0,f1,b1,f2,b2 of a site lies
    // at site + one of two offsets -- the regular one or the one through the periodic wrap -- so a site needs
    // one flag bit per slot instead of a 16-byte neighbour record (lattice.rs:44-99 builds tables; on a
    // hypercubic lattice of any extents they collapse to this).  Byte offsets into a u32[N] array.
    uint32_t hc;         // 1: hc_tab is built and the on-chip sweeps use it (all bonds non-zero iff dense_nn == nn)
    int32_t hc_d0[6];    // regular offset of each slot (x 4)
    int32_t hc_dd[6];    // (wrapped - regular) offset of each slot (x 4)
    // random-stream workgroups: waves per workgroup (4 or 8; all of them share one jump table) and the reciprocals
    // ceil(2^32 / x) with which the kernels divide small indices (device_util.hpp div_small)
    uint32_t rng_waves;
    uint32_t inv_temps, inv_replicas, inv_W;
    // 1: four to six forward directions (bcc, fcc, ...) with unit couplings: Metropolis sweeps take the integer table
    // route of the reference (sweep.rs:109-185 applies to any z') -- acceptance levels in three bit planes from the
    // random-stream kernel, bit-sliced neighbour counts in k_sweep_wide -- while the handle stays a `general` one for
    // everything else (heat bath, energies kept as f32 bits)
    uint32_t wide;
    // > 0 (with `wide`): the lattice's dependency levels are far thinner than a wave -- chunks of <= 16 sites of one
    // level, swept by one wave with a site's bonds spread over a quad of lanes (k_sweep_wide, rec_thin)
    uint32_t thin_chunks;
    uint32_t thin_dense; // 1: every bond of every site of every realisation is non-zero (no masks, h = nn everywhere)
    // Register-resident sweep (k_sweep_reg, kernels_reg.hip): 16 x 16 x 16, three directions, every bond non-zero.
    // One wave holds a tile word's 4096 sites in 64 VGPRs per lane; no LDS, no barrier.
    uint32_t reg_ok;      // 1: the handle's shape admits it (set once by pa_sim_create; PEAPODS_NO_REG=1 clears it)
    uint32_t planes_perm; // per launch: the random-stream workgroups file a site's acceptance planes at reg_pos[site]
                          //   (the order k_sweep_reg reads them in) instead of at the site
};

// hc_tab entry of a site (schedule order): site | wrap flags << 14 | coupling sign bits << 20
constexpr uint32_t kHcSiteBits = 14, kHcFlagShift = 14, kHcNegShift = 20, kHcNzShift = 26; // 14 + 6 + 6 + 6 bits

struct DevPtrs {
    uint32_t *S;             // [n_real][W][N] current spins
    uint32_t *S_next;        // [n_real][W][N] k_sweep writes here (double buffer; = S on the large path)
    uint2 *planes;           // [n_real][W][N]  .x/.y = bit 0/1 of the acceptance level
    uint32_t *planes_hi;     // [n_real][W][N]  bit 2 of the acceptance level (Gibbs sweeps; allocated on first use)
    uint32_t *cursor;        // [n_real][8][SP]  xoshiro state of the system in each slot
    uint32_t *cursor_out;    // same shape: where a random-stream launch leaves the advanced states (its workgroups all start
                             // from `cursor`, so the one that finishes a system's stream must not overwrite it); the host
                             // exchanges the two after the launch
    uint32_t *system_ids;    // [n_real][n_sys]  slot -> system
    uint32_t *slot_of;       // [n_real][n_sys]  system -> slot
    uint32_t *system_ids_out; // same shapes: where a RELAYED tempering event (PtParams::relay: the measurement block
    uint32_t *slot_of_out;    //   rides in the next sweep's random-stream launch) leaves the new permutation -- the
                             //   random-stream workgroups of that launch still read the old one; the host exchanges
                             //   the pairs after the launch
    const uint32_t *thr;     // [2][2][3][n_temps] thresholds: even set (dE=-2,-4,-6), odd set (-1,-3,-5);
                             //   second copy shifted left by 8 (valid when thr_preshifted)
    const uint32_t *thr_gibbs; // [2][7][n_temps] Gibbs thresholds, falling: even set dE = 2z', 2z'-2, ..;
                             //   odd set dE = 2z'-1, 2z'-3, ..; unused entries 0
    const uint32_t *mpar;    // [n_real][MW]     parity of the non-zero bond count per site
    const uint4 *rec;        // [n_real][N + 1] schedule order (+ dummy), 16-bit fields holding 4 * index
                             //   (large lattices: two uint4 per site = site, f0,b0,f1,b1,f2,b2, jmeta):
                             //   site,f0 | b0,f1 | b1,f2 | b2,jmeta
                             //   jmeta = neg[0..5] | nz[6..11] | h[12..13] for bonds f0,b0,f1,b1,f2,b2
    const uint32_t *pass_start; // [n_passes + 1]
    const uint32_t *hc_tab;  // [n_real][N] schedule order: site | wrap flags << 14 | neg bits << 20 (SimDims::hc)
    const uint4 *rec_wide;   // [n_real][2 (N + 1)] SimDims::wide, schedule order (+ dummy): thirteen 16-bit LDS byte
                             //   offsets (site, f0, b0, ..., f5, b5), h = ceil(m / 2), then neg[0..11] | nz[12..23]
    int32_t *Esum_pre, *Msum_pre;     // [n_real][n_sys] the sums a sweep left, copied by the Houdayer seed launch for the measurement
                                      //   blocks that ride in the growth launch (which replaces Esum / Msum)
    int32_t *Esum_relay, *Msum_relay; // [n_real][n_sys] sums exchanged by a relayed tempering event, until k_adopt_relay_sums
    uint32_t *relay_swapped;          // [n_real] 1: that event accepted a swap in this realisation
    const uint4 *rec_thin;   // [n_real][thin_chunks][64] SimDims::thin_chunks: lane 4 q + j of a chunk = part j of its q-th site:
                             //   x = site | nb[3j] << 16, y = nb[3j+1] | nb[3j+2] << 16 (LDS byte offsets),
                             //   z = neg[0..2] | nz[3..5] | h << 8 of those three slots (idle: the spare slot, no bond)
    const uint32_t *thr_wide; // [2][7][n_temps] SimDims::wide: Metropolis thresholds in the layout of thr_gibbs --
                             //   even set dE = -2, -4, ..., odd set dE = -1, -3, ... (falling; unused entries 0)
    const uint32_t *seg;     // [n_levels][pipe_K + 1] schedule positions where each slab's part of a level starts
    uint32_t *progress;      // [n_real * W][pipe_K] levels completed by each slab workgroup (epoch-based, monotone)
    const uint32_t *halo_start; // [pipe_K][n_levels + 2] k_sweep_slab: the exchange records a slab first reads at level l are
                             //   entries start[l] .. start[l + 1] of its region of xchg (ordered by that level)
    const uint2 *slab_seg;   // [n_levels][pipe_K] (first padded schedule position, sites) of a slab at a level;
                             //   positions are padded so that every segment starts at a multiple of 4
    const uint32_t *pos_of;  // [N] site -> padded schedule position (k_rng files the acceptance planes there)
    const uint32_t *reg_pos; // [N] site -> (level mod 16, z mod 4, lane) slot of k_sweep_reg (SimDims::planes_perm)
    const uint32_t *reg_cpl; // [n_real][4][3][64] k_sweep_reg: per lane, coupling sign bits of its 64 sites: word (m, dir),
                             //   bit 2 j = forward bond, bit 2 j + 1 = backward bond of the site in register (m, j)
    uint32_t *slab_error;    // [1] set when a slab gave up waiting for an exchange record
    unsigned long long *slab_stamps; // diagnostic (pa_debug_slab_stamps), normally null: [workgroups][n_levels + 4] s_memrealtime per level
    const uint4 *recS;       // [n_real][planes_stride] by padded position: 8 x u16 = LDS word of the site and of
                             //   f0,b0,f1,b1,f2,b2 (for a new value of another slab: the value word of its staged exchange record), then jmeta
    const uint32_t *xportS;  // [planes_stride] by padded position: 1 + exchange record the site's new value goes to, or 0
    uint4 *xchg;             // [n_real * W][pipe_K][2 * plane] exchange records {value, tag, value ^ tag, ~tag} in the
                             //   order of the reading slab's staged entries; tag = epoch base + level + 1
    const uint32_t *jump;    // [B][64][2][16][4] nibble tables of T^(b*BS) (jump_state_lds)
    int32_t *Esum;           // [n_real][n_sys] by slot: sum_{i,d} J s s (exact integer)
    int32_t *Msum;           // [n_real][n_sys] by slot: sum_i s_i
    const int32_t *nz_bonds; // [n_real] number of non-zero forward bonds
    int32_t *dot_spin;       // [n_real][n_pairs][n_temps]
    int32_t *dot_link;       // [n_real][n_pairs][n_temps]
    int32_t *dot_spin_prev;  // the pair the latest overlap launch wrote: the host exchanges dot_* and dot_*_prev after
    int32_t *dot_link_prev;  //   every overlap launch, measurement blocks read *_prev (an overlap block of their own
                             //   launch may already be writing dot_* for the next sweep)
    double *mom;             // [n_real][5][n_temps] running f64 sums
    double *ovm;             // [n_real][6][n_temps]
    uint64_t *hist;          // [n_real][n_temps][N + 1]
    double *qlq;             // [n_real][n_temps][N + 1]
    double *ql2q;            // [n_real][n_temps][N + 1]
    uint32_t *hist_touched;  // [n_real][n_temps][ceil((N + 1) / 32)] bins of hist / qlq / ql2q written since the
                             //   accumulators were last cleared: the three big arrays are never memset (302 MB at
                             //   the bench shape); a bin's first value of a run is stored, later ones are added,
                             //   and the getter reads untouched bins as zero
    uint64_t *edge_att;      // [n_real][n_edges]
    uint64_t *edge_acc;      // [n_real][n_edges]
    uint64_t *round_trips;   // [n_real][n_sys] by system
    uint8_t *trip_state;     // [n_real][n_sys] by system
    const float *ln_table;   // [2^24] ln(k * 2^-24) as computed by the host libm
    const float *dbeta;      // [n_edges] 1/T[t] - 1/T[t+1] (f32)
    uint32_t *swap_masks;    // [n_real][2][W] accepted tempering swaps of the last event (deferred exchange)
    uint32_t *swap_epoch;    // [n_real] bumped by every event that accepted a swap
    uint32_t *applied_epoch; // [n_real][W] epoch whose swaps are already in S
    uint32_t *big_counts;     // [n_real][n_sys][4] u32 scratch counters of the large-lattice kernels
    uint32_t *big_ticket;     // [n_real] workgroups of k_energy_overlap_big done with the realisation (the last one finalises)
    const uint16_t *nbr_site; // [N][12] neighbours by site: f0, b0, f1, b1, ... f5, b5 (self when unused)
    const uint32_t *sw_wrap;  // SimDims::hc: [3][MW] bitmaps over the sites -- the forward neighbour f0 / f1 / f2 lies through the wrap
    const uint32_t *sw_cpl;   // SimDims::hc: [n_real][6][MW] bitmaps -- forward coupling negative (3 directions), then non-zero (3)
    const uint32_t *jsite;    // [n_real][N] by site: neg[0..11] | nz[12..23] for bonds f0,b0,...,f5,b5
    const uint4 *fwd_big;     // [N] large lattices: forward neighbours f0, f1, f2 of every site (self when unused), by site
    const uint32_t *nbr_wide; // [N][16] in schedule order: site, f0, b0, ..., f5, b5, 0, 0, 0 (general path)
    const float *jorg_p;      // [n_temps] 1 - exp(-4 / T) by the host libm (Joerg bond probability)
    uint16_t *dfs_stack;      // [n_real][n_temps * n_pairs][N] scratch of k_jorg
    uint32_t *hd_scratch;     // [n_real][2][W][N] k_houdayer -> k_houdayer_grow: active masks, seeded clusters
    uint64_t *pair_rng;      // [n_real][n_temps * n_pairs][4] xoshiro state of the overlap-move generators
    // "sw" inner variant of the overlap-cluster moves (kernels_cluster.hip; built on first use)
    uint32_t *pair_slots;    // [n_real][n_temps][n_replicas] slots in the shuffled order of each temperature
    uint32_t *slot_bits;     // [n_real][SP][MW] spins of every slot as a bitmap over the sites (k_slot_bitmaps, per call)
    uint32_t *sw_scratch;    // [tasks][2][N] large lattices: the union-find's per-site arrays of every k_cluster_sw
                             //   workgroup (uncached device memory; tasks = n_real * max(n_temps * n_pairs, n_sys))
    const uint32_t *jump_unit; // [cluster_stream_units + 64][64][2][16][4] nibble tables of T^(64 u), then of T^0 .. T^63
    uint8_t *stream_bits;    // [n_real][n_sys][cluster_stream_units * 64 bytes] the cluster generators' draws, classified (unit
                             //   couplings): per task four bit planes of 2 words per unit (k_pair_stream), in half of the room
    uint32_t *stream_raw;    // same shape: the 24-bit draws themselves (other couplings)
    uint32_t *stream_error;  // [1] set if a task ever needed draws beyond its stream (checked by pa_sim_synchronize)
    const uint32_t *jorg_thr;// [n_temps] ceil((1 - exp(-4/T)) 2^24): draw < thr <=> gen::<f32>() < p
    const uint32_t *cmr_thr; // [n_temps] CMR blue bonds: 1 - r^2, r = exp(-2/T) (red bonds use fk_thr = 1 - r)
    const uint32_t *fk_thr;  // [n_temps] the same for the FK bond probability 1 - exp(-2/T)
    float *ac_m2, *ac_q2;    // [n_real][ac_stride][n_temps] per-sweep series of <m^2> and <q^2> (f32, as pushed)
    double *eq_sum;          // [n_real][2][n_temps] running sums of the equilibration diagnostic
    double *eq_ckpt;         // [n_real][64][2][n_temps] checkpoint averages
    // cluster statistics / graph observations / snapshots (kernels_cluster.hip; built on first use)
    uint64_t *cs_fk_csd;     // [n_real][n_temps][N + 1] FK cluster-size histogram
    uint64_t *cs_ov_csd;     // [n_real][cs_modes][n_temps][N + 1] overlap-cluster size histograms per build mode
    uint32_t *cs_top4_raw;   // [n_real][n_temps * n_pairs][4] four largest clusters of the current call
    double *cs_top4_acc;     // [n_real][cs_modes][n_temps][4] running sums of top4 / N (pair order)
    uint64_t *obs_acc;       // [n_real][4 kinds][n_temps][kObsWords + N + 1] observation accumulators:
                             //   count, top4[4], bonds, winding[4] (x, y, either, both), large, then sizes[N + 1]
    uint32_t *snap_bonds;    // [n_temps][2][6][MW] realisation 0, pair 0: accepted bonds (blue / only; red extension)
    uint32_t *snap_ids;      // [n_temps][N] cluster ids in the reference's union-find convention (k_uf_replay)
    uint32_t *snap_blue;     // [n_temps][N] CMR blue-cluster ids
    int8_t *snap_spins;      // [n_temps][2][N] spins of the pair before the move
    uint32_t *snap_sids;     // [n_temps][4] systems of the pair, then 1 if the ids are to be rebuilt, then unused
    // general-coupling path only
    const float *Jf;         // [n_real][N][nn] forward couplings as given
    const float *temps;      // [n_temps]
    float *thr_f;            // [n_real][W][N][32] thresholds (T/2) ln(..) of the current sweep
    uint32_t *xw;            // [n_real][W][N][4] bond sign words S_i ^ S_fwd(i,d)
};

// What a cluster call reports besides moving spins (simulation/mod.rs:217-335).
enum : uint32_t {
    kAuxSizes = 1,    // cluster-size histogram (+ four largest clusters for the overlap moves)
    kAuxObserve = 2,  // ClusterAction::Observe: the bonds are drawn, then nothing moves and no coin is drawn
    kAuxSnapshot = 4, // realisation 0, pair 0 of every temperature: spins, systems and bonds exported
    kAuxMetrics = 8,  // graph metrics of a recorded observation into obs_acc (with kAuxSizes)
};
constexpr uint32_t kObsWords = 11;
struct ClusterAux {
    uint32_t flags;
    uint32_t mode_idx; // row of cs_ov_csd / cs_top4_acc (round-robin index of the build mode)
    uint32_t n_modes;  // rows allocated
    uint32_t kind;     // obs_acc row: 0 fk, 1 houdayer, 2 jorg, 3 cmr_blue
    uint32_t square;   // 1: canonical 2D hypercubic lattice -> winding is measured (lattice.rs:31-38)
    uint32_t stream_units; // units of 64 draws k_pair_stream made for this call (set by the launcher)
};

// Room in a cluster generator's pre-drawn stream, in units of 64 draws: every forward bond, two draws per
// site (CMR: blue and grey clusters) and the seed draw of a single-cluster form.
__host__ __device__ inline uint32_t cluster_stream_units(const SimDims &d) { return (d.N * d.nn + 2u * d.N + 64u + 63u) / 64u + 1u; }

// k_sweep_slab: ring depths and the LDS word where the staged exchange records begin (the host writes the
// word of a staged neighbour's value into the site records)
constexpr uint32_t kSlabDepth = 6; // levels of records / planes / export targets resident in the ring
constexpr uint32_t kSlabAhead = 3; // exchange records are fetched this many levels ahead of their use (5: slower)
constexpr uint32_t kSlabStage = kSlabAhead + 1; // levels of staged exchange records, 128 records of 4 words each
__host__ __device__ inline uint32_t slab_stage_word(uint32_t own_sites, uint32_t plane_sites) {
    return own_sites + 2 * plane_sites + kSlabDepth * 256 * 7;
}

struct PtParams {
    int32_t schedule;     // PA_PT_*
    uint32_t first_parity;
    uint32_t hot_slot, cold_slot;
    // equilibration diagnostic (statistics/equilibration.rs): accumulate this sweep's replica-averaged
    // energy and pair-averaged link overlap; equil_ckpt = 1 + checkpoint index to snapshot, 0 = none
    uint32_t equil, equil_ckpt, equil_count;
    // autocorrelation series (autocorrelation.rs): ac_index = 1 + index of this recorded sweep in the
    // per-realisation history of ac_stride entries, 0 = not collected
    uint32_t ac_index, ac_stride;
    // bit 0: the moment sums start from zero with this launch; bit 1: the overlap sums and the P(q) bitmap do
    // (the accumulators are never memset: the first accumulating launch of a run stores instead of adding)
    uint32_t fresh;
    // 1: this event runs inside the random-stream launch of the NEXT sweep (k_rng_overlap): its workgroups replay
    // the swap decisions themselves and fetch their generators from the slots the systems come from, so the event
    // neither permutes cursors nor rewrites energies, and it files the permutation in system_ids_out / slot_of_out
    uint32_t relay;
    // the recorded sweep whose OVERLAP this launch accumulates (1 + index, 0 = none): the same sweep as ac_index on the
    // launch-per-phase protocol, the one before it where the overlap is measured inside the next sweep's launch
    uint32_t ac_index_q;
};

// All launchers enqueue on `stream` and return the hipGetLastError() of the launch.
// k_sweep_reg (kernels_reg.hip): the register-resident Metropolis sweep of 16^3 lattices (SimDims::reg_ok)
uint32_t reg_slot_of_site(uint32_t site);
hipError_t launch_sweep_reg(const SimDims &d, const DevPtrs &p, bool with_energy, hipStream_t stream,
                            unsigned long long *stamps = nullptr); // stamps: diagnostic, [workgroups][8] s_memrealtime
inline bool reg_sweep_applies(const SimDims &d, bool gibbs) { return d.reg_ok && !gibbs; }
hipError_t launch_rng(const SimDims &d, const DevPtrs &p, bool gibbs, hipStream_t stream);
hipError_t launch_sweep(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                        bool gibbs, hipStream_t stream);
hipError_t launch_overlap(const SimDims &d, const DevPtrs &p, hipStream_t stream);
hipError_t launch_rng_overlap(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool gibbs = false);
// The random stream of a sweep with the previous sweep's overlap measurement (with_overlap) and / or its deferred
// measurement block (statistics + tempering event, pt.relay = 1) riding in the same launch.
// experiment (PEAPODS_FLAGGED): stream + sweep of one Metropolis sweep in one launch, sweep workgroups released by counters
bool flagged_sweep_applies(const SimDims &d);
hipError_t launch_stream_sweep_flagged(const SimDims &d, const DevPtrs &p, uint32_t *flags, uint32_t epoch, uint32_t *timed_out,
                                       hipStream_t stream);
hipError_t launch_rng_relay(const SimDims &d, const DevPtrs &p, bool with_overlap, const PtParams &pt, bool do_accum,
                            bool accum_overlap, bool do_pt, hipStream_t stream, bool gibbs = false);
bool rng_relay_supported(const SimDims &d);
bool rng_relay_ladder_supported(const SimDims &d);
hipError_t launch_adopt_relay_sums(const SimDims &d, const DevPtrs &p, hipStream_t stream);
hipError_t launch_measure_overlap(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, bool accum_overlap,
                                  bool do_pt, hipStream_t stream);
hipError_t launch_sweep_pipe(const SimDims &d, const DevPtrs &p, bool gibbs, uint32_t epoch, hipStream_t stream,
                             bool apply_swaps = false); // apply_swaps: honoured by the LDS-resident kernel (SimDims::pipe_lds) only
hipError_t launch_energy_overlap_big(const SimDims &d, const DevPtrs &p, hipStream_t stream); // (d.big, d.nn <= 3)
hipError_t launch_apply_swaps(const SimDims &d, const DevPtrs &p, hipStream_t stream);
hipError_t launch_houdayer(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_energy = false);
hipError_t launch_houdayer_seeds(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_overlap);
hipError_t launch_houdayer_grow(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_energy);
hipError_t launch_houdayer_grow_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, hipStream_t stream,
                                        bool with_energy);
bool houdayer_computes_energy(const SimDims &d); // with_energy is honoured: no energy-only sweep launch needed after the move
size_t houdayer_lds_bytes(const SimDims &d);
hipError_t launch_jorg(const SimDims &d, const DevPtrs &p, hipStream_t stream);
size_t jorg_lds_bytes(const SimDims &d);
hipError_t launch_sweep_stamped(const SimDims &d, const DevPtrs &p, unsigned long long *stamps,
                                hipStream_t stream);
hipError_t launch_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_overlap,
                          bool do_accum, bool accum_overlap, bool do_pt, hipStream_t stream);
hipError_t launch_measure_stamped(const SimDims &d, const DevPtrs &p, const PtParams &pt,
                                  unsigned long long *stamps, hipStream_t stream);
size_t sweep_lds_bytes(const SimDims &d, bool gibbs);
size_t sweep_wide_lds_bytes(const SimDims &d);
bool takes_wide_route(const SimDims &d, bool gibbs); // Metropolis sweep of a SimDims::wide handle
size_t sweep_hc_lds_bytes(const SimDims &d, bool gibbs = false);
size_t sweep_slab_lds_bytes(const SimDims &d);
size_t sweep_slab_lds_bytes_for(uint32_t own_sites, uint32_t plane_sites, uint32_t n_levels);
// kernels_cluster.hip
// (paired: the shuffles of build_tasks -- and the per-slot bitmaps -- have been made by rider blocks of the overlap launch,
// k_overlap_pairing)
hipError_t launch_cluster_sw(const SimDims &d, const DevPtrs &p, int oc_mode, uint32_t group, bool wolff,
                             const ClusterAux &aux, hipStream_t stream, bool paired = false);
hipError_t launch_top4_accumulate(const SimDims &d, const DevPtrs &p, const ClusterAux &aux, hipStream_t stream);
hipError_t launch_uf_replay(const SimDims &d, const DevPtrs &p, bool cmr, hipStream_t stream);
hipError_t launch_cmr_wolff(const SimDims &d, const DevPtrs &p, hipStream_t stream);
hipError_t launch_jorg_wave(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool paired = false);
hipError_t launch_cmr_wave(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool paired = false);
// kernels.hip: riders of the overlap moves (k_overlap_pairing, k_energy_measure)
bool move_riders_supported(const SimDims &d);
hipError_t launch_overlap_pairing(const SimDims &d, const DevPtrs &p, hipStream_t stream, bool with_bitmaps);
hipError_t launch_energy_measure(const SimDims &d, const DevPtrs &p, const PtParams &pt, bool do_accum, bool accum_overlap,
                                 hipStream_t stream);
size_t wolff_wave_lds_bytes(const SimDims &d);
size_t cmr_wolff_lds_bytes(const SimDims &d);
size_t cluster_sw_lds_bytes(const SimDims &d);
size_t cluster_sw_lds_bytes_global_labels(const SimDims &d); // large lattices: per-site arrays in device memory
hipError_t launch_fk(const SimDims &d, const DevPtrs &p, bool wolff, const ClusterAux &aux, hipStream_t stream);
size_t fk_wolff_lds_bytes(const SimDims &d);
hipError_t configure_cluster_kernels();
hipError_t launch_exp_selftest(float *out, hipStream_t stream);
// kernels_general.hip
hipError_t launch_rng_f(const SimDims &d, const DevPtrs &p, bool gibbs, hipStream_t stream);
hipError_t launch_sweep_f(const SimDims &d, const DevPtrs &p, bool with_energy, bool do_sweep,
                          hipStream_t stream);
hipError_t launch_log_selftest(float *out_ln, float *out_logit, hipStream_t stream);
hipError_t configure_general_kernels();
size_t sweep_f_lds_bytes(const SimDims &d);
size_t measure_lds_bytes(const SimDims &d);
hipError_t configure_kernels(const SimDims &d);

} // namespace pa
