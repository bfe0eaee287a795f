// pa_sim: the device-resident ensemble and the C ABI (include/peapods_amd.h).
//
// Host orchestration only: builds the static tables (neighbour schedule,
// acceptance thresholds, jump tables), seeds the ensemble the way the
// reference does (src/lib.rs:157-164, realization.rs:155-211), and sequences
// the kernels of kernels.hip in the reference's per-sweep phase order
// (spin-sim/src/simulation/mod.rs:405-797).  No arithmetic of the hot path
// runs on the host after create()/reset().
#include "../../include/peapods_amd.h"
#include "engine.hpp"
#include "host_util.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace pa;

// ---- launch tracer (engine.hpp PA_LAUNCH): open on the thread that runs pa_sim_profile_step ----
namespace pa {
thread_local bool g_trace_on = false;
}
namespace {
struct TraceMark {
    const char *name;
    hipStream_t stream;
    hipEvent_t event;
};
thread_local std::vector<TraceMark> g_trace_marks;
thread_local hipError_t g_trace_error = hipSuccess;
} // namespace
void pa::trace_mark(const char *name, hipStream_t stream) {
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreate(&e);
    if (rc == hipSuccess) rc = hipEventRecord(e, stream);
    if (rc != hipSuccess && g_trace_error == hipSuccess) g_trace_error = rc;
    g_trace_marks.push_back({name, stream, e});
}


namespace {

struct DeviceBuffers {
    std::vector<void *> allocations;
    template <typename T> hipError_t alloc(T **ptr, size_t count) {
        void *raw = nullptr;
        hipError_t e = hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) return e;
        allocations.push_back(raw);
        *ptr = static_cast<T *>(raw);
        return hipSuccess;
    }
    void release() {
        for (void *a : allocations) (void)hipFree(a);
        allocations.clear();
    }
};

std::mutex g_ln_mutex;
std::map<int, float *> g_ln_tables; // per device, lives for the process

} // namespace

struct pa_sim {
    int device = 0;
    hipStream_t stream = nullptr;
    LatticeTables lat;
    SimDims d{};
    DevPtrs p{};
    DeviceBuffers buffers;
    std::vector<float> temps;
    uint64_t n_replicas = 0, n_temps = 0, n_real = 0;
    uint64_t constructor_seed = 0;
    uint64_t first_realization = 0;
    uint32_t hot_slot = 0, cold_slot = 0;
    uint32_t next_parity = 0;
    // non-const aliases of the static tables (filled once in create)
    uint32_t *thr = nullptr, *mpar = nullptr, *pass_start = nullptr, *jump = nullptr;
    uint4 *rec = nullptr, *rec_wide = nullptr, *rec_thin = nullptr;
    uint32_t *hc_tab = nullptr, *thr_wide = nullptr;
    uint32_t *reg_pos = nullptr, *reg_cpl = nullptr; // k_sweep_reg tables (SimDims::reg_ok)
    uint16_t *nbr_site = nullptr;
    uint32_t *jsite = nullptr, *nbr_wide = nullptr;
    uint32_t *sw_wrap = nullptr, *sw_cpl = nullptr; // bitmaps of the word-parallel cluster phases (SimDims::hc)
    uint32_t *flag_words = nullptr, flag_groups = 0, flag_epoch = 0; // experiment PEAPODS_FLAGGED (sweep_only)
    unsigned long long *cluster_stamps = nullptr; // pa_debug_cluster_stamps: [tasks][16] phase stamps of the last k_cluster_sw launch
    uint4 *fwd_big = nullptr;
    float *jorg_p = nullptr;
    uint32_t *thr_gibbs = nullptr; // built by the first Gibbs sample (ensure_gibbs)
    uint32_t *seg = nullptr;       // slab segments of the pipelined large-lattice sweep
    uint32_t *halo_start = nullptr; // k_sweep_slab tables (halo_sites_host: which site each staged entry is, host only)
    uint2 *slab_seg = nullptr;
    uint32_t *pos_of = nullptr;
    uint4 *recS = nullptr;
    uint32_t *xportS = nullptr;
    std::vector<uint32_t> halo_sites_host, halo_start_host;
    uint32_t pipe_epoch = 1;       // bumped by every k_sweep_pipe launch (progress counters are monotone)
    std::vector<uint64_t> equil_sweeps; // checkpoints of the last sample() with the equilibration diagnostic
    uint64_t ac_lag = 0, ac_samples = 0, ac_capacity = 0; // autocorrelation series of the last sample()
    bool ac_fft = false;                                  // evaluate them with the FFT backend
    float *Jf = nullptr, *temps_dev = nullptr; // general-coupling path
    uint32_t *jump_unit = nullptr;
    uint32_t *jorg_thr = nullptr, *fk_thr = nullptr, *cmr_thr = nullptr; // cluster kernels
    int32_t *nz_bonds = nullptr;
    float *dbeta = nullptr;
    // cluster statistics / observations / snapshots of the last sample()
    bool square = false;              // canonical 2D hypercubic lattice (lattice.rs:31-38): winding is defined
    uint32_t cs_modes = 0;            // rows allocated in cs_ov_csd / cs_top4_acc
    bool cs_ready = false, snap_ready = false;
    bool cs_fk = false, cs_overlap = false; // what the last sample() collected
    bool obs_kind[4] = {false, false, false, false};
    uint32_t cs_n_modes = 0;
    uint64_t top4_calls[4] = {0, 0, 0, 0};
    struct Snapshot {
        uint64_t sweep_id;
        int32_t mode_idx;
        bool has_blue;
        std::vector<uint32_t> ids, blue;
        std::vector<int8_t> spins;
        std::vector<uint64_t> sids;
    };
    std::vector<Snapshot> snapshots;
    std::vector<uint32_t> snap_ids_host, snap_blue_host; // ids persist from call to call (overlap.rs:246-248)
    // bookkeeping of the last sample()
    uint64_t n_recorded = 0;
    uint64_t n_sweeps_run = 0; // sweeps the last sample() enqueued (= n_sweeps unless it was interrupted)
    bool overlap_collected = false;
    bool swaps_pending = false; // a tempering event ran after the last sweep (deferred bit-column swaps)
    std::atomic<bool> interrupt{false};
    std::string error;
    // Realisations are independent, so the sweep pipeline runs as several interleaved groups of
    // realisations on separate streams: the VALU-bound random-stream kernel of one group overlaps
    // the latency-bound sweep / measurement / tempering kernels of another.
    struct Group {
        SimDims d;
        DevPtrs p;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        uint32_t fresh = 3; // see pa_sim::fresh
    };
    std::vector<Group> groups;
    uint32_t stream_groups_wanted = 0; // pa_sim_desc.stream_groups (0: the default)
    // Two consecutive steady-state sweeps of the plain protocol captured as one hipGraph per stream group (run_sweeps):
    // the four dependent launches of a pair then cost the graph's 1.6 us per hop instead of the stream's 2.8 us
    // (build_probe/gaps.hip).  `graph_key` hashes every kernel argument of the pair; a mismatch recaptures.
    std::vector<void *> uncached_areas; // hipExtMallocWithFlags areas (cluster scratch of large lattices), freed with the handle
    std::vector<hipGraphExec_t> pair_graphs;
    uint64_t graph_key = 0;
    // Running sums are never memset between sample() calls: bit 0 (moments) / bit 1 (overlap moments, P(q)
    // bitmap) says that the next accumulating k_measure launch starts them from zero (PtParams::fresh).  One
    // copy per stream group (run_sweeps) and one for whole-ensemble launches (the standalone phase entries).
    uint32_t fresh = 3;
    hipEvent_t fork = nullptr;
    hipEvent_t batch_events[2] = {nullptr, nullptr}; // run-ahead bound of run_sweeps (cooperative cancel)
};

namespace {

int fail(pa_sim *sim, int code, const std::string &msg) {
    if (sim) sim->error = msg;
    return code;
}

#define PA_HIP(sim, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(sim, PA_ERR_DEVICE,                                                        \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                        \
    } while (0)

void write_err(char *err, size_t len, const std::string &msg) {
    if (err && len) std::snprintf(err, len, "%s", msg.c_str());
}

// ln(k * 2^-24) for every 24-bit k, with the host libm: the parallel-tempering
// criterion compares against ln(gen::<f32>()) (tempering.rs:88-91) and the
// device must see the host's logf bit for bit.  64 MiB, once per device.
int ensure_ln_table(pa_sim *sim) {
    std::lock_guard<std::mutex> lock(g_ln_mutex);
    auto it = g_ln_tables.find(sim->device);
    if (it == g_ln_tables.end()) {
        const std::vector<float> &table = uniform_log_table();
        float *dev = nullptr;
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), sizeof(float) << 24));
        PA_HIP(sim, hipMemcpy(dev, table.data(), sizeof(float) << 24, hipMemcpyHostToDevice));
        it = g_ln_tables.emplace(sim->device, dev).first;
    }
    sim->p.ln_table = it->second;
    return PA_OK;
}

// General-coupling sweeps evaluate ln on the device (kernels_general.hip restates glibc's logf).
// Before the first such sweep on a device, every one of the 2^24 possible draws is pushed through
// it and compared with the host libm, for both ln(u) and ln(u / (1 - u)).
std::map<int, bool> g_log_verified;

int ensure_device_log(pa_sim *sim) {
    std::lock_guard<std::mutex> lock(g_ln_mutex);
    auto it = g_log_verified.find(sim->device);
    if (it == g_log_verified.end()) {
        const size_t n = size_t(1) << 24;
        float *dev_ln = nullptr, *dev_logit = nullptr;
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev_ln), n * sizeof(float)));
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev_logit), n * sizeof(float)));
        std::vector<float> got_ln(n), got_logit(n);
        hipError_t e = launch_log_selftest(dev_ln, dev_logit, sim->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(got_ln.data(), dev_ln, n * sizeof(float), hipMemcpyDeviceToHost, sim->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(got_logit.data(), dev_logit, n * sizeof(float), hipMemcpyDeviceToHost, sim->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(sim->stream);
        (void)hipFree(dev_ln);
        (void)hipFree(dev_logit);
        if (e != hipSuccess) return fail(sim, PA_ERR_DEVICE, std::string("logarithm self-test: ") + hipGetErrorString(e));
        const bool same = std::memcmp(got_ln.data(), uniform_log_table().data(), n * sizeof(float)) == 0 &&
                          std::memcmp(got_logit.data(), uniform_logit_table(nullptr).data(), n * sizeof(float)) == 0;
        it = g_log_verified.emplace(sim->device, same).first;
    }
    if (!it->second)
        return fail(sim, PA_ERR_UNSUPPORTED,
                    "the device logarithm does not reproduce this host's logf on the 2^24 draw grid; "
                    "general-coupling sweeps would not match the reference");
    return PA_OK;
}

// Cluster moves with non-unit couplings evaluate exp on the device (device_util.hpp restates glibc's
// expf); before the first such move on a device 2^24 arguments in [-128, 0] are compared with the host.
std::map<int, bool> g_exp_verified;

int ensure_device_exp(pa_sim *sim) {
    std::lock_guard<std::mutex> lock(g_ln_mutex);
    auto it = g_exp_verified.find(sim->device);
    if (it == g_exp_verified.end()) {
        const size_t n = size_t(1) << 24;
        float *dev = nullptr;
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), n * sizeof(float)));
        std::vector<float> got(n), want(n);
        hipError_t e = launch_exp_selftest(dev, sim->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(got.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, sim->stream);
        const unsigned n_threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> workers;
        for (unsigned k = 0; k < n_threads; ++k)
            workers.emplace_back([&, k] {
                for (size_t v = n * k / n_threads; v < n * (k + 1) / n_threads; ++v)
                    want[v] = expf(-(static_cast<float>(v) * 0x1p-17f));
            });
        for (auto &t : workers) t.join();
        if (e == hipSuccess) e = hipStreamSynchronize(sim->stream);
        (void)hipFree(dev);
        if (e != hipSuccess) return fail(sim, PA_ERR_DEVICE, std::string("exponential self-test: ") + hipGetErrorString(e));
        it = g_exp_verified.emplace(sim->device, std::memcmp(got.data(), want.data(), n * sizeof(float)) == 0).first;
    }
    if (!it->second)
        return fail(sim, PA_ERR_UNSUPPORTED,
                    "the device exponential does not reproduce this host's expf; cluster moves with "
                    "non-unit couplings would not match the reference");
    return PA_OK;
}

template <typename T> int upload(pa_sim *sim, T *dst, const std::vector<T> &src) {
    if (src.empty()) return PA_OK;
    PA_HIP(sim, hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice,
                               sim->stream));
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    return PA_OK;
}

template <typename T> int download(pa_sim *sim, std::vector<T> &dst, const T *src, size_t count) {
    dst.resize(count);
    if (!count) return PA_OK;
    PA_HIP(sim, hipMemcpyAsync(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost,
                               sim->stream));
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    return PA_OK;
}

// Seeds every realisation exactly as Realization::new / reset do, packs the
// spins slot-major and uploads the dynamic state.
int seed_state(pa_sim *sim, uint64_t seed) {
    const SimDims &d = sim->d;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    std::vector<uint32_t> S(tile_words * d.n_real, 0u);
    std::vector<uint32_t> cursor(static_cast<size_t>(d.n_real) * 8 * d.SP, 0u);
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> workers;
    for (unsigned k = 0; k < n_threads; ++k)
        workers.emplace_back([&, k] {
            for (uint32_t real = k; real < d.n_real; real += n_threads) {
                const uint64_t base = realization_seed(seed, sim->first_realization + real); // src/lib.rs:160
                uint32_t *tile = S.data() + tile_words * real;
                for (uint32_t sys = 0; sys < d.n_sys; ++sys) {
                    Xoshiro rng = Xoshiro::from_seed_u64(child_seed(base, kSystemSeedDomain, sys));
                    const uint32_t word = sys >> 5, bit = 1u << (sys & 31u);
                    for (uint32_t site = 0; site < d.N; ++site) // gen::<f32>() < 0.5 -> -1
                        if (rng.next_draw24() < (1u << 23)) tile[static_cast<size_t>(word) * d.N + site] |= bit;
                    for (int w = 0; w < 4; ++w) {
                        cursor[(static_cast<size_t>(real) * 8 + 2 * w) * d.SP + sys] =
                            static_cast<uint32_t>(rng.s[w]);
                        cursor[(static_cast<size_t>(real) * 8 + 2 * w + 1) * d.SP + sys] =
                            static_cast<uint32_t>(rng.s[w] >> 32);
                    }
                }
            }
        });
    for (auto &t : workers) t.join();

    // pair generators of the overlap-cluster moves (realization.rs:186-194)
    const uint32_t n_pair_rngs = d.n_temps * d.n_pairs;
    std::vector<uint64_t> pair_rng(static_cast<size_t>(d.n_real) * n_pair_rngs * 4);
    for (uint32_t real = 0; real < d.n_real; ++real) {
        const uint64_t base = realization_seed(seed, sim->first_realization + real);
        for (uint32_t i = 0; i < n_pair_rngs; ++i) {
            const Xoshiro rng = Xoshiro::from_seed_u64(child_seed(base, kPairSeedDomain, i));
            std::memcpy(&pair_rng[(static_cast<size_t>(real) * n_pair_rngs + i) * 4], rng.s, 32);
        }
    }

    std::vector<uint32_t> ids(static_cast<size_t>(d.n_real) * d.n_sys);
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = static_cast<uint32_t>(i % d.n_sys);
    std::vector<uint8_t> trip(static_cast<size_t>(d.n_real) * d.n_sys, 0);
    for (uint32_t real = 0; real < d.n_real; ++real) // PtState::reset, realization.rs:46-71
        for (uint32_t rep = 0; rep < d.n_replicas; ++rep)
            trip[static_cast<size_t>(real) * d.n_sys + rep * d.n_temps + sim->hot_slot] = 1;

    int rc;
    if ((rc = upload(sim, sim->p.S, S))) return rc;
    if ((rc = upload(sim, sim->p.cursor, cursor))) return rc;
    if ((rc = upload(sim, sim->p.system_ids, ids))) return rc;
    if ((rc = upload(sim, sim->p.slot_of, ids))) return rc;
    if ((rc = upload(sim, sim->p.trip_state, trip))) return rc;
    if ((rc = upload(sim, sim->p.pair_rng, pair_rng))) return rc;
    PA_HIP(sim, hipMemsetAsync(sim->p.edge_att, 0, sizeof(uint64_t) * d.n_real * d.n_edges, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.edge_acc, 0, sizeof(uint64_t) * d.n_real * d.n_edges, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.round_trips, 0, sizeof(uint64_t) * d.n_real * d.n_sys, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.swap_epoch, 0, sizeof(uint32_t) * d.n_real, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.relay_swapped, 0, sizeof(uint32_t) * d.n_real, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.applied_epoch, 0, sizeof(uint32_t) * d.n_real * d.W, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.swap_masks, 0, sizeof(uint32_t) * d.n_real * 2 * d.W, sim->stream));
    sim->swaps_pending = false;
    sim->next_parity = 0;
    // Realization::new computes the initial energies (realization.rs:196-197)
    PA_HIP(sim, launch_sweep(sim->d, sim->p, true, false, false, sim->stream));
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    return PA_OK;
}

int zero_accumulators(pa_sim *sim) {
    // No device work: the first accumulating k_measure of the run stores instead of adding (moment sums) and
    // clears its rows of the P(q) bitmap; the three histogram arrays (302 MB at the bench shape) are
    // validated by that bitmap and never cleared.  A run that records nothing leaves the flags set and the
    // getters report zeros.
    sim->fresh = 3;
    for (auto &g : sim->groups) g.fresh = 3;
    return PA_OK;
}

// the "start from zero" bits of `fresh` a measurement with these flags consumes
uint32_t consume_fresh(uint32_t &fresh, const SimDims &d, bool do_accum, bool accum_overlap) {
    const uint32_t bits = (do_accum ? fresh & 1u : 0u) | ((accum_overlap && d.n_pairs) ? fresh & 2u : 0u);
    fresh &= ~bits;
    return bits;
}

// launch_measure with the "start from zero" bits of `fresh` this launch consumes
hipError_t measure_launch(uint32_t &fresh, const SimDims &d, const DevPtrs &p, PtParams pt, bool do_accum,
                          bool accum_overlap, bool do_pt, hipStream_t stream) {
    pt.fresh = consume_fresh(fresh, d, do_accum, accum_overlap);
    return launch_measure(d, p, pt, false, do_accum, accum_overlap, do_pt, stream);
}

// The dot products of an overlap launch go to dot_*; measurement blocks read dot_*_prev: exchange after the launch.
void toggle_dots(DevPtrs &p) {
    std::swap(p.dot_spin, p.dot_spin_prev);
    std::swap(p.dot_link, p.dot_link_prev);
}
hipError_t overlap_launch(const SimDims &d, DevPtrs &p, hipStream_t stream) {
    const hipError_t e = launch_overlap(d, p, stream);
    toggle_dots(p);
    return e;
}
// A relayed tempering event filed the new slot <-> system permutation in the second pair of tables.
void toggle_system_tables(pa_sim *sim) {
    std::swap(sim->p.system_ids, sim->p.system_ids_out);
    std::swap(sim->p.slot_of, sim->p.slot_of_out);
    for (auto &g : sim->groups) {
        std::swap(g.p.system_ids, g.p.system_ids_out);
        std::swap(g.p.slot_of, g.p.slot_of_out);
    }
}
// true while no launch since the last zero_accumulators has accumulated into (bit 0: moments, bit 1: overlap)
bool nothing_accumulated(const pa_sim *sim, uint32_t bit) {
    if (!(sim->fresh & bit)) return false;
    for (const auto &g : sim->groups)
        if (!(g.fresh & bit)) return false;
    return true;
}

// DevPtrs of the sub-ensemble starting at realisation r0 (every per-realisation array advanced).
DevPtrs offset_ptrs(const DevPtrs &p, const SimDims &d, uint32_t r0) {
    DevPtrs q = p;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    const size_t bins = static_cast<size_t>(d.n_temps) * (d.N + 1);
    q.S += tile_words * r0;
    q.S_next += tile_words * r0;
    q.swap_epoch += r0;
    q.applied_epoch += static_cast<size_t>(d.W) * r0;
    if (d.general) {
        q.Jf += static_cast<size_t>(d.N) * d.nn * r0;
        q.thr_f += tile_words * 32 * r0;
        q.xw += tile_words * (d.nn <= 3 ? 4 : 8) * r0;
    } else {
        q.planes += static_cast<size_t>(d.planes_stride) * d.W * r0;
    }
    if (d.wide) {
        q.planes += static_cast<size_t>(d.planes_stride) * d.W * r0;
        q.planes_hi += static_cast<size_t>(d.planes_stride) * d.W * r0;
        if (d.thin_chunks) q.rec_thin += static_cast<size_t>(d.thin_chunks) * 64 * r0;
    }
    if (q.rec_wide) q.rec_wide += (static_cast<size_t>(d.N) + 1) * 2 * r0;
    q.cursor += static_cast<size_t>(8) * d.SP * r0;
    q.cursor_out += static_cast<size_t>(8) * d.SP * r0;
    q.system_ids += static_cast<size_t>(d.n_sys) * r0;
    q.slot_of += static_cast<size_t>(d.n_sys) * r0;
    q.system_ids_out += static_cast<size_t>(d.n_sys) * r0;
    q.slot_of_out += static_cast<size_t>(d.n_sys) * r0;
    q.mpar += static_cast<size_t>(d.MW) * r0;
    q.rec += (static_cast<size_t>(d.N) + 1) * (d.big ? 2 : 1) * r0;
    if (d.hc) q.hc_tab += static_cast<size_t>(d.N) * r0;
    if (d.reg_ok) q.reg_cpl += static_cast<size_t>(12 * 64) * r0;
    if (d.pipe_lds) q.recS += static_cast<size_t>(d.planes_stride) * r0;
    if (d.pipe_K) q.progress += static_cast<size_t>(d.W) * d.pipe_K * r0;
    if (d.pipe_lds) q.xchg += static_cast<size_t>(d.W) * d.pipe_K * 2 * d.plane * r0;

    q.big_counts += static_cast<size_t>(4) * d.n_sys * r0;
    q.big_ticket += r0;
    q.Esum += static_cast<size_t>(d.n_sys) * r0;
    q.Msum += static_cast<size_t>(d.n_sys) * r0;
    q.Esum_pre += static_cast<size_t>(d.n_sys) * r0;
    q.Msum_pre += static_cast<size_t>(d.n_sys) * r0;
    q.Esum_relay += static_cast<size_t>(d.n_sys) * r0;
    q.Msum_relay += static_cast<size_t>(d.n_sys) * r0;
    q.relay_swapped += r0;
    q.nz_bonds += r0;
    q.dot_spin += static_cast<size_t>(d.n_pairs) * d.n_temps * r0;
    q.dot_link += static_cast<size_t>(d.n_pairs) * d.n_temps * r0;
    q.dot_spin_prev += static_cast<size_t>(d.n_pairs) * d.n_temps * r0;
    q.dot_link_prev += static_cast<size_t>(d.n_pairs) * d.n_temps * r0;
    q.mom += static_cast<size_t>(5) * d.n_temps * r0;
    q.ovm += static_cast<size_t>(6) * d.n_temps * r0;
    if (d.n_pairs) {
        q.hist += bins * r0;
        q.qlq += bins * r0;
        q.ql2q += bins * r0;
        q.hist_touched += static_cast<size_t>(d.n_temps) * ((d.N + 32) / 32) * r0;
    }
    q.edge_att += static_cast<size_t>(d.n_edges) * r0;
    q.edge_acc += static_cast<size_t>(d.n_edges) * r0;
    q.round_trips += static_cast<size_t>(d.n_sys) * r0;
    q.trip_state += static_cast<size_t>(d.n_sys) * r0;
    q.swap_masks += static_cast<size_t>(2) * d.W * r0;
    q.pair_rng += static_cast<size_t>(4) * d.n_temps * d.n_pairs * r0;
    q.jsite += static_cast<size_t>(d.N) * r0;
    if (q.sw_cpl) q.sw_cpl += static_cast<size_t>(6) * d.MW * r0;
    if (q.eq_sum) {
        q.eq_sum += static_cast<size_t>(2) * d.n_temps * r0;
        q.eq_ckpt += static_cast<size_t>(64) * 2 * d.n_temps * r0;
    }
    q.dfs_stack += static_cast<size_t>(std::max(d.n_temps * d.n_pairs, d.n_sys)) * d.N * r0;
    if (q.hd_scratch) q.hd_scratch += 2 * tile_words * r0;
    if (q.slot_bits) q.slot_bits += static_cast<size_t>(d.SP) * d.MW * r0;
    const size_t stream_per_real = static_cast<size_t>(d.n_sys) * cluster_stream_units(d) * 64;
    if (q.stream_bits) q.stream_bits += stream_per_real * r0;
    if (q.stream_raw) q.stream_raw += stream_per_real * r0;
    return q;
}

// the cluster-statistics arrays of the sub-ensemble starting at realisation r0
void offset_cluster_stats(DevPtrs &q, const DevPtrs &p, const SimDims &d, uint32_t modes, uint32_t r0) {
    const size_t bins = static_cast<size_t>(d.n_temps) * (d.N + 1);
    q.cs_fk_csd = p.cs_fk_csd + bins * r0;
    q.cs_ov_csd = p.cs_ov_csd + bins * modes * r0;
    q.cs_top4_raw = p.cs_top4_raw + static_cast<size_t>(4) * d.n_temps * std::max(d.n_pairs, 1u) * r0;
    q.cs_top4_acc = p.cs_top4_acc + static_cast<size_t>(4) * d.n_temps * modes * r0;
    q.obs_acc = p.obs_acc + static_cast<size_t>(4) * d.n_temps * (kObsWords + d.N + 1) * r0;
    q.snap_bonds = p.snap_bonds;
    q.snap_ids = p.snap_ids;
    q.snap_blue = p.snap_blue;
    q.snap_spins = p.snap_spins;
    q.snap_sids = p.snap_sids;
}

int make_groups(pa_sim *sim) {
    uint32_t n_groups = 2; // two interleaved groups overlap a little (5 %); three or more collapse
    if (sim->stream_groups_wanted >= 1 && sim->stream_groups_wanted <= 16) n_groups = sim->stream_groups_wanted;
    else if (const char *env = dev_env("PEAPODS_STREAMS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 16) n_groups = static_cast<uint32_t>(v);
    }
    n_groups = std::min<uint32_t>(n_groups, sim->d.n_real);
    PA_HIP(sim, hipEventCreateWithFlags(&sim->fork, hipEventDisableTiming));
    for (auto &e : sim->batch_events) PA_HIP(sim, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t r0 = sim->d.n_real * g / n_groups, r1 = sim->d.n_real * (g + 1) / n_groups;
        pa_sim::Group grp;
        grp.d = sim->d;
        grp.d.n_real = r1 - r0;
        grp.p = offset_ptrs(sim->p, sim->d, r0);
        PA_HIP(sim, hipStreamCreateWithFlags(&grp.stream, hipStreamNonBlocking));
        PA_HIP(sim, hipEventCreateWithFlags(&grp.done, hipEventDisableTiming));
        sim->groups.push_back(grp);
    }
    return PA_OK;
}

// k_sweep reads S and writes S_next, the random-stream launch reads cursor and writes cursor_out: exchange the
// roles everywhere after every sweep's launches.
void toggle_spin_buffers(pa_sim *sim) {
    // (the random-stream launch of the sweep left the advanced generator states in cursor_out)
    std::swap(sim->p.cursor, sim->p.cursor_out);
    for (auto &g : sim->groups) std::swap(g.p.cursor, g.p.cursor_out);
    if (sim->p.S_next == sim->p.S) return; // paths that sweep in place
    std::swap(sim->p.S, sim->p.S_next);
    for (auto &g : sim->groups) std::swap(g.p.S, g.p.S_next);
}

// Sweep of one group (and whatever has to precede it): on the large path pending tempering swaps
// are materialised first; the LDS kernel applies them itself while loading.
hipError_t enqueue_sweep(pa_sim *sim, const SimDims &d, const DevPtrs &p, bool with_energy,
                         bool gibbs, hipStream_t stream, bool overlap_too = false) {
    // (the LDS-resident slab kernel exchanges the deferred bit columns while it loads: two launches less per sweep)
    const bool at_load = d.pipe_K && d.pipe_lds && !dev_env("PEAPODS_NO_SWAPS_AT_LOAD");
    if ((d.big || d.general) && sim->swaps_pending && !at_load) {
        hipError_t e = launch_apply_swaps(d, p, stream);
        if (e != hipSuccess) return e;
    }
    if (d.pipe_K) { // pipelined large-lattice sweep S -> S_next, then the energies of the new state
        hipError_t e = launch_sweep_pipe(d, p, gibbs, sim->pipe_epoch++, stream, at_load && sim->swaps_pending);
        if (e != hipSuccess || !with_energy) return e;
        DevPtrs q = p;
        q.S = p.S_next;
        // overlap_too: the overlap of the new state in the same launch as its energies (the caller exchanges the dots)
        if (overlap_too) return launch_energy_overlap_big(d, q, stream);
        return launch_sweep(d, q, true, false, false, stream);
    }
    return launch_sweep(d, p, with_energy, true, gibbs, stream);
}

// Gibbs sweeps need seven thresholds per temperature and parity class and a third level plane;
// both are built by the first call that asks for them (sweep.rs:262-284).
int ensure_gibbs(pa_sim *sim) {
    if (sim->thr_gibbs || sim->d.general) return PA_OK; // the general path evaluates the rule per draw
    const SimDims &d = sim->d;
    if (!d.big && sweep_lds_bytes(d, true) > 160 * 1024)
        return fail(sim, PA_ERR_UNSUPPORTED, "lattice too large for the on-chip Gibbs sweep");
    std::vector<uint32_t> table;
    const float unit = 1.0f;
    if (!build_gibbs_table(&unit, 1, sim->temps.data(), d.n_temps, d.nn, table))
        return fail(sim, PA_ERR_UNSUPPORTED,
                    "the host libm's ln(u / (1 - u)) is not monotone on the 2^24 grid");
    const uint32_t width = 4 * d.nn + 1, off = 2 * d.nn;
    for (uint32_t t = 0; t < d.n_temps; ++t)
        for (uint32_t e = 1; e < width; ++e)
            if (table[t * width + e] < table[t * width + e - 1])
                return fail(sim, PA_ERR_UNSUPPORTED, "Gibbs acceptance table is not monotone in the energy change");
    std::vector<uint32_t> thr(14 * d.n_temps, 0u);
    for (uint32_t t = 0; t < d.n_temps; ++t)
        for (uint32_t j = 0; j < 7; ++j) {
            if (2 * j <= 2 * off) thr[(0 + j) * d.n_temps + t] = table[t * width + 2 * off - 2 * j];         // dE = 2z' - 2j
            if (2 * j + 1 <= 2 * off - 1) thr[(7 + j) * d.n_temps + t] = table[t * width + 2 * off - 1 - 2 * j]; // 2z' - 1 - 2j
        }
    PA_HIP(sim, sim->buffers.alloc(&sim->thr_gibbs, thr.size()));
    PA_HIP(sim, sim->buffers.alloc(&sim->p.planes_hi, static_cast<size_t>(d.planes_stride) * d.W * d.n_real));
    if (int rc = upload(sim, sim->thr_gibbs, thr)) return rc;
    sim->p.thr_gibbs = sim->thr_gibbs;
    uint32_t r0 = 0;
    for (auto &g : sim->groups) {
        g.p.thr_gibbs = sim->thr_gibbs;
        g.p.planes_hi = sim->p.planes_hi + static_cast<size_t>(d.planes_stride) * d.W * r0;
        r0 += g.d.n_real;
    }
    return PA_OK;
}

// Make S consistent for every reader other than k_sweep (getters, standalone phases, end of run).
int flush_swaps(pa_sim *sim) {
    if (!sim->swaps_pending) return PA_OK;
    PA_HIP(sim, launch_apply_swaps(sim->d, sim->p, sim->stream));
    sim->swaps_pending = false;
    return PA_OK;
}

// The union-find cluster moves read their generators' draws from streams made by jump-ahead in units
// of 64 draws (k_pair_stream): tables T^(64 u) for every unit a move can reach, the stream buffers,
// integer bond thresholds and the pairing scratch are built by the first call that asks for them.
int ensure_cluster_sw(pa_sim *sim) {
    if (sim->jump_unit) return PA_OK;
    const SimDims &d = sim->d;
    // A task's union-find normally lives in its workgroup's LDS.  Where that does not fit (the reference's own Rust
    // benchmark: 128 x 128 sites, examples/bench.rs:12) the two per-site arrays move to an uncached scratch area of
    // device memory, one slice per workgroup (k_cluster_sw<.., .., true>); the bitmaps stay on chip.
    const bool global_labels = cluster_sw_lds_bytes(d) > 160 * 1024;
    if (d.N > 65535 || (global_labels && cluster_sw_lds_bytes_global_labels(d) > 160 * 1024))
        return fail(sim, PA_ERR_UNSUPPORTED, "lattice too large for the cluster kernels (at most 65535 sites)");
    if (global_labels) {
        const size_t tasks = static_cast<size_t>(d.n_real) * std::max(d.n_temps * std::max(d.n_pairs, 1u), d.n_sys);
        void *area = nullptr;
        PA_HIP(sim, hipExtMallocWithFlags(&area, tasks * 2 * d.N * sizeof(uint32_t), hipDeviceMallocUncached));
        sim->uncached_areas.push_back(area);
        sim->p.sw_scratch = static_cast<uint32_t *>(area);
    }
    const uint32_t n_units = cluster_stream_units(d);
    // ... followed by the tables of T^0 .. T^63: a generator that ends k draws into a unit takes the unit's jump, then T^k
    std::vector<uint32_t> jump_unit(static_cast<size_t>(n_units + 64) * 64 * 16 * 8);
    {
        JumpMatrix step = JumpMatrix::steps(64);
        JumpMatrix acc = JumpMatrix::identity();
        for (uint32_t b = 0; b < n_units; ++b) {
            acc.nibble_table(&jump_unit[static_cast<size_t>(b) * 64 * 16 * 8]);
            if (b + 1 < n_units) acc = step.compose(acc);
        }
        step = JumpMatrix::steps(1);
        acc = JumpMatrix::identity();
        for (uint32_t k = 0; k < 64; ++k) {
            acc.nibble_table(&jump_unit[static_cast<size_t>(n_units + k) * 64 * 16 * 8]);
            if (k + 1 < 64) acc = step.compose(acc);
        }
    }
    // gen::<f32>() < p on the 2^24 grid as an integer threshold; p by the host's expf for a
    // satisfied unit bond: Joerg 1 - exp(-4/T) (overlap.rs:430), FK 1 - exp(-2/T) (fk.rs:74,119)
    // CMR (overlap.rs:601-603): blue bonds 1 - r * r with r = exp(-2|J|/T) in f32, red bonds 1 - r
    std::vector<uint32_t> thr(d.n_temps), thr_fk(d.n_temps), thr_cmr(d.n_temps);
    auto as_threshold = [](float bond_p) {
        const double scaled = std::ceil(static_cast<double>(bond_p) * 16777216.0);
        return scaled <= 0.0 ? 0u : (scaled >= 16777216.0 ? (1u << 24) : static_cast<uint32_t>(scaled));
    };
    for (uint32_t t = 0; t < d.n_temps; ++t) {
        thr[t] = as_threshold(1.0f - expf(-4.0f * 1.0f / sim->temps[t]));
        thr_fk[t] = as_threshold(1.0f - expf(-2.0f * 1.0f / sim->temps[t]));
        const float r = expf(-2.0f * 1.0f / sim->temps[t]);
        thr_cmr[t] = as_threshold(1.0f - r * r);
    }
    PA_HIP(sim, sim->buffers.alloc(&sim->jump_unit, jump_unit.size()));
    const size_t stream_per_real = static_cast<size_t>(d.n_sys) * n_units * 64;
    PA_HIP(sim, sim->buffers.alloc(&sim->p.stream_error, 1));
    PA_HIP(sim, hipMemsetAsync(sim->p.stream_error, 0, sizeof(uint32_t), sim->stream));
    if (d.unit)
        PA_HIP(sim, sim->buffers.alloc(&sim->p.stream_bits, stream_per_real * d.n_real));
    else
        PA_HIP(sim, sim->buffers.alloc(&sim->p.stream_raw, stream_per_real * d.n_real));
    PA_HIP(sim, sim->buffers.alloc(&sim->jorg_thr, thr.size()));
    PA_HIP(sim, sim->buffers.alloc(&sim->fk_thr, thr_fk.size()));
    PA_HIP(sim, sim->buffers.alloc(&sim->cmr_thr, thr_cmr.size()));
    const size_t slots_per_real = static_cast<size_t>(d.n_temps) * d.n_replicas;
    PA_HIP(sim, sim->buffers.alloc(&sim->p.pair_slots, slots_per_real * d.n_real));
    PA_HIP(sim, sim->buffers.alloc(&sim->p.slot_bits, static_cast<size_t>(d.n_real) * d.SP * d.MW));
    int rc;
    if ((rc = upload(sim, sim->jump_unit, jump_unit)) ||
        (rc = upload(sim, sim->jorg_thr, thr)) ||
        (rc = upload(sim, sim->fk_thr, thr_fk)) || (rc = upload(sim, sim->cmr_thr, thr_cmr)))
        return rc;
    if (configure_cluster_kernels() != hipSuccess) return fail(sim, PA_ERR_DEVICE, "kernel LDS configuration failed");
    sim->p.jump_unit = sim->jump_unit;
    sim->p.jorg_thr = sim->jorg_thr;
    sim->p.fk_thr = sim->fk_thr;
    sim->p.cmr_thr = sim->cmr_thr;
    uint32_t r0 = 0;
    for (auto &g : sim->groups) {
        g.p.jump_unit = sim->jump_unit;
        g.p.stream_error = sim->p.stream_error;
        g.p.stream_bits = sim->p.stream_bits ? sim->p.stream_bits + stream_per_real * r0 : nullptr;
        g.p.stream_raw = sim->p.stream_raw ? sim->p.stream_raw + stream_per_real * r0 : nullptr;
        g.p.jorg_thr = sim->jorg_thr;
        g.p.fk_thr = sim->fk_thr;
        g.p.cmr_thr = sim->cmr_thr;
        g.p.pair_slots = sim->p.pair_slots + slots_per_real * r0;
        g.p.sw_scratch = sim->p.sw_scratch
                             ? sim->p.sw_scratch + static_cast<size_t>(std::max(d.n_temps * std::max(d.n_pairs, 1u), d.n_sys)) * 2 * d.N * r0
                             : nullptr;
        g.p.slot_bits = sim->p.slot_bits + static_cast<size_t>(d.SP) * d.MW * r0;
        r0 += g.d.n_real;
    }
    return PA_OK;
}

// Buffers of the equilibration diagnostic, built by the first call that asks for it.
int ensure_equil(pa_sim *sim) {
    const SimDims &d = sim->d;
    if (!sim->p.eq_sum) {
        PA_HIP(sim, sim->buffers.alloc(&sim->p.eq_sum, static_cast<size_t>(d.n_real) * 2 * d.n_temps));
        PA_HIP(sim, sim->buffers.alloc(&sim->p.eq_ckpt, static_cast<size_t>(d.n_real) * 64 * 2 * d.n_temps));
        uint32_t r0 = 0;
        for (auto &g : sim->groups) {
            g.p.eq_sum = sim->p.eq_sum + static_cast<size_t>(2) * d.n_temps * r0;
            g.p.eq_ckpt = sim->p.eq_ckpt + static_cast<size_t>(64) * 2 * d.n_temps * r0;
            r0 += g.d.n_real;
        }
    }
    PA_HIP(sim, hipMemsetAsync(sim->p.eq_sum, 0, sizeof(double) * d.n_real * 2 * d.n_temps, sim->stream));
    return PA_OK;
}

// Per-sweep series of the autocorrelation observables: [n_real][capacity][n_temps] f32, grown on demand.
int ensure_autocorr(pa_sim *sim, uint64_t n_samples) {
    const SimDims &d = sim->d;
    if (n_samples > sim->ac_capacity) {
        const size_t count = static_cast<size_t>(d.n_real) * n_samples * d.n_temps;
        if (count * 2 * sizeof(float) > (size_t(8) << 30))
            return fail(sim, PA_ERR_UNSUPPORTED, "autocorrelation history above 8 GiB");
        // (earlier, smaller buffers stay with the handle until it is destroyed)
        PA_HIP(sim, sim->buffers.alloc(&sim->p.ac_m2, count));
        PA_HIP(sim, sim->buffers.alloc(&sim->p.ac_q2, count));
        sim->ac_capacity = n_samples;
    }
    uint32_t r0 = 0;
    for (auto &g : sim->groups) {
        g.p.ac_m2 = sim->p.ac_m2 + static_cast<size_t>(sim->ac_capacity) * d.n_temps * r0;
        g.p.ac_q2 = sim->p.ac_q2 + static_cast<size_t>(sim->ac_capacity) * d.n_temps * r0;
        r0 += g.d.n_real;
    }
    return PA_OK;
}

// Accumulators of the cluster statistics (ClusterStats, GraphObservationAccum, snapshots), allocated by
// the first sample() that asks for them and cleared by every such sample().
int ensure_cluster_stats(pa_sim *sim, uint32_t modes, bool snapshots) {
    const SimDims &d = sim->d;
    modes = std::max(modes, 1u);
    const size_t bins = static_cast<size_t>(d.n_real) * d.n_temps * (d.N + 1);
    const size_t pairs = std::max(d.n_pairs, 1u);
    if (!sim->cs_ready || modes > sim->cs_modes) {
        if (!sim->cs_ready) {
            PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.cs_fk_csd), sizeof(uint64_t) * bins));
            PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.cs_top4_raw),
                                  sizeof(uint32_t) * 4 * d.n_real * d.n_temps * pairs));
            PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.obs_acc),
                                  sizeof(uint64_t) * 4 * d.n_real * d.n_temps * (kObsWords + d.N + 1)));
        }
        // the new buffers first, then swap: a failed allocation leaves the old ones in place and valid
        uint64_t *csd = nullptr;
        double *top4 = nullptr;
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&csd), sizeof(uint64_t) * bins * modes));
        if (hipMalloc(reinterpret_cast<void **>(&top4), sizeof(double) * 4 * d.n_real * d.n_temps * modes) != hipSuccess) {
            (void)hipFree(csd);
            return fail(sim, PA_ERR_DEVICE, "hipMalloc: cluster statistics");
        }
        PA_HIP(sim, hipStreamSynchronize(sim->stream));
        if (sim->p.cs_ov_csd) (void)hipFree(sim->p.cs_ov_csd);
        if (sim->p.cs_top4_acc) (void)hipFree(sim->p.cs_top4_acc);
        sim->p.cs_ov_csd = csd;
        sim->p.cs_top4_acc = top4;
        sim->cs_modes = modes;
        sim->cs_ready = true;
    }
    if (snapshots && !sim->snap_ready) {
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.snap_bonds), sizeof(uint32_t) * d.n_temps * 2 * 6 * d.MW));
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.snap_ids), sizeof(uint32_t) * d.n_temps * d.N));
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.snap_blue), sizeof(uint32_t) * d.n_temps * d.N));
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.snap_spins), static_cast<size_t>(d.n_temps) * 2 * d.N));
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->p.snap_sids), sizeof(uint32_t) * d.n_temps * 4));
        sim->snap_ready = true;
    }
    PA_HIP(sim, hipMemsetAsync(sim->p.cs_fk_csd, 0, sizeof(uint64_t) * bins, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.cs_ov_csd, 0, sizeof(uint64_t) * bins * sim->cs_modes, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.cs_top4_acc, 0, sizeof(double) * 4 * d.n_real * d.n_temps * sim->cs_modes, sim->stream));
    PA_HIP(sim, hipMemsetAsync(sim->p.obs_acc, 0, sizeof(uint64_t) * 4 * d.n_real * d.n_temps * (kObsWords + d.N + 1),
                               sim->stream));
    if (snapshots) { // the buffers of a run start empty (mod.rs:311-334)
        PA_HIP(sim, hipMemsetAsync(sim->p.snap_ids, 0, sizeof(uint32_t) * d.n_temps * d.N, sim->stream));
        PA_HIP(sim, hipMemsetAsync(sim->p.snap_blue, 0, sizeof(uint32_t) * d.n_temps * d.N, sim->stream));
    }
    uint32_t r0 = 0;
    for (auto &g : sim->groups) {
        offset_cluster_stats(g.p, sim->p, d, sim->cs_modes, r0);
        r0 += g.d.n_real;
    }
    return PA_OK;
}

std::string validate_text(const pa_sample_config *cfg, uint64_t n_replicas, int *code) {
    *code = PA_ERR_INVALID;
    // spin-sim/src/config.rs:180-247 (message texts are the reference's)
    if (cfg->n_sweeps < 1) return "n_sweeps must be >= 1";
    if (cfg->warmup_sweeps > cfg->n_sweeps) return "warmup_sweeps must be <= n_sweeps";
    if (cfg->sweep_mode != PA_SWEEP_METROPOLIS && cfg->sweep_mode != PA_SWEEP_GIBBS)
        return "unknown sweep_mode, expected 'metropolis' or 'gibbs'";
    if (cfg->pt_schedule != PA_PT_SINGLE_RANDOM_EDGE && cfg->pt_schedule != PA_PT_FULL_LADDER)
        return "unknown pt_schedule, expected 'single_random_edge' or 'full_ladder'";
    if (cfg->autocorrelation_backend != PA_AUTOCORR_RING && cfg->autocorrelation_backend != PA_AUTOCORR_FFT)
        return "unknown autocorrelation_backend, expected 'ring' or 'fft'";
    if (cfg->autocorrelation_backend == PA_AUTOCORR_FFT && cfg->autocorrelation_max_lag == 0)
        return "autocorrelation_backend='fft' requires autocorrelation_max_lag";
    if (cfg->overlap_cluster_interval > 0) {
        if (cfg->n_overlap_cluster_modes < 1) return "overlap_cluster modes must not be empty";
        // simulation/mod.rs:207-214
        if (n_replicas < 2)
            return "overlap cluster requires n_replicas >= max group_size (" +
                   std::to_string(n_replicas) + " < 2)";
    }
    if (cfg->cluster_interval > 0 && cfg->cluster_mode != PA_CLUSTER_WOLFF && cfg->cluster_mode != PA_CLUSTER_SW)
        return "unknown cluster_mode, expected 'wolff' or 'sw'";
    if (cfg->cluster_interval > 0 && cfg->cluster_action == PA_ACTION_OBSERVE && cfg->cluster_mode == PA_CLUSTER_WOLFF)
        return "cluster_action='observe' requires cluster_mode='sw'";
    if (cfg->overlap_cluster_interval > 0) {
        if (cfg->snapshot_interval > 0 && cfg->snapshot_interval % cfg->overlap_cluster_interval != 0)
            return "snapshot_interval must be a positive multiple of overlap_cluster interval";
        if (cfg->overlap_cluster_action == PA_ACTION_OBSERVE) {
            if (cfg->overlap_cluster_mode == PA_CLUSTER_WOLFF)
                return "overlap_cluster_action='observe' requires overlap_cluster_mode='sw'";
            if (cfg->snapshot_interval > 0)
                return "snapshot_interval is not supported with overlap_cluster_action='observe'";
            for (int k = 0; k < cfg->n_overlap_cluster_modes && k < 4; ++k)
                if ((cfg->overlap_cluster_modes[k] & 0xFF) == PA_OC_HOUDAYER && (cfg->overlap_cluster_modes[k] >> 8) > 2)
                    return "overlap_cluster_action='observe' does not support experimental houdN with N > 2";
        }
    }
    *code = PA_ERR_UNSUPPORTED;
    if (cfg->overlap_cluster_interval > 0) {
        for (int k = 0; k < cfg->n_overlap_cluster_modes && k < 4; ++k) {
            // mode = PA_OC_* | group_size << 8 (group size 0 = 2; "houdN", config.rs:124-141)
            const int kind = cfg->overlap_cluster_modes[k] & 0xFF, group = cfg->overlap_cluster_modes[k] >> 8;
            if (kind != PA_OC_HOUDAYER && kind != PA_OC_JORG && kind != PA_OC_CMR) {
                *code = PA_ERR_INVALID;
                return "unknown overlap-cluster build mode";
            }
            if (group != 0 && (kind != PA_OC_HOUDAYER || group < 2 || group % 2 != 0 || group > 32)) {
                *code = PA_ERR_INVALID;
                return "Houdayer group size must be even and >= 2, got " + std::to_string(group);
            }
            if (group > 2 && static_cast<uint64_t>(group) > n_replicas) { // simulation/mod.rs:207-214
                *code = PA_ERR_INVALID;
                return "overlap cluster requires n_replicas >= max group_size (" + std::to_string(n_replicas) +
                       " < " + std::to_string(group) + ")";
            }
        }
        if (cfg->overlap_cluster_mode != PA_CLUSTER_WOLFF && cfg->overlap_cluster_mode != PA_CLUSTER_SW) {
            *code = PA_ERR_INVALID;
            return "unknown overlap_cluster_mode, expected 'wolff' or 'sw'";
        }
        if (n_replicas > 32) return "overlap cluster moves support at most 32 replicas";
    }
    *code = PA_OK;
    return "";
}

int run_sweeps(pa_sim *sim, const pa_sample_config *cfg) {
    const SimDims &d = sim->d;
    if (d.N > 65535 && (cfg->overlap_cluster_interval > 0 || cfg->cluster_interval > 0))
        return fail(sim, PA_ERR_UNSUPPORTED, "cluster moves are built for lattices of at most 65535 sites");
    bool needs_exp = cfg->cluster_interval > 0 && !d.unit; // bond probabilities 1 - exp(-c |J| / T) per bond
    if (!d.unit && cfg->overlap_cluster_interval > 0)
        for (int k = 0; k < cfg->n_overlap_cluster_modes && k < 4; ++k)
            needs_exp = needs_exp || (cfg->overlap_cluster_modes[k] & 0xFF) == PA_OC_JORG ||
                        (cfg->overlap_cluster_modes[k] & 0xFF) == PA_OC_CMR;
    if (cfg->overlap_cluster_interval > 0 && d.n_temps * d.n_pairs > 1024)
        return fail(sim, PA_ERR_UNSUPPORTED, "overlap-cluster moves support at most 1024 pairs per realisation");
    const bool gibbs = cfg->sweep_mode == PA_SWEEP_GIBBS;
    int rc = gibbs ? ensure_gibbs(sim) : PA_OK;
    if (rc) return rc;
    const bool oc_sw = cfg->overlap_cluster_interval > 0 && cfg->overlap_cluster_mode == PA_CLUSTER_SW;
    bool oc_cmr = false;
    for (int k = 0; cfg->overlap_cluster_interval > 0 && k < cfg->n_overlap_cluster_modes && k < 4; ++k)
        oc_cmr = oc_cmr || (cfg->overlap_cluster_modes[k] & 0xFF) == PA_OC_CMR || (cfg->overlap_cluster_modes[k] >> 8) > 2 ||
                 d.nn > 3;
    const bool fk = cfg->cluster_interval > 0;
    // (FK single-cluster form: one wave per system with the lattice tables in the LDS, else one lane per system with
    // the whole tile there)
    // (a call that reports sizes takes the union-find form; with warm-up sweeps the calls before the first recorded one
    // are searches all the same, mod.rs:437-484)
    const bool fk_sizes_always = cfg->collect_cluster_stats && cfg->warmup_sweeps == 0;
    if (fk && cfg->cluster_mode == PA_CLUSTER_WOLFF && !fk_sizes_always && cfg->cluster_action != PA_ACTION_OBSERVE &&
        fk_wolff_lds_bytes(d) > 160 * 1024 && !(wolff_wave_lds_bytes(d) <= 160 * 1024 && d.N <= 65535))
        return fail(sim, PA_ERR_UNSUPPORTED, "lattice too large for the FK cluster kernel");
    // Joerg single-cluster moves run one wave per pair when the lattice tables fit the LDS (k_wolff_wave)
    bool jorg_wave = false;
    for (int k = 0; cfg->overlap_cluster_interval > 0 && !oc_sw && k < cfg->n_overlap_cluster_modes && k < 4; ++k)
        jorg_wave = jorg_wave || (cfg->overlap_cluster_modes[k] & 0xFF) == PA_OC_JORG;
    const bool wave_ok = d.N <= 65535 && wolff_wave_lds_bytes(d) <= 160 * 1024 && cluster_sw_lds_bytes(d) <= 160 * 1024;
    jorg_wave = jorg_wave && wave_ok;
    // the single-cluster Houdayer search keeps the realisation's tile in the LDS; an ensemble whose tile does not fit
    // takes the move's union-find form, which reaches the same clusters from the same draws (overlap.rs:179-339)
    // (... and so does a large-lattice handle, whatever its slot count: the search kernels read the small-lattice site
    // records and a scratch tile that such a handle does not have)
    const bool houdayer_tile_fits = houdayer_lds_bytes(d) <= 160 * 1024 && !d.big;
    const bool has_houdayer_uf = cfg->overlap_cluster_interval > 0 && !houdayer_tile_fits;
    if ((oc_sw || fk || oc_cmr || jorg_wave || has_houdayer_uf) && (rc = ensure_cluster_sw(sim))) return rc;
    if (needs_exp && (rc = ensure_device_exp(sim))) return rc;
    if (cfg->equilibration_diagnostic && (rc = ensure_equil(sim))) return rc;
    // autocorrelation (mod.rs:342-345): max_lag capped at a quarter of the measurement sweeps, at least 1
    const uint64_t n_meas = cfg->n_sweeps - cfg->warmup_sweeps;
    const bool autocorr = cfg->autocorrelation_max_lag > 0;
    sim->ac_lag = autocorr ? std::max<uint64_t>(1, std::min<uint64_t>(cfg->autocorrelation_max_lag, n_meas / 4)) : 0;
    sim->ac_samples = 0;
    sim->ac_fft = cfg->autocorrelation_backend == PA_AUTOCORR_FFT;
    if (autocorr && n_meas > 0 && (rc = ensure_autocorr(sim, n_meas))) return rc;
    if ((rc = zero_accumulators(sim))) return rc;
    // what the cluster calls report besides moving spins (mod.rs:216-335)
    const bool has_oc = cfg->overlap_cluster_interval > 0;
    const uint32_t n_modes = has_oc ? static_cast<uint32_t>(cfg->n_overlap_cluster_modes) : 0u;
    const bool observe_fk = fk && cfg->cluster_action == PA_ACTION_OBSERVE;
    const bool observe_overlap = has_oc && cfg->overlap_cluster_action == PA_ACTION_OBSERVE;
    const bool collect_fk = fk && (cfg->collect_cluster_stats || observe_fk);
    const bool collect_overlap = has_oc && (cfg->collect_cluster_stats || observe_overlap) && d.n_pairs > 0;
    const uint64_t snapshot_interval = has_oc ? cfg->snapshot_interval : 0;
    sim->cs_fk = collect_fk;
    sim->cs_overlap = collect_overlap;
    sim->cs_n_modes = n_modes;
    sim->snapshots.clear();
    for (int k = 0; k < 4; ++k) sim->obs_kind[k] = false, sim->top4_calls[k] = 0;
    sim->obs_kind[0] = observe_fk;
    for (uint32_t m = 0; observe_overlap && m < n_modes; ++m) sim->obs_kind[cfg->overlap_cluster_modes[m] & 0xFF] = true;
    if (collect_fk || collect_overlap || snapshot_interval > 0) {
        if ((rc = ensure_cluster_sw(sim))) return rc;
        if ((rc = ensure_cluster_stats(sim, n_modes, snapshot_interval > 0))) return rc;
        if (snapshot_interval > 0) {
            sim->snap_ids_host.assign(static_cast<size_t>(d.n_temps) * d.N, 0u);
            sim->snap_blue_host.assign(static_cast<size_t>(d.n_temps) * d.N, 0u);
        }
    }
    const bool want_overlap = d.n_pairs > 0 && cfg->measure_overlap;
    sim->n_recorded = 0;
    sim->n_sweeps_run = 0;
    sim->overlap_collected = want_overlap;
    // every capability check comes before the first launch: a configuration this build declines is
    // rejected with the state untouched and no group stream forked
    // (calls that report sizes or observe take the union-find form -- every call when observing, the recorded ones when
    // collecting statistics: with warm-up sweeps the searches run first, overlap.rs:179-183)
    const bool oc_uf_always = observe_overlap || (collect_overlap && cfg->warmup_sweeps == 0);
    if (has_oc && !oc_sw && !wave_ok && !oc_uf_always) {
        // (depth-first Joerg / CMR searches without the one-wave-per-pair kernel: their one-lane-per-pair fallbacks
        // hold the realisation's tile and a bitmap per pair in the LDS)
        for (int k = 0; k < cfg->n_overlap_cluster_modes && k < 4; ++k) {
            const int kind = cfg->overlap_cluster_modes[k] & 0xFF;
            if ((kind == PA_OC_JORG && jorg_lds_bytes(d) > 160 * 1024) || (kind == PA_OC_CMR && cmr_wolff_lds_bytes(d) > 160 * 1024))
                return fail(sim, PA_ERR_UNSUPPORTED, "lattice too large for the single-cluster search of this overlap move "
                                                     "(its union-find form, overlap_cluster_mode='sw', has no such limit)");
        }
    }
    // equilibration diagnostic: energies and the link overlap of EVERY sweep (mod.rs:492,527)
    const bool equil = cfg->equilibration_diagnostic != 0;
    sim->equil_sweeps.clear();
    if (equil) { // EquilDiagnosticAccum::new, equilibration.rs:18-38
        for (uint64_t pw = 128; pw < cfg->n_sweeps; pw *= 2) sim->equil_sweeps.push_back(pw);
        if (sim->equil_sweeps.empty() || sim->equil_sweeps.back() != cfg->n_sweeps)
            sim->equil_sweeps.push_back(cfg->n_sweeps);
        if (sim->equil_sweeps.size() > 64) return fail(sim, PA_ERR_UNSUPPORTED, "more than 64 equilibration checkpoints");
    }
    // fork: every group stream starts after what is already queued on the main stream -- if anything is: after a
    // synchronised call the main stream is idle, and an event that every group's first kernel has to wait for would
    // only put a marker packet's round trip through the command processor in front of a short call
    if (hipStreamQuery(sim->stream) != hipSuccess) {
        PA_HIP(sim, hipEventRecord(sim->fork, sim->stream));
        for (auto &g : sim->groups) PA_HIP(sim, hipStreamWaitEvent(g.stream, sim->fork, 0));
    }
    uint64_t overlap_call_count = 0; // mod.rs:300
    size_t equil_next = 0;
    // (the heat bath rides the same launches when its sweep runs on the implicit-neighbour kernel: three planes in LDS)
    const bool gibbs_fast = gibbs && d.hc && sweep_hc_lds_bytes(d, true) <= 160 * 1024 && !dev_env("PEAPODS_NO_HC");
    const bool fused = want_overlap && (!gibbs || gibbs_fast) && !fk && !equil && cfg->overlap_cluster_interval == 0 &&
                       !d.big && !d.general && !dev_env("PEAPODS_NO_FUSE");
    bool overlap_owed = false; // the last recorded sweep's overlap has not been measured yet
    // the measurement block of the previous sweep when it is relayed into the next sweep's random-stream launch
    struct {
        bool any = false, do_accum = false, accum_overlap = false, do_pt = false;
        PtParams pt{};
    } pending;
    const bool relay = fused && !dev_env("PEAPODS_NO_RELAY") &&
                       (cfg->pt_schedule == PA_PT_SINGLE_RANDOM_EDGE ? rng_relay_supported(d) : rng_relay_ladder_supported(d));
    // (opt-in: measured slower than two launches)
    const bool slab_at_load = d.pipe_K && d.pipe_lds && !dev_env("PEAPODS_NO_SWAPS_AT_LOAD");
    const bool relay_nf = !fused && !gibbs && (!d.big || slab_at_load) && !d.general && cfg->pt_schedule == PA_PT_SINGLE_RANDOM_EDGE &&
                          rng_relay_supported(d) && !dev_env("PEAPODS_NO_RELAY");
    // Cooperative cancel (src/lib.rs:304-308, mod.rs:406-408): the host runs at most two batches of
    // kBatch sweeps ahead of the device, so a request is honoured within that much queued work; the
    // state then is the one after the last enqueued sweep (a continuation from it is well defined).
    constexpr uint64_t kBatch = 256;
    hipEvent_t batch_done[2] = {sim->batch_events[0], sim->batch_events[1]};
    bool batch_pending[2] = {false, false};
    // The sweeps are enqueued inside a lambda so that every failure (a HIP error, an interrupt) leaves
    // through the join below: the main stream always waits for the group streams before anything else
    // -- a getter, a reset, the next call -- touches the buffers.
    // One sweep of the plain protocol on the on-chip kernels: its launches (launch = true) and the host's bookkeeping --
    // buffer roles, the pending measurement block, counters.  With launch = false only the bookkeeping runs: the launches
    // are being replayed from a captured graph (graph_pair).
    // (the captured-graph path, opt-in: see graph_pair below)
    const bool graph_enabled = fused && relay && dev_env("PEAPODS_GRAPH") != nullptr && cfg->pt_interval == 1;
    auto fused_sweep = [&](uint64_t sweep_id, bool launch) -> int {
        const bool record = sweep_id >= cfg->warmup_sweeps; // mod.rs:410
        const bool pt_this = cfg->pt_interval > 0 && sweep_id % cfg->pt_interval == 0; // :486-488
        // Plain protocol on the LDS kernels: the overlap of the state sweep s-1 left is measured
        // inside the launch that draws the random stream of sweep s, and its statistics are
        // accumulated by the measurement block of sweep s.  Every accumulator still receives
        // its values in sweep order, so the f64 sums are unchanged.
        // With `relay` the measurement block of sweep s-1 (statistics + tempering event) rides in that
        // launch too: the random-stream workgroups replay the swap decisions and fetch their generators
        // from the slots the systems come from, so a sweep is two launches, not three.
        const bool relayed = relay && pending.any;
        // (enqueue order: every group's random stream first, then every group's sweep -- a launch costs the host
        // ~6 us, and in a short call the group enqueued second runs that much behind the first from its first
        // kernel to its last: two launches behind with the order group by group, profiles/r03_timeline_20steps.txt)
        for (auto &g : sim->groups) {
            if (relayed) {
                PtParams q = pending.pt;
                q.relay = 1;
                q.fresh = consume_fresh(g.fresh, g.d, pending.do_accum, pending.accum_overlap);
                if (launch)
                    PA_HIP(sim, launch_rng_relay(g.d, g.p, overlap_owed, q, pending.do_accum, pending.accum_overlap,
                                                 pending.do_pt, g.stream, gibbs));
            } else if (overlap_owed) {
                if (launch) PA_HIP(sim, launch_rng_overlap(g.d, g.p, g.stream, gibbs));
            } else {
                if (launch) PA_HIP(sim, launch_rng(g.d, g.p, gibbs, g.stream));
            }
            if (overlap_owed) toggle_dots(g.p);
        }
        // (a relayed tempering event files the new permutation but cannot touch the cached sums, which the launch's
        // random-stream workgroups read: where the sweep behind it does not recompute them -- pt_interval > 1 inside the
        // warm-up -- the exchanged sums it staged are filed now, so that pa_sim_get_energies / _magnetization_sums hold
        // what the reference's caches hold: the last computed values, with their systems)
        if (launch && relayed && pending.do_pt && d.n_temps >= 2 && !(record || pt_this))
            for (auto &g : sim->groups) PA_HIP(sim, launch_adopt_relay_sums(g.d, g.p, g.stream));
        if (launch)
            for (auto &g : sim->groups) PA_HIP(sim, enqueue_sweep(sim, g.d, g.p, record || pt_this, gibbs, g.stream));
        if (relayed && pending.do_pt && d.n_temps >= 2) toggle_system_tables(sim);
        pending.any = false;
        sim->swaps_pending = false;
        toggle_spin_buffers(sim);
        PtParams pt{cfg->pt_schedule, sim->next_parity, sim->hot_slot, sim->cold_slot};
        if (autocorr) { // this sweep's m^2 sample; the q^2 sample of the sweep whose overlap is accumulated with it
            pt.ac_index = record ? static_cast<uint32_t>(sim->n_recorded + 1) : 0u;
            pt.ac_index_q = overlap_owed ? static_cast<uint32_t>(sim->n_recorded) : 0u;
            pt.ac_stride = static_cast<uint32_t>(sim->ac_capacity);
        }
        if (record || pt_this || overlap_owed) {
            if (relay) { // deferred into the next sweep's launch (or the tail of the run)
                pending.any = true;
                pending.pt = pt;
                pending.do_accum = record;
                pending.accum_overlap = overlap_owed;
                pending.do_pt = pt_this;
            } else {
                for (auto &g : sim->groups) // (never on the graph path: it requires `relay`)
                    PA_HIP(sim, measure_launch(g.fresh, g.d, g.p, pt, record, overlap_owed, pt_this, g.stream));
                if (pt_this) sim->swaps_pending = true;
            }
        }
        overlap_owed = record;
        if (record) sim->n_recorded += 1;
        if (pt_this && cfg->pt_schedule == PA_PT_FULL_LADDER) sim->next_parity = 1 - sim->next_parity;
        return PA_OK;
    };
    // A pair of consecutive steady-state sweeps (every flag of both equal to the pair before: recorded, tempering event
    // every sweep, measurement relayed, accumulators past their first store) as ONE graph launch per stream group.  The
    // first such pair of a handle is captured from the ordinary launch code (stream capture on every group's stream)
    // and instantiated; `graph_key` hashes every argument the captured kernels received -- SimDims, all device
    // pointers (their double-buffer roles return after two sweeps), the tempering parameters -- and a later pair with
    // another key (a Gibbs call after a Metropolis one, the other ladder parity, reallocated buffers) is captured afresh.
    constexpr int kGraphDeclined = -1;
    // Measured on the bench ensemble and NOT adopted (opt-in: PEAPODS_GRAPH=1, kept exact by test_gpu_robustness.py):
    // steady state 57.6 -> 59.0 us per step, 20-step calls 64 -> 79 us per step.  hipGraphLaunch costs the host ~25 us
    // on this stack (two launches per pair and group: 0.58 ms of enqueueing per 20 steps against 0.41 ms launch by
    // launch), and the dependent-launch hop it shortens in an empty-kernel chain (1.6 against 2.8 us,
    // build_probe/gaps.hip) is not what separates these kernels: most of the ~5 us before a large kernel's first wave
    // is its own wave-launch ramp.
    auto pair_key = [&]() -> uint64_t {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *ptr, size_t n) {
            const unsigned char *b = static_cast<const unsigned char *>(ptr);
            for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
        };
        for (auto &g : sim->groups) {
            mix(&g.d, sizeof(g.d));
            mix(&g.p, sizeof(g.p));
        }
        mix(&pending.pt, sizeof(pending.pt));
        const uint32_t flags[4] = {static_cast<uint32_t>(cfg->pt_schedule), sim->next_parity, gibbs ? 1u : 0u,
                                   static_cast<uint32_t>(sim->groups.size())};
        mix(flags, sizeof(flags));
        return h | 1ull; // (0 = nothing captured)
    };
    auto graph_pair = [&](uint64_t sweep_id) -> int {
        const uint64_t key = pair_key();
        if (sim->graph_key == key && sim->pair_graphs.size() == sim->groups.size()) {
            for (size_t gi = 0; gi < sim->groups.size(); ++gi)
                PA_HIP(sim, hipGraphLaunch(sim->pair_graphs[gi], sim->groups[gi].stream));
            if (int rc = fused_sweep(sweep_id, false)) return rc;
            return fused_sweep(sweep_id + 1, false);
        }
        for (auto &ge : sim->pair_graphs)
            if (ge) (void)hipGraphExecDestroy(ge);
        sim->pair_graphs.clear();
        sim->graph_key = 0;
        size_t capturing = 0;
        hipError_t e = hipSuccess;
        for (; capturing < sim->groups.size() && e == hipSuccess; ++capturing)
            e = hipStreamBeginCapture(sim->groups[capturing].stream, hipStreamCaptureModeRelaxed);
        if (e != hipSuccess) { // no capture on this stack: end what was begun and run launch by launch
            for (size_t gi = 0; gi + 1 < capturing; ++gi) {
                hipGraph_t junk = nullptr;
                (void)hipStreamEndCapture(sim->groups[gi].stream, &junk);
                if (junk) (void)hipGraphDestroy(junk);
            }
            (void)hipGetLastError();
            return kGraphDeclined;
        }
        int rc = fused_sweep(sweep_id, true);
        if (rc == PA_OK) rc = fused_sweep(sweep_id + 1, true);
        std::vector<hipGraph_t> graphs(sim->groups.size(), nullptr);
        for (size_t gi = 0; gi < sim->groups.size(); ++gi) {
            const hipError_t ee = hipStreamEndCapture(sim->groups[gi].stream, &graphs[gi]);
            if (ee != hipSuccess && rc == PA_OK) rc = fail(sim, PA_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
        }
        sim->pair_graphs.assign(sim->groups.size(), nullptr);
        for (size_t gi = 0; gi < sim->groups.size() && rc == PA_OK; ++gi) {
            const hipError_t ee = hipGraphInstantiate(&sim->pair_graphs[gi], graphs[gi], nullptr, nullptr, 0);
            if (ee != hipSuccess) rc = fail(sim, PA_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ee));
        }
        for (auto &gr : graphs)
            if (gr) (void)hipGraphDestroy(gr);
        if (rc != PA_OK) { // (the two sweeps' bookkeeping has run but nothing was launched: the state is unusable)
            for (auto &ge : sim->pair_graphs)
                if (ge) (void)hipGraphExecDestroy(ge);
            sim->pair_graphs.clear();
            return rc;
        }
        sim->graph_key = key;
        for (size_t gi = 0; gi < sim->groups.size(); ++gi)
            PA_HIP(sim, hipGraphLaunch(sim->pair_graphs[gi], sim->groups[gi].stream));
        return PA_OK;
    };
    auto enqueue_all = [&]() -> int {
    for (uint64_t sweep_id = 0; sweep_id < cfg->n_sweeps; ++sweep_id) {
        if (sweep_id % kBatch == 0) {
            const unsigned slot = static_cast<unsigned>((sweep_id / kBatch) & 1u);
            if (batch_pending[slot]) { // the batch before the previous one: wait for it, polling the flag
                for (;;) {
                    const hipError_t q = hipEventQuery(batch_done[slot]);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) return fail(sim, PA_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
                    if (sim->interrupt.load(std::memory_order_relaxed)) break;
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                batch_pending[slot] = false;
            }
            if (sweep_id > 0) { // marks the end of the batch just enqueued (last group stream finishes last)
                const unsigned prev = slot ^ 1u;
                PA_HIP(sim, hipEventRecord(batch_done[prev], sim->groups.back().stream));
                batch_pending[prev] = true;
            }
        }
        if ((sweep_id & 31u) == 0 && sim->interrupt.load(std::memory_order_relaxed)) {
            sim->interrupt.store(false);
            return fail(sim, PA_ERR_INTERRUPTED, "interrupted");
        }
        sim->n_sweeps_run = sweep_id + 1;
        const bool record = sweep_id >= cfg->warmup_sweeps; // mod.rs:410
        const bool pt_this = cfg->pt_interval > 0 && sweep_id % cfg->pt_interval == 0; // :486-488
        const bool fk_this = fk && sweep_id % cfg->cluster_interval == 0;               // :434-437
        if (fused) {
            // Two steady-state sweeps at a time through a captured graph, when nothing about them differs from the pair
            // before (see graph_pair above the loop); otherwise launch by launch.
            if (graph_enabled && sweep_id + 1 < cfg->n_sweeps && sweep_id >= cfg->warmup_sweeps && pt_this &&
                (sweep_id % kBatch) != kBatch - 1 && relay && pending.any && pending.do_accum && pending.accum_overlap &&
                pending.do_pt && overlap_owed && !autocorr && d.n_temps >= 2 && !sim->interrupt.load(std::memory_order_relaxed)) {
                bool fresh_left = false;
                for (auto &g : sim->groups) fresh_left = fresh_left || g.fresh != 0;
                if (!fresh_left) {
                    const int rc = graph_pair(sweep_id);
                    if (rc == PA_OK) {
                        sim->n_sweeps_run = sweep_id + 2;
                        ++sweep_id; // (the pair covers the next sweep too)
                        continue;
                    }
                    if (rc != kGraphDeclined) return rc;
                }
            }
            if (int rc = fused_sweep(sweep_id, true)) return rc;
            continue;
        }
        // (general protocol) the previous sweep's measurement launch that carries its tempering event rides in this
        // sweep's random-stream launch when it can (relay_nf): one launch less per sweep
        const bool relayed_nf = relay_nf && pending.any;
        bool big_overlap_done = false;
        for (auto &g : sim->groups) {
            if (relayed_nf) {
                PtParams q = pending.pt;
                q.relay = 1;
                q.fresh = consume_fresh(g.fresh, g.d, pending.do_accum, pending.accum_overlap);
                PA_HIP(sim, launch_rng_relay(g.d, g.p, false, q, pending.do_accum, pending.accum_overlap, pending.do_pt,
                                             g.stream));
            } else {
                PA_HIP(sim, launch_rng(g.d, g.p, gibbs, g.stream));                     // :412-432
            }
            // energies are computed after every mutating move of the sweep (:492-509); on the pipelined large-lattice
            // path the overlap of the swept state is measured by the same launch when nothing comes between the two
            if (relayed_nf && pending.do_pt && d.n_temps >= 2 && !(record || pt_this || equil)) // (the event's exchanged sums: see fused_sweep)
                PA_HIP(sim, launch_adopt_relay_sums(g.d, g.p, g.stream));
            const bool with_e = (record || pt_this || equil) && !fk_this;
            const bool big_fuse = g.d.pipe_K && with_e && !fk_this && !equil && record && want_overlap && d.n_pairs && d.nn <= 3;
            // (slab path: the relayed event of this sweep's random-stream launch has left bit columns to exchange at load)
            if (relayed_nf && pending.do_pt && d.n_temps >= 2 && d.big) sim->swaps_pending = true;
            PA_HIP(sim, enqueue_sweep(sim, g.d, g.p, with_e, gibbs, g.stream, big_fuse));
            if (big_fuse) toggle_dots(g.p);
            big_overlap_done = big_fuse; // (one stream group on this path)
        }
        if (relayed_nf && pending.do_pt && d.n_temps >= 2) toggle_system_tables(sim);
        pending.any = false;
        bool pt_deferred = false;
        sim->swaps_pending = false;
        toggle_spin_buffers(sim);
        uint32_t equil_ckpt = 0; // 1 + index of the checkpoint this sweep completes
        if (equil && equil_next < sim->equil_sweeps.size() && sweep_id + 1 == sim->equil_sweeps[equil_next])
            equil_ckpt = static_cast<uint32_t>(++equil_next);
        for (auto &g : sim->groups) {
            if (fk_this) { // clusters::fk_update (:439-484), then the energies
                ClusterAux aux{};
                if (collect_fk && record) aux.flags |= kAuxSizes;
                if (observe_fk) aux.flags |= kAuxObserve | (record ? kAuxMetrics : 0u);
                aux.n_modes = sim->cs_modes;
                aux.square = sim->square ? 1u : 0u;
                PA_HIP(sim, launch_fk(g.d, g.p, cfg->cluster_mode == PA_CLUSTER_WOLFF, aux, g.stream));
                if (record || pt_this || equil) PA_HIP(sim, launch_sweep(g.d, g.p, true, false, false, g.stream));
            }
            const bool oc_this = cfg->overlap_cluster_interval > 0 &&
                                 sweep_id % cfg->overlap_cluster_interval == 0; // :598-599
            // The plain Houdayer single-cluster move (k_houdayer + k_houdayer_grow) only reads the spins in its first
            // launch: this sweep's overlap measurement rides in it, and the statistics follow between the two launches
            // (the same order of effects: sweep, overlap and statistics on the swept state, then the move).
            bool overlap_in_houdayer = false;
            if (oc_this && record && want_overlap && !equil && d.n_pairs && !d.big && d.nn <= 3) {
                const uint64_t mode_now = overlap_call_count % static_cast<uint64_t>(cfg->n_overlap_cluster_modes);
                const int mode = cfg->overlap_cluster_modes[mode_now];
                const bool first = &g == &sim->groups.front() && sim->first_realization == 0;
                const bool snap_now = snapshot_interval > 0 && sweep_id % snapshot_interval == 0;
                overlap_in_houdayer = (mode & 0xFF) == PA_OC_HOUDAYER && (mode >> 8) <= 2 && !oc_sw && !collect_overlap &&
                                      !observe_overlap && !(snap_now && first) && houdayer_tile_fits;
            }
            // The moves that keep their launches to themselves (Joerg, CMR, every union-find form): the shuffles of
            // build_tasks ride in the overlap launch, which also puts the sweep's sums aside, and the statistics of the swept
            // state ride in the launch that recomputes the energies after the move -- two launches less per step.
            bool move_riders = false, move_is_uf = false;
            if (oc_this && record && want_overlap && !equil && pt_this && move_riders_supported(d) && !overlap_in_houdayer &&
                !collect_overlap && !observe_overlap && snapshot_interval == 0 && !big_overlap_done &&
                !dev_env("PEAPODS_NO_MOVE_RIDERS")) {
                const int mode = cfg->overlap_cluster_modes[overlap_call_count % static_cast<uint64_t>(cfg->n_overlap_cluster_modes)];
                const int kind = mode & 0xFF;
                const bool uf = oc_sw || (mode >> 8) > 2 || (kind == PA_OC_HOUDAYER && (d.nn > 3 || !houdayer_tile_fits));
                move_riders = uf || (kind == PA_OC_JORG && jorg_wave) || (kind == PA_OC_CMR && wave_ok);
                move_is_uf = uf;
            }
            if (move_riders) {
                PA_HIP(sim, launch_overlap_pairing(g.d, g.p, g.stream, move_is_uf));
                toggle_dots(g.p);
            } else if (((record && want_overlap) || equil) && d.n_pairs && !overlap_in_houdayer && !big_overlap_done)
                PA_HIP(sim, overlap_launch(g.d, g.p, g.stream)); // :527-529
            PtParams pt{cfg->pt_schedule, sim->next_parity, sim->hot_slot, sim->cold_slot,
                        equil ? 1u : 0u, equil_ckpt, static_cast<uint32_t>(sweep_id + 1),
                        (autocorr && record) ? static_cast<uint32_t>(sim->n_recorded + 1) : 0u,
                        static_cast<uint32_t>(sim->ac_capacity)};
            pt.ac_index_q = pt.ac_index;
            if (!oc_this) {
                if (record || pt_this || equil) { // statistics :543-578, tempering :748-796
                    if (relay_nf && pt_this) { // rides in the next sweep's random-stream launch
                        pending.any = true;
                        pending.pt = pt;
                        pending.do_accum = record;
                        pending.accum_overlap = record && want_overlap;
                        pending.do_pt = true;
                        pt_deferred = true;
                    } else {
                        PA_HIP(sim, measure_launch(g.fresh, g.d, g.p, pt, record, record && want_overlap, pt_this,
                                                   g.stream));
                    }
                }
            } else {
                if ((record || equil) && !overlap_in_houdayer && !move_riders)
                    PA_HIP(sim, measure_launch(g.fresh, g.d, g.p, pt, record, record && want_overlap, false, g.stream));
                const PtParams pt_stats = pt; // (move_riders: the statistics launch comes after the move)
                pt.equil = 0; // the tempering launch below must not accumulate the diagnostic again
                const uint64_t mode_idx = overlap_call_count % static_cast<uint64_t>(cfg->n_overlap_cluster_modes);
                const int oc_mode = cfg->overlap_cluster_modes[mode_idx];
                const int oc_kind = oc_mode & 0xFF;
                const uint32_t oc_group = (oc_mode >> 8) > 2 ? static_cast<uint32_t>(oc_mode >> 8) : 2u;
                // ... of GLOBAL realisation 0 (realization_idx == 0, mod.rs:302): a handle holding a later shard
                // of the ensemble takes none, and its first realisation stays on the search path like the rest
                const bool first_group = &g == &sim->groups.front() && sim->first_realization == 0;
                bool energy_done = false; // the move's kernel left the new energies itself (Houdayer growth kernel)
                // mod.rs:604-658: sizes / metrics on recorded calls, snapshots of the first realisation
                const bool take_snapshot = snapshot_interval > 0 && sweep_id % snapshot_interval == 0 && record;
                ClusterAux aux{};
                if (collect_overlap && record) aux.flags |= kAuxSizes;
                if (observe_overlap) aux.flags |= kAuxObserve | (record ? kAuxMetrics : 0u);
                aux.mode_idx = static_cast<uint32_t>(mode_idx);
                aux.n_modes = sim->cs_modes;
                aux.kind = static_cast<uint32_t>(oc_kind);
                aux.square = sim->square ? 1u : 0u;
                if (aux.flags & kAuxSizes)
                    PA_HIP(sim, hipMemsetAsync(g.p.cs_top4_raw, 0, sizeof(uint32_t) * 4 * g.d.n_real * d.n_temps * d.n_pairs,
                                               g.stream));
                // union-find variants; groups above two and lattices with more than three forward
                // directions also for the single-cluster Houdayer move (same components, same seed rule).
                // Every call that reports sizes or takes a snapshot is on the union-find path too
                // (`use_uf`, overlap.rs:179-183,376-380,640-644) -- a snapshot only for the first realisation.
                const bool uf_all = oc_sw || oc_group > 2 || (oc_kind == PA_OC_HOUDAYER && (d.nn > 3 || !houdayer_tile_fits)) ||
                                    aux.flags != 0;
                const bool dfs_same = oc_kind == PA_OC_HOUDAYER; // deterministic bonds: both paths consume the same draws
                if (uf_all || (take_snapshot && first_group && dfs_same)) {
                    ClusterAux a2 = aux;
                    if (take_snapshot && first_group) a2.flags |= kAuxSnapshot;
                    DevPtrs qq = g.p;
                    qq.slab_stamps = sim->cluster_stamps; // diagnostic (pa_debug_cluster_stamps), normally null
                    PA_HIP(sim, launch_cluster_sw(g.d, qq, oc_kind, oc_group, !oc_sw, a2, g.stream, move_riders));
                } else if (take_snapshot && first_group) {
                    // the first realisation alone takes the union-find path, the others the search
                    SimDims d0 = g.d, d1 = g.d;
                    d0.n_real = 1;
                    d1.n_real = g.d.n_real - 1;
                    ClusterAux a2 = aux;
                    a2.flags |= kAuxSnapshot;
                    PA_HIP(sim, launch_cluster_sw(d0, g.p, oc_kind, oc_group, true, a2, g.stream));
                    if (d1.n_real > 0) {
                        DevPtrs p1 = offset_ptrs(g.p, g.d, 1);
                        if (oc_kind == PA_OC_CMR && wave_ok)
                            PA_HIP(sim, launch_cmr_wave(d1, p1, g.stream));
                        else if (oc_kind == PA_OC_CMR)
                            PA_HIP(sim, launch_cmr_wolff(d1, p1, g.stream));
                        else if (jorg_wave)
                            PA_HIP(sim, launch_jorg_wave(d1, p1, g.stream));
                        else
                            PA_HIP(sim, launch_jorg(d1, p1, g.stream));
                    }
                } else if (oc_kind == PA_OC_CMR && wave_ok)
                    PA_HIP(sim, launch_cmr_wave(g.d, g.p, g.stream, move_riders)); // :845-965, one wave per pair
                else if (oc_kind == PA_OC_CMR)
                    PA_HIP(sim, launch_cmr_wolff(g.d, g.p, g.stream));
                else if (oc_kind == PA_OC_JORG && jorg_wave)
                    PA_HIP(sim, launch_jorg_wave(g.d, g.p, g.stream, move_riders)); // one wave per pair, tables in the LDS
                else if (oc_kind == PA_OC_JORG)
                    PA_HIP(sim, launch_jorg(g.d, g.p, g.stream)); // :601-602,660-680
                else {
                    energy_done = pt_this && !observe_overlap && houdayer_computes_energy(g.d);
                    if (overlap_in_houdayer) {
                        PA_HIP(sim, launch_houdayer_seeds(g.d, g.p, g.stream, true));
                        toggle_dots(g.p);
                        // (the measurement of the swept state rides in the growth launch, reading the sums the seed
                        // launch put aside)
                        PtParams q = pt;
                        q.fresh = consume_fresh(g.fresh, g.d, record, true);
                        PA_HIP(sim, launch_houdayer_grow_measure(g.d, g.p, q, record, g.stream, energy_done));
                    } else {
                        PA_HIP(sim, launch_houdayer(g.d, g.p, g.stream, energy_done));
                    }
                }
                if (aux.flags & kAuxSizes) PA_HIP(sim, launch_top4_accumulate(g.d, g.p, aux, g.stream)); // :725-735
                if (take_snapshot && first_group) { // :682-714
                    PA_HIP(sim, launch_uf_replay(g.d, g.p, oc_kind == PA_OC_CMR, g.stream));
                    PA_HIP(sim, hipStreamSynchronize(g.stream));
                    pa_sim::Snapshot snap;
                    snap.sweep_id = sweep_id;
                    snap.mode_idx = static_cast<int32_t>(mode_idx);
                    snap.has_blue = oc_kind == PA_OC_CMR;
                    const size_t cells = static_cast<size_t>(d.n_temps) * d.N;
                    snap.ids.resize(cells);
                    snap.spins.resize(2 * cells);
                    std::vector<uint32_t> sids(static_cast<size_t>(d.n_temps) * 4);
                    PA_HIP(sim, hipMemcpy(snap.ids.data(), sim->p.snap_ids, sizeof(uint32_t) * cells, hipMemcpyDeviceToHost));
                    PA_HIP(sim, hipMemcpy(snap.spins.data(), sim->p.snap_spins, 2 * cells, hipMemcpyDeviceToHost));
                    PA_HIP(sim, hipMemcpy(sids.data(), sim->p.snap_sids, sizeof(uint32_t) * sids.size(), hipMemcpyDeviceToHost));
                    if (snap.has_blue) {
                        snap.blue.resize(cells);
                        PA_HIP(sim, hipMemcpy(snap.blue.data(), sim->p.snap_blue, sizeof(uint32_t) * cells, hipMemcpyDeviceToHost));
                    }
                    for (uint32_t t = 0; t < d.n_temps; ++t) {
                        snap.sids.push_back(sids[t * 4]);
                        snap.sids.push_back(sids[t * 4 + 1]);
                    }
                    sim->snapshots.push_back(std::move(snap));
                }
                if (pt_this) { // energies are recomputed after a mutating move (:749-756)
                    if (move_riders) {
                        PtParams q = pt_stats;
                        q.fresh = consume_fresh(g.fresh, g.d, record, record && want_overlap);
                        PA_HIP(sim, launch_energy_measure(g.d, g.p, q, record, record && want_overlap, g.stream));
                    } else if (!observe_overlap && !energy_done)
                        PA_HIP(sim, launch_sweep(g.d, g.p, true, false, false, g.stream));
                    if (relay_nf) { // the tempering event rides in the next sweep's random-stream launch
                        pending.any = true;
                        pending.pt = pt;
                        pending.do_accum = pending.accum_overlap = false;
                        pending.do_pt = true;
                        pt_deferred = true;
                    } else {
                        PA_HIP(sim, measure_launch(g.fresh, g.d, g.p, pt, false, false, true, g.stream));
                    }
                }
            }
        }
        if (pt_this && !pt_deferred) sim->swaps_pending = true;
        if (record) sim->n_recorded += 1;
        if (cfg->overlap_cluster_interval > 0 && sweep_id % cfg->overlap_cluster_interval == 0) {
            if (collect_overlap && record) // :734
                sim->top4_calls[overlap_call_count % static_cast<uint64_t>(cfg->n_overlap_cluster_modes)] += 1;
            overlap_call_count += 1; // :744
        }
        if (pt_this && cfg->pt_schedule == PA_PT_FULL_LADDER) sim->next_parity = 1 - sim->next_parity;
    }
    return PA_OK;
    };
    int status = enqueue_all();
    // The tail of the fused protocol -- also after an interrupt or a failed launch, so that the accumulators hold
    // whole sweeps only and the last sweep's tempering event has run: the deferred measurement block of the last
    // sweep (as its own launch now), then that sweep's overlap.
    if (pending.any && overlap_owed && relay) {
        // one launch for both: the overlap blocks write dot_*, the measurement block reads dot_*_prev -- the data flow
        // of every relayed launch of the run, without its random-stream workgroups
        for (auto &g : sim->groups) {
            PtParams q = pending.pt;
            q.relay = 0;
            q.fresh = consume_fresh(g.fresh, g.d, pending.do_accum, pending.accum_overlap);
            hipError_t e = launch_measure_overlap(g.d, g.p, q, pending.do_accum, pending.accum_overlap, pending.do_pt, g.stream);
            toggle_dots(g.p);
            PtParams pt{cfg->pt_schedule, sim->next_parity, sim->hot_slot, sim->cold_slot};
            pt.ac_index_q = autocorr ? static_cast<uint32_t>(sim->n_recorded) : 0u;
            pt.ac_stride = static_cast<uint32_t>(sim->ac_capacity);
            if (e == hipSuccess) e = measure_launch(g.fresh, g.d, g.p, pt, false, true, false, g.stream);
            if (e != hipSuccess && status == PA_OK) status = fail(sim, PA_ERR_DEVICE, std::string("fused tail: ") + hipGetErrorString(e));
        }
        if (pending.do_pt) sim->swaps_pending = true;
        pending.any = false;
        overlap_owed = false;
    }
    if (pending.any) {
        for (auto &g : sim->groups) {
            const hipError_t e = measure_launch(g.fresh, g.d, g.p, pending.pt, pending.do_accum, pending.accum_overlap,
                                                pending.do_pt, g.stream);
            if (e != hipSuccess && status == PA_OK) status = fail(sim, PA_ERR_DEVICE, std::string("launch_measure: ") + hipGetErrorString(e));
        }
        if (pending.do_pt) sim->swaps_pending = true;
        pending.any = false;
    }
    if (overlap_owed) {
        PtParams pt{cfg->pt_schedule, sim->next_parity, sim->hot_slot, sim->cold_slot};
        pt.ac_index_q = (fused && autocorr) ? static_cast<uint32_t>(sim->n_recorded) : 0u;
        pt.ac_stride = static_cast<uint32_t>(sim->ac_capacity);
        for (auto &g : sim->groups) {
            hipError_t e = overlap_launch(g.d, g.p, g.stream);
            if (e == hipSuccess) e = measure_launch(g.fresh, g.d, g.p, pt, false, true, false, g.stream);
            if (e != hipSuccess && status == PA_OK) status = fail(sim, PA_ERR_DEVICE, std::string("overlap tail: ") + hipGetErrorString(e));
        }
        overlap_owed = false;
    }
    sim->ac_samples = autocorr ? sim->n_recorded : 0;
    // join: the main stream continues after every group -- on every path out of this function
    for (auto &g : sim->groups) {
        const hipError_t e1 = hipEventRecord(g.done, g.stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(sim->stream, g.done, 0) : e1;
        if (e2 != hipSuccess) {
            (void)hipStreamSynchronize(g.stream); // no event: wait for the group on the host instead
            if (status == PA_OK) status = fail(sim, PA_ERR_DEVICE, std::string("join of a stream group: ") + hipGetErrorString(e2));
        }
    }
    if (status != PA_OK) {
        const std::string keep = sim->error;
        (void)flush_swaps(sim);
        (void)hipStreamSynchronize(sim->stream);
        sim->error = keep;
        return status;
    }
    // The last tempering event's bit-column swaps stay deferred (swaps_pending): the next sweep applies them
    // while it loads, every other reader (getters, standalone phases) materialises them first (flush_swaps).
    return PA_OK;
}

} // namespace

extern "C" {


int pa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint64_t pa_splitmix64(uint64_t value) { return splitmix64(value); }
uint64_t pa_child_seed(uint64_t root, uint64_t domain, uint64_t index) {
    return child_seed(root, domain, index);
}
uint64_t pa_realization_seed(uint64_t root, uint64_t realization) {
    return realization_seed(root, realization);
}

void pa_xoshiro_state(uint64_t seed, uint64_t n_skip, uint64_t out[4]) {
    Xoshiro rng = Xoshiro::from_seed_u64(seed);
    for (uint64_t i = 0; i < n_skip; ++i) rng.next_u64();
    std::memcpy(out, rng.s, 32);
}

void pa_xoshiro_jump(const uint64_t in[4], uint64_t n_steps, uint64_t out[4]) {
    // through the same nibble-table form the device consumes
    JumpMatrix m = JumpMatrix::steps(n_steps);
    std::vector<uint32_t> table(64 * 16 * 8);
    m.nibble_table(table.data());
    uint32_t st[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) {
        st[2 * w] = static_cast<uint32_t>(in[w]);
        st[2 * w + 1] = static_cast<uint32_t>(in[w] >> 32);
    }
    for (int p = 0; p < 64; ++p) {
        const uint32_t nib = (st[p >> 3] >> ((p & 7) * 4)) & 15u;
        for (int j = 0; j < 8; ++j) acc[j] ^= table[(p * 2 + j / 4) * 64 + nib * 4 + j % 4];
    }
    for (int w = 0; w < 4; ++w) out[w] = (static_cast<uint64_t>(acc[2 * w + 1]) << 32) | acc[2 * w];
}

int pa_build_metropolis_table(const float *couplings, uint64_t n_couplings,
                              const float *temperatures, uint64_t n_temperatures,
                              uint64_t n_neighbors, uint32_t *out) {
    std::vector<uint32_t> table;
    if (!build_metropolis_table(couplings, n_couplings, temperatures, n_temperatures, n_neighbors,
                                table))
        return PA_ERR_UNSUPPORTED;
    std::memcpy(out, table.data(), table.size() * sizeof(uint32_t));
    return PA_OK;
}

int pa_build_gibbs_table(const float *couplings, uint64_t n_couplings, const float *temperatures,
                         uint64_t n_temperatures, uint64_t n_neighbors, uint32_t *out) {
    std::vector<uint32_t> table;
    if (!build_gibbs_table(couplings, n_couplings, temperatures, n_temperatures, n_neighbors, table))
        return PA_ERR_UNSUPPORTED;
    std::memcpy(out, table.data(), table.size() * sizeof(uint32_t));
    return PA_OK;
}

uint64_t pa_build_sweep_levels(uint32_t n_dims, const uint64_t *shape, uint32_t n_neighbors,
                               const int64_t *neighbor_offsets, uint32_t *level_out) {
    LatticeTables lat = LatticeTables::build(n_dims, shape, n_neighbors, neighbor_offsets);
    std::vector<uint32_t> level;
    const uint32_t n_levels = build_levels(lat, level);
    std::memcpy(level_out, level.data(), level.size() * sizeof(uint32_t));
    return n_levels;
}

void pa_autocorrelation_tau(const float *series, uint64_t n_samples, uint64_t n_cols, uint64_t max_lag, double *tau_out,
                            int32_t backend) {
    autocorrelation_tau(series, n_samples, n_cols, max_lag, tau_out, backend == PA_AUTOCORR_FFT);
}

void pa_aggregate_mean(const double *per_realization, uint64_t n_realizations, uint64_t len,
                       double *out) {
    for (uint64_t i = 0; i < len; ++i) out[i] = 0.0;
    for (uint64_t r = 0; r < n_realizations; ++r)
        for (uint64_t i = 0; i < len; ++i) out[i] += per_realization[r * len + i];
    if (n_realizations > 1) { // a single realisation is returned as is (mod.rs:874-885)
        const double n = static_cast<double>(n_realizations);
        for (uint64_t i = 0; i < len; ++i) out[i] /= n;
    }
}

void pa_aggregate_sum_u64(const uint64_t *per_realization, uint64_t n_realizations, uint64_t len,
                          uint64_t *out) {
    for (uint64_t i = 0; i < len; ++i) out[i] = 0;
    for (uint64_t r = 0; r < n_realizations; ++r)
        for (uint64_t i = 0; i < len; ++i) out[i] += per_realization[r * len + i];
}

void pa_aggregate_sum_f64(const double *per_realization, uint64_t n_realizations, uint64_t len,
                          double *out) {
    for (uint64_t i = 0; i < len; ++i) out[i] = 0.0;
    for (uint64_t r = 0; r < n_realizations; ++r)
        for (uint64_t i = 0; i < len; ++i) out[i] += per_realization[r * len + i];
}

int pa_validate_sample_config(const pa_sample_config *cfg, uint64_t n_replicas, char *err,
                              size_t err_len) {
    int code;
    const std::string text = validate_text(cfg, n_replicas, &code);
    if (code != PA_OK) write_err(err, err_len, text);
    return code;
}

int pa_sim_create(const pa_sim_desc *desc, pa_sim **out, char *err, size_t err_len) {
    *out = nullptr;
    auto reject = [&](int code, const std::string &msg) {
        write_err(err, err_len, msg);
        return code;
    };
    if (!desc || desc->n_dims < 1 || desc->n_dims > 8 || !desc->shape)
        return reject(PA_ERR_INVALID, "lattice_shape must have 1..8 dimensions");
    for (uint32_t k = 0; k < desc->n_dims; ++k)
        if (desc->shape[k] < 1) return reject(PA_ERR_INVALID, "lattice extents must be >= 1");
    if (desc->n_temps < 1) return reject(PA_ERR_INVALID, "temperatures must not be empty");
    if (desc->n_replicas < 1) return reject(PA_ERR_INVALID, "n_replicas must be >= 1");
    if (desc->n_realizations < 1 || !desc->couplings || !desc->temperatures)
        return reject(PA_ERR_INVALID, "couplings and temperatures are required");

    int n_devices = 0;
    if (hipGetDeviceCount(&n_devices) != hipSuccess || n_devices < 1)
        return reject(PA_ERR_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (desc->device < 0 || desc->device >= n_devices)
        return reject(PA_ERR_DEVICE, "HIP device ordinal out of range");

    pa_sim *sim = new pa_sim();
    auto bail = [&](int code) {
        write_err(err, err_len, sim->error);
        pa_sim_destroy(sim);
        return code;
    };
    sim->device = desc->device;
    sim->lat = LatticeTables::build(desc->n_dims, desc->shape,
                                    desc->neighbor_offsets ? desc->n_neighbors : desc->n_dims,
                                    desc->neighbor_offsets);
    const LatticeTables &lat = sim->lat;
    sim->square = desc->neighbor_offsets == nullptr && desc->n_dims == 2; // Lattice::new, lattice.rs:31-38
    if (lat.n_neighbors > 6) {
        sim->error = "more than 6 forward directions are not built on the HIP path yet";
        return bail(PA_ERR_UNSUPPORTED);
    }
    if (lat.n_spins > (1u << 22) || desc->n_temps > 1024) {
        sim->error = "lattices above 4194304 sites (or more than 1024 temperatures) are not built "
                     "on the HIP path yet";
        return bail(PA_ERR_UNSUPPORTED);
    }
    sim->n_replicas = desc->n_replicas;
    sim->n_temps = desc->n_temps;
    sim->n_real = desc->n_realizations;
    sim->constructor_seed = desc->seed;
    sim->first_realization = desc->first_realization;
    sim->stream_groups_wanted = desc->stream_groups;
    sim->temps.assign(desc->temperatures, desc->temperatures + desc->n_temps);

    SimDims &d = sim->d;
    d.N = static_cast<uint32_t>(lat.n_spins);
    d.n_sys = static_cast<uint32_t>(desc->n_replicas * desc->n_temps);
    d.W = (d.n_sys + 31) / 32;
    d.SP = d.W * 32;
    d.n_temps = static_cast<uint32_t>(desc->n_temps);
    d.n_replicas = static_cast<uint32_t>(desc->n_replicas);
    d.n_pairs = d.n_replicas / 2;
    d.n_real = static_cast<uint32_t>(desc->n_realizations);
    d.nn = lat.n_neighbors;
    // sites per RNG block (tuning knob PEAPODS_RNG_BLOCK): 128 gives twice the waves of 256 for 7 % more
    // jump-ahead work and is 7 % faster per step on the bench workload (64, 96, 160, 256, 512 measured)
    uint32_t target_block = 128;
    // a jump table (32 KB) is staged once per workgroup: with few systems sharing it (small ensembles on large
    // lattices) the staging traffic dominates and longer blocks win (64^3 x 64 systems: 43 -> 32 us per sweep)
    // small ensembles altogether (the sizes of the reference's physics scripts): the block is one lane's serial
    // loop, so shorter blocks until ~2 workgroups per CU exist (32^2 x 8 systems: 46 -> 34 us per sweep; 8^3
    // Gaussian x 64 samples: 173 -> 102 us)
    const uint64_t rng_groups = (static_cast<uint64_t>(d.n_real) * d.W + 7) / 8; // workgroups per block of sites
    if (static_cast<uint64_t>(d.n_real) * d.W * 32 < 4096 && rng_groups * (d.N / 256) >= 512)
        target_block = 256;
    else if (rng_groups * (d.N / 128) < 512)
        target_block = static_cast<uint32_t>(std::min<uint64_t>(128, std::max<uint64_t>(32, rng_groups * d.N / 512)));
    if (const char *env = dev_env("PEAPODS_RNG_BLOCK")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 16 && v <= 65536) target_block = static_cast<uint32_t>(v);
    }
    {
        auto recip = [](uint32_t x) { return static_cast<uint32_t>(((1ull << 32) + x - 1) / x); }; // (x >= 2; x = 1: see div_small)
        d.inv_temps = recip(d.n_temps);
        d.inv_replicas = recip(d.n_replicas);
        d.inv_W = recip(d.W);
        // div_small is exact while dividend x divisor < 2^32: tile words (< n_real W + 8) by W, slots (< 32 W) by
        // n_temps, ladder tasks (<= 256) by n_replicas.  An ensemble beyond that is declined, never mis-indexed.
        const uint64_t n_tw = static_cast<uint64_t>(d.n_real) * d.W + 8;
        if (n_tw * d.W >= (1ull << 32) || static_cast<uint64_t>(d.SP) * d.n_temps >= (1ull << 32) ||
            512ull * d.n_replicas >= (1ull << 32)) {
            sim->error = "ensemble too large for the 32-bit index arithmetic of the random-stream kernels "
                         "(n_realizations x ceil(n_systems / 32)^2 must stay below 2^32)";
            return bail(PA_ERR_UNSUPPORTED);
        }
        d.rng_waves = 4;
        if (const char *env = dev_env("PEAPODS_RNG_WAVES")) {
            const long v = std::strtol(env, nullptr, 10);
            if (v >= 4 && v <= 8) d.rng_waves = static_cast<uint32_t>(v);
        }
    }
    d.B = std::max<uint32_t>(1, (d.N + target_block / 2) / target_block);
    d.BS = (d.N + d.B - 1) / d.B;
    d.B = (d.N + d.BS - 1) / d.BS;
    d.MW = (d.N + 31) / 32;
    // ---- acceptance rule: integer table (unit couplings, sweep.rs:109-139) or the logarithm
    const size_t n_coup = static_cast<size_t>(d.N) * d.nn;
    std::vector<uint32_t> table;
    d.unit = build_metropolis_table(desc->couplings, n_coup * d.n_real, sim->temps.data(), d.n_temps, d.nn,
                                    table) ? 1 : 0;
    // the bit-sliced table kernels hold up to three forward directions; lattices with four to six
    // (bcc, fcc, ...) take the logarithmic rule, which decides every flip identically (the table
    // is the bisection of that rule, sweep.rs:141-155)
    d.general = (!d.unit || d.nn > 3) ? 1 : 0;
    // the LDS kernels hold a word's N sites (sweep) / the W-word tile (tempering swaps) on chip
    // (the lattice decides: a sweep workgroup holds ONE word's N sites, whatever the number of words.  Kernels that
    // keep a realisation's whole W-word tile on chip -- the single-cluster Houdayer search k_houdayer, the one-lane-
    // per-pair fallbacks -- are chosen per call in run_sweeps only where the tile fits, so an ensemble of many slots,
    // 16^3 x 64 temperatures x 8 replicas = 16 words, keeps the on-chip path and every cluster move)
    d.big = d.N > 12000 ? 1 : 0;
    if (d.general && sweep_f_lds_bytes(d) > 160 * 1024) d.big = 1;
    // unit couplings on four to six forward directions: Metropolis sweeps by the integer table (SimDims::wide)
    d.wide = (d.general && d.unit && d.nn > 3 && d.nn <= 6 && !d.big && d.N <= 16383 &&
              sweep_wide_lds_bytes(d) <= 160 * 1024 && !dev_env("PEAPODS_NO_WIDE")) ? 1 : 0;
    std::vector<uint32_t> thr_wide;
    if (d.wide) { // the table's columns in the layout the random-stream kernel reads for the heat bath (thr_gibbs)
        const uint32_t width = 4 * d.nn + 1, off = 2 * d.nn;
        thr_wide.assign(static_cast<size_t>(14) * d.n_temps, 0u);
        for (uint32_t t = 0; t < d.n_temps && d.wide; ++t) {
            for (uint32_t j = 0; j < 7; ++j) {
                if (2 * (j + 1) <= off) thr_wide[(0 + j) * d.n_temps + t] = table[t * width + off - 2 * (j + 1)]; // dE = -2 (j + 1)
                if (2 * j + 1 <= off) thr_wide[(7 + j) * d.n_temps + t] = table[t * width + off - (2 * j + 1)];   // dE = -(2 j + 1)
            }
            for (uint32_t e = 1; e <= off; ++e) // the levels are nested only if the table falls with |dE| (it does: sweep.rs:141-159)
                if (table[t * width + e] < table[t * width + e - 1]) d.wide = 0;
        }
    }
    if (d.big && d.nn > 3) {
        sim->error = "lattices with more than 3 forward directions are built for the on-chip kernels only "
                     "(about 4000 sites)";
        return bail(PA_ERR_UNSUPPORTED);
    }
    d.n_edges = std::max<uint32_t>(1, d.n_temps - 1);
    if (d.W > 64) {
        sim->error = "more than 2048 systems per realisation are not built on the HIP path yet";
        return bail(PA_ERR_UNSUPPORTED);
    }

    // ---- acceptance thresholds of the table path
    const uint32_t width = 4 * d.nn + 1, off = 2 * d.nn;
    std::vector<uint32_t> thr(12 * d.n_temps, 0u);
    for (uint32_t t = 0; t < d.n_temps && !d.general; ++t) {
        for (uint32_t j = 0; j < 3; ++j) {
            const uint32_t even_e = 2 * (j + 1), odd_e = 2 * j + 1; // |dE|
            if (even_e <= off) thr[(0 + j) * d.n_temps + t] = table[t * width + off - even_e];
            if (odd_e <= off) thr[(3 + j) * d.n_temps + t] = table[t * width + off - odd_e];
        }
        for (uint32_t e = 1; e <= off; ++e) // levels need thresholds that fall with |dE|
            if (table[t * width + off - e] > table[t * width + off - e + 1]) {
                sim->error = "acceptance table is not monotone in the energy change";
                return bail(PA_ERR_UNSUPPORTED);
            }
    }

    d.thr_preshifted = d.general ? 0 : 1;
    for (uint32_t k = 0; k < 6 * d.n_temps; ++k) {
        if (thr[k] >= (1u << 24)) d.thr_preshifted = 0;
        thr[6 * d.n_temps + k] = thr[k] << 8;
    }

    // ---- dependency levels -> schedule order, cut into passes of <= 256 sites
    std::vector<uint32_t> level;
    const uint32_t n_levels = build_levels(lat, level);
    std::vector<uint32_t> order(d.N);
    for (uint32_t i = 0; i < d.N; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return level[a] < level[b]; });
    std::vector<uint32_t> level_start(n_levels + 1, 0);
    for (uint32_t i = 0; i < d.N; ++i) level_start[level[i] + 1] += 1;
    for (uint32_t l = 0; l < n_levels; ++l) level_start[l + 1] += level_start[l];
    std::vector<uint32_t> pass_start;
    for (uint32_t l = 0; l < n_levels; ++l)
        for (uint32_t b = level_start[l]; b < level_start[l + 1]; b += (d.big ? 1024 : 256))
            pass_start.push_back(b);
    pass_start.push_back(d.N);
    d.n_passes = static_cast<uint32_t>(pass_start.size()) - 1;

    // ---- large unit-coupling lattices: slabs of the pipelined sweep (k_sweep_pipe).  K slabs of
    // consecutive site indices are admissible when every smaller neighbour j of a site i that lies
    // m >= 1 slabs below satisfies level(j) + m <= level(i).
    d.n_levels = n_levels;
    d.slab_lead = 0; // (measured at 64^3: 0..4 levels make no difference, every further level costs 7 us per sweep)
    if (const char *env = dev_env("PEAPODS_SLAB_LEAD")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 0 && v <= 64) d.slab_lead = static_cast<uint32_t>(v);
    }
    d.pipe_K = 0;
    d.pipe_lds = 0;
    d.plane = 0;
    std::vector<uint32_t> seg;
    if (d.big && !d.general) {
        uint32_t widest = 1;
        for (uint32_t l = 0; l < n_levels; ++l) widest = std::max(widest, level_start[l + 1] - level_start[l]);
        // about 96 sites per slab and level, with slabs made of whole hyperplanes of the slowest
        // dimension (the largest divisor of its extent that does not exceed the target)
        const uint32_t target = std::min<uint32_t>(64, std::max<uint32_t>(1, widest / 96));
        uint32_t K = 1;
        for (uint32_t c = 1; c <= target; ++c)
            if (lat.shape[0] % c == 0) K = c;
        if (const char *env = dev_env("PEAPODS_PIPE_K")) {
            const long v = std::strtol(env, nullptr, 10);
            if (v >= 1 && v <= 256) K = static_cast<uint32_t>(v);
        }
        auto admissible = [&](uint32_t slabs) {
            const uint32_t slab = (d.N + slabs - 1) / slabs;
            for (uint32_t i = 0; i < d.N; ++i)
                for (uint32_t q = 0; q < d.nn; ++q)
                    for (uint32_t j : {lat.fwd[i * d.nn + q], lat.bwd[i * d.nn + q]}) {
                        if (j >= i) continue;
                        const uint32_t m = i / slab - j / slab;
                        if (m >= 1 && level[j] + m > level[i]) return false;
                    }
            return true;
        };
        // LDS-resident slabs (k_sweep_slab): whole hyperplanes of the slowest dimension, as thick as the LDS
        // and one site per thread and level allow (fewer, thicker slabs = a shorter pipeline).  Every outside
        // neighbour of a slab must lie in the plane just below or just above it, and be read only as a NEW
        // value (below, and the first plane through the periodic wrap of the last slab) or only as an OLD one.
        const uint32_t plane = d.N / lat.shape[0];
        d.plane = plane;
        d.pipe_lds = 0;
        std::vector<uint32_t> halo_sites, halo_start;
        auto build_halo = [&](uint32_t slabs) -> bool {
            if (slabs < 2 || lat.shape[0] % slabs != 0 || !admissible(slabs)) return false;
            const uint32_t own = d.N / slabs;
            if (sweep_slab_lds_bytes_for(own, plane, n_levels) > 160 * 1024) return false;
            if (sweep_slab_lds_bytes_for(own, plane, n_levels) / 4 > 65536 || own < 2 * plane) return false; // 16-bit LDS words; a site feeds one slab
            halo_sites.assign(static_cast<size_t>(slabs) * 2 * plane, 0u);
            halo_start.assign(static_cast<size_t>(slabs) * (n_levels + 2), 0u);
            const uint32_t unset = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < slabs; ++k) {
                const uint32_t a = k * own, lo0 = (a + d.N - plane) % d.N, hi0 = (a + own) % d.N;
                if (lo0 == hi0) return false; // the two halo planes must be distinct
                std::vector<uint32_t> need_lo(plane, unset), need_hi(plane, unset), last_lo(plane, 0), last_hi(plane, 0);
                std::vector<uint8_t> old_lo(plane, 0), old_hi(plane, 0);
                for (uint32_t i = a; i < a + own; ++i)
                    for (uint32_t q = 0; q < d.nn; ++q)
                        for (uint32_t j : {lat.fwd[i * d.nn + q], lat.bwd[i * d.nn + q]}) {
                            if (j >= a && j < a + own) continue;
                            const uint32_t rl = (j + d.N - lo0) % d.N, rh = (j + d.N - hi0) % d.N;
                            if (rl < plane) {
                                if (j < i) need_lo[rl] = std::min(need_lo[rl], level[i]), last_lo[rl] = std::max(last_lo[rl], level[i]);
                                else old_lo[rl] = 1;
                            } else if (rh < plane) {
                                if (j < i) need_hi[rh] = std::min(need_hi[rh], level[i]), last_hi[rh] = std::max(last_hi[rh], level[i]);
                                else old_hi[rh] = 1;
                            } else {
                                return false;
                            }
                        }
                std::vector<std::pair<uint32_t, uint32_t>> wanted; // (level that first reads it, site)
                for (uint32_t r = 0; r < plane; ++r) {
                    if ((need_lo[r] != unset && old_lo[r]) || (need_hi[r] != unset && old_hi[r])) return false;
                    // a new value is staged for exactly one level: every reader of it must sit on that level
                    if ((need_lo[r] != unset && last_lo[r] != need_lo[r]) || (need_hi[r] != unset && last_hi[r] != need_hi[r]))
                        return false;
                    if (need_lo[r] != unset) wanted.emplace_back(need_lo[r], lo0 + r);
                    if (need_hi[r] != unset) wanted.emplace_back(need_hi[r], hi0 + r);
                }
                if (k == 0 && !wanted.empty()) return false; // the first slab waits for nobody
                std::sort(wanted.begin(), wanted.end());
                uint32_t *start = &halo_start[static_cast<size_t>(k) * (n_levels + 2)];
                size_t e = 0;
                for (uint32_t l = 0; l <= n_levels + 1; ++l) {
                    while (e < wanted.size() && wanted[e].first < l) ++e;
                    start[l] = static_cast<uint32_t>(e);
                    if (l > 0 && start[l] - start[l - 1] > 128) return false; // staged entries of one level
                }
                for (size_t x = 0; x < wanted.size(); ++x)
                    halo_sites[static_cast<size_t>(k) * 2 * plane + x] = wanted[x].second;
            }
            return true;
        };
        if (!dev_env("PEAPODS_NO_SLAB_LDS") && lat.shape[0] >= 4) {
            uint32_t best = 0;
            for (uint32_t slabs = 2; slabs <= lat.shape[0] / 2 && !best; ++slabs) { // fewest slabs first
                if (lat.shape[0] % slabs != 0) continue;
                uint32_t longest = 0; // sites of one slab at one level
                const uint32_t own = d.N / slabs;
                for (uint32_t l = 0; l < n_levels; ++l) {
                    std::vector<uint32_t> count(slabs, 0u);
                    for (uint32_t pos = level_start[l]; pos < level_start[l + 1]; ++pos) count[order[pos] / own] += 1;
                    for (uint32_t c : count) longest = std::max(longest, c);
                }
                if (longest <= 256 && build_halo(slabs)) best = slabs;
            }
            if (const char *env = dev_env("PEAPODS_SLAB_K")) {
                const long v = std::strtol(env, nullptr, 10);
                best = (v >= 2 && build_halo(static_cast<uint32_t>(v))) ? static_cast<uint32_t>(v) : 0u;
            }
            if (best) {
                build_halo(best);
                K = best;
                d.pipe_lds = 1;
                sim->halo_sites_host = halo_sites;
                sim->halo_start_host = halo_start;
            }
        }
        while (!d.pipe_lds && K > 1 && !admissible(K)) --K;
        if (K > 1) {
            d.pipe_K = K;
            const uint32_t slab = (d.N + K - 1) / K;
            seg.assign(static_cast<size_t>(n_levels) * (K + 1), 0u);
            for (uint32_t l = 0; l < n_levels; ++l) {
                uint32_t pos = level_start[l];
                for (uint32_t k = 0; k <= K; ++k) {
                    // positions of a level are in ascending site order
                    while (k < K && pos < level_start[l + 1] && order[pos] < k * slab) ++pos;
                    seg[static_cast<size_t>(l) * (K + 1) + k] = k == K ? level_start[l + 1] : pos;
                }
            }
        }
    }

    // ---- per-realisation site records in schedule order (neighbours + coupling metadata)
    const size_t rec_stride = (static_cast<size_t>(d.N) + 1) * (d.big ? 2 : 1); // uint4 per realisation
    std::vector<uint4> rec(static_cast<size_t>(d.n_real) * rec_stride);
    // (the wide records serve the wide sweep and, on any lattice of four to six directions they can address, the overlap)
    const bool want_rec_wide = d.wide || (d.nn > 3 && d.nn <= 6 && d.N <= 16383);
    std::vector<uint4> rec_wide(want_rec_wide ? static_cast<size_t>(d.n_real) * (d.N + 1) * 2 : 0);
    std::vector<uint16_t> nbr_site(static_cast<size_t>(d.N) * 12);
    std::vector<uint32_t> jsite(static_cast<size_t>(d.n_real) * d.N), nbr_wide(static_cast<size_t>(d.N) * 16, 0u);
    bool all_nonzero = true;
    std::vector<uint32_t> mpar(static_cast<size_t>(d.n_real) * d.MW, 0u);
    std::vector<int32_t> nz_bonds(d.n_real, 0);
    d.has_odd = 0;
    for (uint32_t real = 0; real < d.n_real; ++real) {
        const float *J = desc->couplings + n_coup * real;
        for (uint32_t pos = 0; pos < d.N; ++pos) {
            const uint32_t site = order[pos];
            uint32_t nb[12];
            uint32_t neg = 0, nz = 0, m = 0;
            for (uint32_t k = 0; k < 6; ++k) {
                nb[2 * k] = k < d.nn ? lat.fwd[site * d.nn + k] : site;
                nb[2 * k + 1] = k < d.nn ? lat.bwd[site * d.nn + k] : site;
                if (k >= d.nn) continue;
                const float jf = J[site * d.nn + k];             // bond site -> fwd, owned by site
                const float jb = J[nb[2 * k + 1] * d.nn + k];    // bond bwd -> site, owned by bwd
                if (jf != 0.0f) { nz |= 1u << (2 * k); m += 1; nz_bonds[real] += 1; }
                if (jf < 0.0f) neg |= 1u << (2 * k);
                if (jb != 0.0f) { nz |= 1u << (2 * k + 1); m += 1; }
                if (jb < 0.0f) neg |= 1u << (2 * k + 1);
            }
            jsite[static_cast<size_t>(real) * d.N + site] = neg | (nz << 12);
            if (real == 0) {
                nbr_wide[static_cast<size_t>(pos) * 16] = site;
                for (int k = 0; k < 12; ++k) {
                    nbr_wide[static_cast<size_t>(pos) * 16 + 1 + k] = nb[k];
                    if (d.N <= 65535) nbr_site[static_cast<size_t>(site) * 12 + k] = static_cast<uint16_t>(nb[k]);
                }
            }
            const uint32_t h = (m + 1) / 2;
            if (m & 1u) { // odd number of non-zero bonds: the odd threshold set (k_rng)
                mpar[static_cast<size_t>(real) * d.MW + (site >> 5)] |= 1u << (site & 31u);
                d.has_odd = 1;
            }
            if (want_rec_wide) { // thirteen LDS byte offsets, h, then the coupling signs and the non-zero flags of the twelve slots
                uint4 a, b;
                a.x = (site * 4) | (nb[0] * 4 << 16);
                a.y = (nb[1] * 4) | (nb[2] * 4 << 16);
                a.z = (nb[3] * 4) | (nb[4] * 4 << 16);
                a.w = (nb[5] * 4) | (nb[6] * 4 << 16);
                b.x = (nb[7] * 4) | (nb[8] * 4 << 16);
                b.y = (nb[9] * 4) | (nb[10] * 4 << 16);
                b.z = (nb[11] * 4) | (h << 16);
                b.w = neg | (nz << 12);
                rec_wide[(static_cast<size_t>(real) * (d.N + 1) + pos) * 2] = a;
                rec_wide[(static_cast<size_t>(real) * (d.N + 1) + pos) * 2 + 1] = b;
            }
            if (m != 2 * d.nn) all_nonzero = false;
            if (d.nn > 3) continue; // the bit-sliced records below hold three directions
            const uint32_t jm = (neg & 63u) | ((nz & 63u) << 6) | (h << 12);
            if (d.big) { // full-width indices, two uint4 per site
                rec[static_cast<size_t>(real) * rec_stride + 2 * pos] = make_uint4(site, nb[0], nb[1], nb[2]);
                rec[static_cast<size_t>(real) * rec_stride + 2 * pos + 1] = make_uint4(nb[3], nb[4], nb[5], jm);
                continue;
            }
            rec[static_cast<size_t>(real) * rec_stride + pos] = // fields hold LDS byte offsets (4 * index)
                make_uint4((site * 4) | (nb[0] * 4 << 16), (nb[1] * 4) | (nb[2] * 4 << 16),
                           (nb[3] * 4) | (nb[4] * 4 << 16), (nb[5] * 4) | (jm << 16));
        }
        if (want_rec_wide) { // idle lanes: every offset at the spare slot N, no bond
            const uint32_t spare = d.N * 4;
            rec_wide[(static_cast<size_t>(real) * (d.N + 1) + d.N) * 2] = make_uint4(spare | (spare << 16), spare | (spare << 16), spare | (spare << 16), spare | (spare << 16));
            rec_wide[(static_cast<size_t>(real) * (d.N + 1) + d.N) * 2 + 1] = make_uint4(spare | (spare << 16), spare | (spare << 16), spare, 0u);
        }
        const uint32_t dummy = d.N * 4; // idle lanes of a pass work on the spare slot N
        if (!d.big)
            rec[static_cast<size_t>(real) * rec_stride + d.N] =
                make_uint4(dummy | (dummy << 16), dummy | (dummy << 16), dummy | (dummy << 16), dummy);
    }
    d.dense_nn = (all_nonzero && d.nn <= 3) ? d.nn : 0;

    // ---- thin levels (SimDims::thin_chunks): when the lattice's levels hold far fewer sites than a wave -- fcc: N / 16^2
    // levels of 16 sites; five directions: levels of one or two -- the wide sweep is run by one wave, a level in chunks
    // of <= 16 sites with each site's twelve bond slots over four lanes.  Worth it when the chunks are fewer than three
    // times the passes (a chunk costs ~0.2 us, a pass of the four-wave loop ~0.6).
    d.thin_chunks = 0;
    d.thin_dense = all_nonzero ? 1 : 0;
    std::vector<uint4> rec_thin;
    if (d.wide && !dev_env("PEAPODS_NO_THIN")) {
        uint32_t chunks = 0;
        for (uint32_t l = 0; l < n_levels; ++l) chunks += (level_start[l + 1] - level_start[l] + 15) / 16;
        if (chunks < 3 * d.n_passes || dev_env("PEAPODS_FORCE_THIN")) {
            d.thin_chunks = chunks;
            rec_thin.resize(static_cast<size_t>(d.n_real) * chunks * 64);
            for (uint32_t real = 0; real < d.n_real; ++real) {
                uint4 *out = rec_thin.data() + static_cast<size_t>(real) * chunks * 64;
                const uint4 *in = rec_wide.data() + static_cast<size_t>(real) * (d.N + 1) * 2;
                for (uint32_t l = 0; l < n_levels; ++l)
                    for (uint32_t b = level_start[l]; b < level_start[l + 1]; b += 16, out += 64)
                        for (uint32_t q = 0; q < 16; ++q) {
                            const uint32_t pos = b + q < level_start[l + 1] ? b + q : d.N; // (N: the idle record)
                            const uint4 a = in[2 * pos], bb = in[2 * pos + 1];
                            const uint32_t off[13] = {a.x & 0xFFFFu, a.x >> 16, a.y & 0xFFFFu, a.y >> 16, a.z & 0xFFFFu, a.z >> 16,
                                                      a.w & 0xFFFFu, a.w >> 16, bb.x & 0xFFFFu, bb.x >> 16, bb.y & 0xFFFFu,
                                                      bb.y >> 16, bb.z & 0xFFFFu};
                            const uint32_t h = bb.z >> 16, neg = bb.w & 0xFFFu, nz = (bb.w >> 12) & 0xFFFu;
                            for (uint32_t j = 0; j < 4; ++j)
                                out[4 * q + j] = make_uint4(off[0] | (off[1 + 3 * j] << 16), off[2 + 3 * j] | (off[3 + 3 * j] << 16),
                                                            ((neg >> (3 * j)) & 7u) | (((nz >> (3 * j)) & 7u) << 3) | (h << 8), 0u);
                        }
            }
        }
    }

    // ---- implicit neighbours (k_sweep_hc): usable when every neighbour slot of every site lies at one of two
    // offsets from the site (regular / through the periodic wrap) -- true for hypercubic lattices of any
    // extents -- and the on-chip kernels apply.  The table replaces the 16-byte site records on the sweep
    // path: 4 bytes per site (site, wrap flags, coupling signs, non-zero flags), staged in the LDS.
    d.hc = 0;
    std::vector<uint32_t> hc_tab;
    if (!d.big && !d.general && d.nn >= 1 && d.nn <= 3 && d.N < (1u << kHcSiteBits) &&
        !dev_env("PEAPODS_NO_HC")) {
        bool ok = true;
        int64_t d0[6] = {0, 0, 0, 0, 0, 0}, d1[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < 2 * d.nn && ok; ++k) {
            std::map<int64_t, uint32_t> seen; // offset -> sites using it
            for (uint32_t i = 0; i < d.N; ++i) {
                const uint32_t j = (k & 1u) ? lat.bwd[i * d.nn + k / 2] : lat.fwd[i * d.nn + k / 2];
                seen[static_cast<int64_t>(j) - static_cast<int64_t>(i)] += 1;
            }
            if (seen.size() > 2) {
                ok = false;
                break;
            }
            auto it = seen.begin();
            d0[k] = d1[k] = it->first;
            if (seen.size() == 2) {
                auto other = std::next(it);
                if (other->second > it->second) std::swap(it, other); // the common offset is the regular one
                d0[k] = it->first;
                d1[k] = other->first;
            }
        }
        // 66 KB at 16^3 leaves room for two workgroups per CU; beyond one workgroup's LDS the records stay
        SimDims probe = d;
        probe.hc = 1;
        if (ok && sweep_hc_lds_bytes(probe) <= 160 * 1024) {
            d.hc = 1;
            for (uint32_t k = 0; k < 6; ++k) {
                d.hc_d0[k] = static_cast<int32_t>(d0[k] * 4);
                d.hc_dd[k] = static_cast<int32_t>((d1[k] - d0[k]) * 4);
            }
            hc_tab.resize(static_cast<size_t>(d.n_real) * d.N);
            for (uint32_t real = 0; real < d.n_real; ++real)
                for (uint32_t pos = 0; pos < d.N; ++pos) {
                    const uint32_t site = order[pos];
                    uint32_t flags = 0;
                    for (uint32_t k = 0; k < 2 * d.nn; ++k) {
                        const uint32_t j = (k & 1u) ? lat.bwd[site * d.nn + k / 2] : lat.fwd[site * d.nn + k / 2];
                        if (static_cast<int64_t>(j) - static_cast<int64_t>(site) != d0[k]) flags |= 1u << k;
                    }
                    const uint32_t js = jsite[static_cast<size_t>(real) * d.N + site]; // neg[0..11] | nz[12..23]: f0,b0,f1,b1,f2,b2
                    hc_tab[static_cast<size_t>(real) * d.N + pos] =
                        site | (flags << kHcFlagShift) | ((js & 63u) << kHcNegShift) | (((js >> 12) & 63u) << kHcNzShift);
                }
        }
    }

    // ---- register-resident sweep (k_sweep_reg, kernels_reg.hip): 16 x 16 x 16 with three directions and every bond
    // non-zero.  Tables: where the random-stream kernel files a site's acceptance planes (reg_pos), and per lane the
    // coupling signs of its 64 sites (reg_cpl).
    d.reg_ok = 0;
    d.planes_perm = 0;
    std::vector<uint32_t> reg_pos, reg_cpl;
    if (d.hc && d.nn == 3 && d.dense_nn == 3 && desc->n_dims == 3 && !desc->neighbor_offsets && desc->shape[0] == 16 &&
        desc->shape[1] == 16 && desc->shape[2] == 16 && dev_env("PEAPODS_REG")) {
        // (opt-in, PEAPODS_DEVELOPER=1 PEAPODS_REG=1: exact, but measured SLOWER than k_sweep_hc -- 48.8 against 24.2 us per
        // launch on the bench ensemble, a step 86 against 58 us: one wave issues an instruction every ~5 cycles however
        // independent they are, so a word's ~10 K instructions take 30 us on one wave, while k_sweep_hc spreads its 14.5 K
        // over eight.  profiles/NOTES.md, round 4.)
        d.reg_ok = 1;
        reg_pos.resize(d.N);
        for (uint32_t site = 0; site < d.N; ++site) reg_pos[site] = reg_slot_of_site(site);
        reg_cpl.assign(static_cast<size_t>(d.n_real) * 12 * 64, 0u);
        for (uint32_t real = 0; real < d.n_real; ++real)
            for (uint32_t site = 0; site < d.N; ++site) {
                const uint32_t x = site >> 8, y = (site >> 4) & 15u, z = site & 15u;
                const uint32_t lane = (z >> 2) * 16 + y, m = z & 3u, j = (x + y + z) & 15u;
                const uint32_t js = jsite[static_cast<size_t>(real) * d.N + site]; // neg bits of f0,b0,f1,b1,f2,b2
                for (uint32_t dir = 0; dir < 3; ++dir)
                    reg_cpl[(static_cast<size_t>(real) * 12 + m * 3 + dir) * 64 + lane] |= ((js >> (2 * dir)) & 3u) << (2 * j);
            }
    }

    // ---- k_sweep_slab streams its inputs in schedule order: per (level, slab) segment, padded to
    // start at a multiple of four entries (16-byte aligned pieces for the LDS-DMA loads)
    d.planes_stride = d.N;
    std::vector<uint2> slab_seg;
    std::vector<uint32_t> pos_of;
    std::vector<uint4> recS;
    std::vector<uint32_t> xportS;
    if (d.general) { // k_rng_f files a site's thresholds at its schedule position
        pos_of.assign(d.N, 0u);
        for (uint32_t pos = 0; pos < d.N; ++pos) pos_of[order[pos]] = pos;
    }
    if (d.pipe_lds) {
        const uint32_t K = d.pipe_K, own = d.N / K, P = d.plane;
        slab_seg.assign(static_cast<size_t>(n_levels) * K, make_uint2(0u, 0u));
        pos_of.assign(d.N, 0u);
        std::vector<uint32_t> padded_of_pos(d.N, 0u);
        uint32_t cursor = 0;
        for (uint32_t l = 0; l < n_levels; ++l)
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t b = seg[static_cast<size_t>(l) * (K + 1) + k], e = seg[static_cast<size_t>(l) * (K + 1) + k + 1];
                cursor = (cursor + 3u) & ~3u;
                slab_seg[static_cast<size_t>(l) * K + k] = make_uint2(cursor, e - b);
                for (uint32_t pos = b; pos < e; ++pos) {
                    padded_of_pos[pos] = cursor + (pos - b);
                    pos_of[order[pos]] = cursor + (pos - b);
                }
                cursor += e - b;
            }
        d.planes_stride = ((cursor + 3u) & ~3u) + 512; // room for the loader's fixed-size pieces past the last segment
        recS.assign(static_cast<size_t>(d.n_real) * d.planes_stride, make_uint4(0u, 0u, 0u, 0u));
        for (uint32_t real = 0; real < d.n_real; ++real)
            for (uint32_t pos = 0; pos < d.N; ++pos) {
                const uint4 r0 = rec[static_cast<size_t>(real) * rec_stride + 2 * pos];
                const uint4 r1 = rec[static_cast<size_t>(real) * rec_stride + 2 * pos + 1];
                const uint32_t site = r0.x, k = site / own, a = k * own;
                const uint32_t lo0 = (a + d.N - P) % d.N, hi0 = (a + own) % d.N;
                auto loc = [&](uint32_t j) -> uint32_t { // LDS word: own range, lower halo plane, upper halo plane
                    if (j >= a && j < a + own) return j - a;
                    const uint32_t rl = (j + d.N - lo0) % d.N;
                    if (rl < P) return own + rl;
                    return own + P + (j + d.N - hi0) % d.N;
                };
                const uint32_t f[8] = {loc(site), loc(r0.y), loc(r0.z), loc(r0.w), loc(r1.x), loc(r1.y), loc(r1.z), r1.w};
                recS[static_cast<size_t>(real) * d.planes_stride + padded_of_pos[pos]] =
                    make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
            }
        // New values from another slab are not read from a halo plane but from the staging ring the
        // exchange records land in: neighbour field = 0x8000 | entry index within the level's group.
        // xportS: where a site's exchange record goes (1 + record index; 0 = not read by another slab).
        xportS.assign(d.planes_stride, 0u);
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t a = k * own;
            const uint32_t *start = &sim->halo_start_host[static_cast<size_t>(k) * (n_levels + 2)];
            const uint32_t n_entries = start[n_levels + 1];
            std::vector<uint32_t> entry_of(d.N, 0xFFFFFFFFu);
            for (uint32_t e = 0; e < n_entries; ++e) {
                const uint32_t j = sim->halo_sites_host[static_cast<size_t>(k) * 2 * P + e];
                entry_of[j] = e;
                uint32_t &target = xportS[pos_of[j]];
                if (target != 0) { // (excluded by own >= 2 planes; kept as a hard check)
                    sim->error = "internal: a site would feed two slabs";
                    return bail(PA_ERR_UNSUPPORTED);
                }
                target = 1 + k * 2 * P + e;
            }
            for (uint32_t i = a; i < a + own; ++i) {
                uint32_t nbs[6];
                for (uint32_t q = 0; q < 3; ++q) {
                    nbs[2 * q] = q < d.nn ? lat.fwd[i * d.nn + q] : i;
                    nbs[2 * q + 1] = q < d.nn ? lat.bwd[i * d.nn + q] : i;
                }
                for (uint32_t real = 0; real < d.n_real; ++real) {
                    uint4 &r = recS[static_cast<size_t>(real) * d.planes_stride + pos_of[i]];
                    uint32_t f[8] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16, r.z & 0xFFFFu, r.z >> 16, r.w & 0xFFFFu, r.w >> 16};
                    for (uint32_t q = 0; q < 6; ++q) {
                        const uint32_t j = nbs[q];
                        if (j < i && (j < a || j >= a + own) && entry_of[j] != 0xFFFFFFFFu)
                            f[1 + q] = slab_stage_word(own, P) + ((level[i] % kSlabStage) * 128 + (entry_of[j] - start[level[i]])) * 4;
                    }
                    r = make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
                }
            }
        }
    }

    // ---- jump tables T^(b*BS), b = 0..B-1
    std::vector<uint32_t> jump(static_cast<size_t>(d.B) * 64 * 16 * 8);
    {
        JumpMatrix step = JumpMatrix::steps(d.BS);
        JumpMatrix acc = JumpMatrix::identity();
        for (uint32_t b = 0; b < d.B; ++b) {
            acc.nibble_table(&jump[static_cast<size_t>(b) * 64 * 16 * 8]);
            if (b + 1 < d.B) acc = step.compose(acc);
        }
    }

    // ---- Joerg bond probability per temperature for a satisfied unit bond (clusters/overlap.rs:435)
    std::vector<float> jorg_p(d.n_temps);
    for (uint32_t t = 0; t < d.n_temps; ++t) jorg_p[t] = 1.0f - expf(-4.0f * 1.0f / sim->temps[t]);

    // ---- PT constants
    std::vector<float> dbeta(d.n_edges, 0.0f);
    for (uint32_t t = 0; t + 1 < d.n_temps; ++t)
        dbeta[t] = 1.0f / sim->temps[t] - 1.0f / sim->temps[t + 1]; // tempering.rs:88
    sim->cold_slot = sim->hot_slot = 0; // realization.rs:92-107
    for (uint32_t t = 1; t < d.n_temps; ++t) {
        if (sim->temps[t] < sim->temps[sim->cold_slot]) sim->cold_slot = t;
        if (sim->temps[t] > sim->temps[sim->hot_slot]) sim->hot_slot = t;
    }

    // ---- device
    if (hipSetDevice(sim->device) != hipSuccess) {
        sim->error = "hipSetDevice failed";
        return bail(PA_ERR_DEVICE);
    }
    if (hipStreamCreate(&sim->stream) != hipSuccess) {
        sim->error = "hipStreamCreate failed";
        return bail(PA_ERR_DEVICE);
    }
    int rc = ensure_ln_table(sim);
    if (rc) return bail(rc);
    if (d.general && (rc = ensure_device_log(sim))) return bail(rc);

    DevPtrs &p = sim->p;
    DeviceBuffers &bufs = sim->buffers;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    const size_t bins = d.n_pairs ? static_cast<size_t>(d.n_real) * d.n_temps * (d.N + 1) : 0;
    const size_t n_pair_slots = static_cast<size_t>(d.n_real) * d.n_pairs * d.n_temps;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    ok(bufs.alloc(&p.S, tile_words * d.n_real));
    if ((d.big && !d.pipe_K) || d.general)
        p.S_next = p.S; // these paths sweep in place
    else
        ok(bufs.alloc(&p.S_next, tile_words * d.n_real));
    if (!pos_of.empty()) ok(bufs.alloc(&sim->pos_of, pos_of.size()));
    if (d.pipe_lds) {
        ok(bufs.alloc(&sim->halo_start, sim->halo_start_host.size()));
        ok(bufs.alloc(&sim->slab_seg, slab_seg.size()));
        ok(bufs.alloc(&sim->recS, recS.size()));
        ok(bufs.alloc(&sim->xportS, xportS.size()));
        ok(bufs.alloc(&p.xchg, static_cast<size_t>(d.n_real) * d.W * d.pipe_K * 2 * d.plane + 256));
    }
    if (d.pipe_K) {
        ok(bufs.alloc(&sim->seg, seg.size()));
        ok(bufs.alloc(&p.progress, static_cast<size_t>(d.n_real) * d.W * d.pipe_K));
        ok(bufs.alloc(&p.slab_error, 4));
    }
    ok(bufs.alloc(&p.swap_epoch, static_cast<size_t>(d.n_real)));

    ok(bufs.alloc(&p.applied_epoch, static_cast<size_t>(d.n_real) * d.W));
    ok(bufs.alloc(&p.planes, (d.general && !d.wide) ? 1 : static_cast<size_t>(d.planes_stride) * d.W * d.n_real));
    if (d.wide) {
        ok(bufs.alloc(&p.planes_hi, static_cast<size_t>(d.planes_stride) * d.W * d.n_real));
        if (d.thin_chunks) ok(bufs.alloc(&sim->rec_thin, rec_thin.size()));
        ok(bufs.alloc(&sim->thr_wide, thr_wide.size()));
    }
    if (!rec_wide.empty()) ok(bufs.alloc(&sim->rec_wide, rec_wide.size()));
    if (d.general) {
        ok(bufs.alloc(&sim->Jf, n_coup * d.n_real));
        ok(bufs.alloc(&sim->temps_dev, static_cast<size_t>(d.n_temps)));
        ok(bufs.alloc(&p.thr_f, tile_words * d.n_real * 32));
        ok(bufs.alloc(&p.xw, tile_words * d.n_real * (d.nn <= 3 ? 4 : 8)));
    }
    ok(bufs.alloc(&p.cursor, static_cast<size_t>(d.n_real) * 8 * d.SP));
    ok(bufs.alloc(&p.cursor_out, static_cast<size_t>(d.n_real) * 8 * d.SP));
    ok(bufs.alloc(&p.system_ids, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.slot_of, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&sim->thr, thr.size()));
    ok(bufs.alloc(&sim->mpar, mpar.size()));
    ok(bufs.alloc(&sim->rec, rec.size()));
    ok(bufs.alloc(&sim->pass_start, pass_start.size()));
    if (d.hc) ok(bufs.alloc(&sim->hc_tab, hc_tab.size()));
    if (d.reg_ok) {
        ok(bufs.alloc(&sim->reg_pos, reg_pos.size()));
        ok(bufs.alloc(&sim->reg_cpl, reg_cpl.size()));
    }
    ok(bufs.alloc(&sim->jump, jump.size()));
    ok(bufs.alloc(&p.Esum, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.Msum, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.Esum_pre, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.Msum_pre, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.Esum_relay, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.Msum_relay, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.relay_swapped, static_cast<size_t>(d.n_real)));
    ok(bufs.alloc(&sim->nz_bonds, nz_bonds.size()));
    ok(bufs.alloc(&p.dot_spin, n_pair_slots));
    ok(bufs.alloc(&p.dot_link, n_pair_slots));
    ok(bufs.alloc(&p.dot_spin_prev, n_pair_slots));
    ok(bufs.alloc(&p.dot_link_prev, n_pair_slots));
    ok(bufs.alloc(&p.system_ids_out, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.slot_of_out, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.mom, static_cast<size_t>(d.n_real) * 5 * d.n_temps));
    ok(bufs.alloc(&p.ovm, static_cast<size_t>(d.n_real) * 6 * d.n_temps));
    ok(bufs.alloc(&p.hist, bins));
    ok(bufs.alloc(&p.qlq, bins));
    ok(bufs.alloc(&p.ql2q, bins));
    ok(bufs.alloc(&p.hist_touched, d.n_pairs ? static_cast<size_t>(d.n_real) * d.n_temps * ((d.N + 32) / 32) : 0));
    ok(bufs.alloc(&p.edge_att, static_cast<size_t>(d.n_real) * d.n_edges));
    ok(bufs.alloc(&p.edge_acc, static_cast<size_t>(d.n_real) * d.n_edges));
    ok(bufs.alloc(&p.round_trips, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&p.trip_state, static_cast<size_t>(d.n_real) * d.n_sys));
    ok(bufs.alloc(&sim->dbeta, dbeta.size()));
    ok(bufs.alloc(&p.swap_masks, static_cast<size_t>(d.n_real) * 2 * d.W));
    ok(bufs.alloc(&p.pair_rng, static_cast<size_t>(d.n_real) * d.n_temps * d.n_pairs * 4));
    ok(bufs.alloc(&sim->nbr_site, nbr_site.size()));
    // the same per-site facts as bitmaps over the sites, for the cluster kernels' word-parallel phases
    std::vector<uint32_t> sw_wrap, sw_cpl;
    if (d.hc) {
        sw_wrap.assign(static_cast<size_t>(3) * d.MW, 0u);
        sw_cpl.assign(static_cast<size_t>(d.n_real) * 6 * d.MW, 0u);
        for (uint32_t real = 0; real < d.n_real; ++real)
            for (uint32_t i = 0; i < d.N; ++i) {
                const uint32_t js = jsite[static_cast<size_t>(real) * d.N + i];
                for (uint32_t dd = 0; dd < 3; ++dd) {
                    if (real == 0 && static_cast<int64_t>(lat.fwd[i * d.nn + (dd < d.nn ? dd : 0)]) - static_cast<int64_t>(i) != d.hc_d0[2 * dd] / 4 && dd < d.nn)
                        sw_wrap[dd * d.MW + (i >> 5)] |= 1u << (i & 31u);
                    if ((js >> (2 * dd)) & 1u) sw_cpl[(static_cast<size_t>(real) * 6 + dd) * d.MW + (i >> 5)] |= 1u << (i & 31u);
                    if ((js >> (12 + 2 * dd)) & 1u) sw_cpl[(static_cast<size_t>(real) * 6 + 3 + dd) * d.MW + (i >> 5)] |= 1u << (i & 31u);
                }
            }
        ok(bufs.alloc(&sim->sw_wrap, sw_wrap.size()));
        ok(bufs.alloc(&sim->sw_cpl, sw_cpl.size()));
    }
    ok(bufs.alloc(&sim->jsite, jsite.size()));
    std::vector<uint4> fwd_big;
    if (d.big) { // forward neighbours by site for the energy / overlap kernels of the large path
        fwd_big.resize(d.N);
        for (uint32_t i = 0; i < d.N; ++i)
            fwd_big[i] = make_uint4(lat.fwd[i * d.nn], d.nn > 1 ? lat.fwd[i * d.nn + 1] : i, d.nn > 2 ? lat.fwd[i * d.nn + 2] : i, 0u);
        ok(bufs.alloc(&sim->fwd_big, fwd_big.size()));
    }
    ok(bufs.alloc(&sim->nbr_wide, nbr_wide.size()));
    ok(bufs.alloc(&sim->jorg_p, jorg_p.size()));
    ok(bufs.alloc(&p.dfs_stack, d.N > 65535 ? 1 : static_cast<size_t>(d.n_real) * std::max(d.n_temps * d.n_pairs, d.n_sys) * d.N));
    if (!d.big && d.n_pairs) ok(bufs.alloc(&p.hd_scratch, 2 * tile_words * d.n_real));
    ok(bufs.alloc(&p.big_counts, static_cast<size_t>(d.n_real) * d.n_sys * 4));
    ok(bufs.alloc(&p.big_ticket, static_cast<size_t>(d.n_real)));
    if (e != hipSuccess) {
        sim->error = std::string("hipMalloc: ") + hipGetErrorString(e);
        return bail(PA_ERR_DEVICE);
    }
    p.thr = sim->thr;
    p.mpar = sim->mpar;
    p.rec = sim->rec;
    p.pass_start = sim->pass_start;
    p.hc_tab = sim->hc_tab;
    p.rec_wide = sim->rec_wide;
    p.rec_thin = sim->rec_thin;
    p.thr_wide = sim->thr_wide;
    if (d.hc && (rc = upload(sim, sim->hc_tab, hc_tab))) return bail(rc);
    p.reg_pos = sim->reg_pos;
    p.reg_cpl = sim->reg_cpl;
    if (d.reg_ok && ((rc = upload(sim, sim->reg_pos, reg_pos)) || (rc = upload(sim, sim->reg_cpl, reg_cpl)))) return bail(rc);
    p.jump = sim->jump;
    p.nz_bonds = sim->nz_bonds;
    p.dbeta = sim->dbeta;
    p.nbr_site = sim->nbr_site;
    p.jsite = sim->jsite;
    p.sw_wrap = sim->sw_wrap;
    p.sw_cpl = sim->sw_cpl;
    p.fwd_big = sim->fwd_big;
    if (d.big && (rc = upload(sim, sim->fwd_big, fwd_big))) return bail(rc);
    p.nbr_wide = sim->nbr_wide;
    p.jorg_p = sim->jorg_p;
    p.seg = sim->seg;
    p.halo_start = sim->halo_start;
    p.slab_seg = sim->slab_seg;
    p.pos_of = sim->pos_of;
    p.recS = sim->recS;
    p.xportS = sim->xportS;
    if (!pos_of.empty() && (rc = upload(sim, sim->pos_of, pos_of))) return bail(rc);
    if (d.pipe_lds) {
        if ((rc = upload(sim, sim->halo_start, sim->halo_start_host))) return bail(rc);
        if ((rc = upload(sim, sim->slab_seg, slab_seg))) return bail(rc);
        if ((rc = upload(sim, sim->recS, recS))) return bail(rc);
        if ((rc = upload(sim, sim->xportS, xportS))) return bail(rc);
        if (hipMemsetAsync(p.xchg, 0, sizeof(uint4) * (static_cast<size_t>(d.n_real) * d.W * d.pipe_K * 2 * d.plane + 256),
                           sim->stream) != hipSuccess) {
            sim->error = "hipMemsetAsync failed";
            return bail(PA_ERR_DEVICE);
        }
    }
    if (d.pipe_K) {
        if ((rc = upload(sim, sim->seg, seg))) return bail(rc);
        if (hipMemsetAsync(p.progress, 0, sizeof(uint32_t) * d.n_real * d.W * d.pipe_K, sim->stream) != hipSuccess ||
            hipMemsetAsync(p.slab_error, 0, sizeof(uint32_t) * 4, sim->stream) != hipSuccess) {
            sim->error = "hipMemsetAsync failed";
            return bail(PA_ERR_DEVICE);
        }
    }
    p.Jf = sim->Jf;
    p.temps = sim->temps_dev;
    if (d.general) {
        const std::vector<float> J(desc->couplings, desc->couplings + n_coup * d.n_real);
        if ((rc = upload(sim, sim->Jf, J)) || (rc = upload(sim, sim->temps_dev, sim->temps))) return bail(rc);
        if (configure_general_kernels() != hipSuccess) {
            sim->error = "kernel LDS configuration failed";
            return bail(PA_ERR_DEVICE);
        }
    }
    if ((rc = upload(sim, sim->thr, thr)) || (rc = upload(sim, sim->mpar, mpar)) ||
        (rc = upload(sim, sim->rec, rec)) || (rc = upload(sim, sim->pass_start, pass_start)) ||
        (!rec_wide.empty() && (rc = upload(sim, sim->rec_wide, rec_wide))) || (d.wide && (rc = upload(sim, sim->thr_wide, thr_wide))) ||
        (d.thin_chunks && (rc = upload(sim, sim->rec_thin, rec_thin))) ||
        (rc = upload(sim, sim->jump, jump)) || (rc = upload(sim, sim->nbr_site, nbr_site)) ||
        (d.hc && ((rc = upload(sim, sim->sw_wrap, sw_wrap)) || (rc = upload(sim, sim->sw_cpl, sw_cpl)))) ||
        (rc = upload(sim, sim->jsite, jsite)) || (rc = upload(sim, sim->nbr_wide, nbr_wide)) || (rc = upload(sim, sim->jorg_p, jorg_p)) ||
        (rc = upload(sim, sim->nz_bonds, nz_bonds)) || (rc = upload(sim, sim->dbeta, dbeta)))
        return bail(rc);
    if (hipMemsetAsync(p.big_counts, 0, sizeof(uint32_t) * d.n_real * d.n_sys * 4, sim->stream) != hipSuccess ||
        hipMemsetAsync(p.big_ticket, 0, sizeof(uint32_t) * d.n_real, sim->stream) != hipSuccess) {
        sim->error = "hipMemsetAsync failed";
        return bail(PA_ERR_DEVICE);
    }
    if (configure_kernels(d) != hipSuccess) {
        sim->error = "kernel LDS configuration failed";
        return bail(PA_ERR_DEVICE);
    }
    if ((rc = make_groups(sim))) return bail(rc);
    if ((rc = zero_accumulators(sim))) return bail(rc);
    if ((rc = seed_state(sim, desc->seed))) return bail(rc);
    *out = sim;
    return PA_OK;
}

void pa_sim_destroy(pa_sim *sim) {
    if (!sim) return;
    if (sim->stream) {
        (void)hipSetDevice(sim->device);
        (void)hipStreamSynchronize(sim->stream);
        for (auto &g : sim->groups)
            if (g.stream) (void)hipStreamSynchronize(g.stream);
        for (auto &ge : sim->pair_graphs)
            if (ge) (void)hipGraphExecDestroy(ge);
        sim->pair_graphs.clear();
        for (auto &g : sim->groups) {
            if (g.stream) (void)hipStreamDestroy(g.stream);
            if (g.done) (void)hipEventDestroy(g.done);
        }
        if (sim->fork) (void)hipEventDestroy(sim->fork);
        for (auto &e : sim->batch_events)
            if (e) (void)hipEventDestroy(e);
        // cluster-statistics buffers are allocated outside DeviceBuffers (they are regrown)
        for (void *extra : {static_cast<void *>(sim->p.cs_fk_csd), static_cast<void *>(sim->p.cs_top4_raw),
                            static_cast<void *>(sim->p.obs_acc), static_cast<void *>(sim->p.cs_ov_csd),
                            static_cast<void *>(sim->p.cs_top4_acc), static_cast<void *>(sim->p.snap_bonds),
                            static_cast<void *>(sim->p.snap_ids), static_cast<void *>(sim->p.snap_blue),
                            static_cast<void *>(sim->p.snap_spins), static_cast<void *>(sim->p.snap_sids)})
            if (extra) (void)hipFree(extra);
        if (sim->cluster_stamps) (void)hipFree(sim->cluster_stamps);
        if (sim->flag_words) (void)hipFree(sim->flag_words);
        for (void *area : sim->uncached_areas) (void)hipFree(area);
        sim->buffers.release();
        (void)hipStreamDestroy(sim->stream);
    }
    delete sim;
}

const char *pa_sim_last_error(const pa_sim *sim) { return sim ? sim->error.c_str() : ""; }

int pa_sim_reset(pa_sim *sim, uint64_t seed) {
    PA_HIP(sim, hipSetDevice(sim->device));
    return seed_state(sim, seed);
}

void pa_sim_request_interrupt(pa_sim *sim) { sim->interrupt.store(true); }

// Waits for a stream: polls for up to 2 ms before handing the thread to the runtime's blocking wait, whose wake-up after
// the last kernel costs 20-30 us (profiles/r03_settle_series.json: a 1-step call 128 -> 107 us, a 20-step call 62.0 ->
// 60.4 us per step).  PEAPODS_SYNC_SPIN_US sets the polling time, 0 turns it off.
static hipError_t wait_for_stream(hipStream_t stream) {
    static const long spin_us = [] {
        const char *env = dev_env("PEAPODS_SYNC_SPIN_US");
        return env ? std::strtol(env, nullptr, 10) : 2000L;
    }();
    if (spin_us > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
        do {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) break; // (idle: the synchronisation below returns at once -- 0.7 us -- and keeps its guarantees)
            if (q != hipErrorNotReady) return q;
        } while (std::chrono::steady_clock::now() < until);
    }
    return hipStreamSynchronize(stream);
}

int pa_sim_synchronize(pa_sim *sim) {
    PA_HIP(sim, hipSetDevice(sim->device));
    // the group streams first: the main stream only holds the join (one wait per group), which is done the moment they are
    for (auto &g : sim->groups) PA_HIP(sim, wait_for_stream(g.stream));
    PA_HIP(sim, wait_for_stream(sim->stream));
    if (sim->d.pipe_K) { // a slab workgroup that gave up waiting leaves the state unusable: say so
        uint32_t gave_up = 0;
        PA_HIP(sim, hipMemcpy(&gave_up, sim->p.slab_error, sizeof(gave_up), hipMemcpyDeviceToHost));
        if (gave_up) return fail(sim, PA_ERR_DEVICE, "a slab of the pipelined sweep gave up waiting for its predecessor");
    }
    if (sim->p.stream_error) {
        uint32_t short_stream = 0;
        PA_HIP(sim, hipMemcpy(&short_stream, sim->p.stream_error, sizeof(short_stream), hipMemcpyDeviceToHost));
        if (short_stream) return fail(sim, PA_ERR_DEVICE, "a cluster move needed more draws than its pre-drawn stream holds");
    }
    return PA_OK;
}

int pa_sim_sample_async(pa_sim *sim, const pa_sample_config *cfg) {
    int code;
    const std::string text = validate_text(cfg, sim->n_replicas, &code);
    if (code != PA_OK) return fail(sim, code, text);
    PA_HIP(sim, hipSetDevice(sim->device));
    sim->interrupt.store(false); // a request belongs to the call it was made during
    return run_sweeps(sim, cfg);
}

int pa_sim_sample(pa_sim *sim, const pa_sample_config *cfg) {
    int rc = pa_sim_sample_async(sim, cfg);
    if (rc) return rc;
    return pa_sim_synchronize(sim);
}

// Device time of EVERY launch of a step of `cfg`'s protocol -- cluster moves included --, by kernel: the launch sequence
// is the one sample() enqueues (run_sweeps itself runs; the handle must have ONE stream group), an event in front of every launch
// (PA_LAUNCH) gives each launch's share of a step, and the same number of steps run again without events gives the
// step time the shares are applied to (events between launches stretch a step by a few per cent).
int pa_sim_profile_step(pa_sim *sim, const pa_sample_config *cfg, uint64_t n_sweeps, char *names_out, uint64_t names_capacity,
                        double *ms_per_step_out, double *launches_per_step_out, uint64_t capacity, uint64_t *n_kernels_out,
                        double *step_ms_out) {
    int code;
    const std::string text = validate_text(cfg, sim->n_replicas, &code);
    if (code != PA_OK) return fail(sim, code, text);
    if (n_sweeps == 0 || cfg->n_sweeps == 0) return fail(sim, PA_ERR_INVALID, "profiling needs at least one sweep");
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = pa_sim_synchronize(sim)) return rc;
    pa_sample_config run = *cfg;
    run.n_sweeps = n_sweeps;
    run.warmup_sweeps = std::min<uint64_t>(cfg->warmup_sweeps, n_sweeps);
    if (sim->groups.size() != 1)
        return fail(sim, PA_ERR_INVALID, "per-launch profiling needs a handle with one stream group (pa_sim_desc.stream_groups = 1): "
                                         "a launch's time beside another group's launches says little about the launch");
    sim->interrupt.store(false);
    g_trace_marks.clear();
    g_trace_error = hipSuccess;
    pa::g_trace_on = true;
    int rc = run_sweeps(sim, &run);
    pa::g_trace_on = false;
    std::vector<TraceMark> marks;
    marks.swap(g_trace_marks);
    // the end of every stream's last launch: an event enqueued behind it, before anything waits
    std::vector<TraceMark> ends;
    for (const auto &m : marks) {
        bool seen = false;
        for (const auto &e : ends) seen = seen || e.stream == m.stream;
        if (seen) continue;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, m.stream) == hipSuccess) ends.push_back({"", m.stream, e});
        else if (g_trace_error == hipSuccess) g_trace_error = hipErrorUnknown;
    }
    auto release = [&] {
        for (auto &m : marks)
            if (m.event) (void)hipEventDestroy(m.event);
        for (auto &m : ends)
            if (m.event) (void)hipEventDestroy(m.event);
    };
    if (rc == PA_OK) rc = pa_sim_synchronize(sim);
    if (rc != PA_OK || g_trace_error != hipSuccess) {
        release();
        return rc != PA_OK ? rc : fail(sim, PA_ERR_DEVICE, std::string("launch tracer: ") + hipGetErrorString(g_trace_error));
    }
    struct Acc { double ms = 0.0; uint64_t launches = 0; };
    std::vector<std::pair<std::string, Acc>> table;
    auto slot_of = [&](const char *raw) -> Acc & {
        std::string name(raw);
        while (!name.empty() && (name.front() == '(' || name.front() == ' ')) name.erase(name.begin());
        while (!name.empty() && (name.back() == ')' || name.back() == ' ')) name.pop_back();
        for (auto &row : table)
            if (row.first == name) return row.second;
        table.push_back({name, Acc{}});
        return table.back().second;
    };
    double traced_ms = 0.0;
    for (size_t i = 0; i < marks.size(); ++i) {
        hipEvent_t until = nullptr;
        for (size_t k = i + 1; k < marks.size() && !until; ++k)
            if (marks[k].stream == marks[i].stream) until = marks[k].event;
        for (size_t k = 0; k < ends.size() && !until; ++k)
            if (ends[k].stream == marks[i].stream) until = ends[k].event;
        float ms = 0.f;
        if (!until || hipEventElapsedTime(&ms, marks[i].event, until) != hipSuccess) continue;
        Acc &a = slot_of(marks[i].name);
        a.ms += ms;
        a.launches += 1;
        traced_ms += ms;
    }
    release();
    // the same steps without events, between two events on the main stream (the run forks from it and joins it)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    PA_HIP(sim, hipEventCreate(&t0));
    PA_HIP(sim, hipEventCreate(&t1));
    PA_HIP(sim, hipEventRecord(t0, sim->stream));
    rc = run_sweeps(sim, &run);
    float plain = 0.f;
    if (rc == PA_OK && hipEventRecord(t1, sim->stream) != hipSuccess) rc = fail(sim, PA_ERR_DEVICE, "hipEventRecord");
    if (rc == PA_OK) rc = pa_sim_synchronize(sim);
    if (rc == PA_OK && hipEventElapsedTime(&plain, t0, t1) != hipSuccess) rc = fail(sim, PA_ERR_DEVICE, "hipEventElapsedTime");
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (rc != PA_OK) return rc;
    const double step_plain = plain / static_cast<double>(n_sweeps);
    const double scale = traced_ms > 0.0 ? (step_plain * static_cast<double>(n_sweeps)) / traced_ms : 1.0;
    std::string names;
    uint64_t n = 0;
    for (const auto &row : table) {
        if (n >= capacity) break;
        if (names.size() + row.first.size() + 2 > names_capacity) break;
        names += row.first;
        names += '\n';
        ms_per_step_out[n] = row.second.ms * scale / static_cast<double>(n_sweeps);
        launches_per_step_out[n] = static_cast<double>(row.second.launches) / static_cast<double>(n_sweeps);
        ++n;
    }
    if (names_capacity) std::snprintf(names_out, names_capacity, "%s", names.c_str());
    if (n_kernels_out) *n_kernels_out = n;
    if (step_ms_out) *step_ms_out = step_plain;
    return PA_OK;
}

// Diagnostic only (not in the public header): shader-clock stamps of one k_sweep workgroup.
int pa_debug_sweep_stamps(pa_sim *sim, unsigned long long out[6]) {
    PA_HIP(sim, hipSetDevice(sim->device));
    unsigned long long *dev = nullptr;
    PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), 6 * sizeof(unsigned long long)));
    SimDims by_site = sim->d;
    by_site.reg_ok = 0; // (the stamped kernels read the acceptance planes by site)
    PA_HIP(sim, launch_rng(by_site, sim->p, false, sim->stream));
    PA_HIP(sim, launch_sweep_stamped(sim->d, sim->p, dev, sim->stream));
    toggle_spin_buffers(sim);
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    PA_HIP(sim, hipMemcpy(out, dev, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return PA_OK;
}

// Diagnostic only: s_memrealtime stamps (100 MHz) of every workgroup of one k_sweep_hc launch, out[n_blocks][8];
// returns the number of workgroups (0 when the handle does not use that kernel).
int pa_debug_sweep_hc_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_blocks) {
    if (!sim->d.hc || sim->d.dense_nn != 3) return 0;
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t n_blocks = static_cast<size_t>(sim->d.n_real) * sim->d.W;
    if (n_blocks > capacity_blocks) return fail(sim, PA_ERR_INVALID, "stamp buffer too small");
    unsigned long long *dev = nullptr;
    PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), n_blocks * 8 * sizeof(unsigned long long)));
    PA_HIP(sim, hipMemset(dev, 0, n_blocks * 8 * sizeof(unsigned long long)));
    SimDims by_site = sim->d;
    by_site.reg_ok = 0; // (the stamped kernels read the acceptance planes by site)
    PA_HIP(sim, launch_rng(by_site, sim->p, false, sim->stream));
    PA_HIP(sim, launch_sweep_stamped(sim->d, sim->p, dev, sim->stream));
    sim->swaps_pending = false;
    toggle_spin_buffers(sim);
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    PA_HIP(sim, hipMemcpy(out, dev, n_blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return static_cast<int>(n_blocks);
}

// Diagnostic only: s_memrealtime stamps of every wave of one k_sweep_reg launch, out[n_blocks][8] (start, spins loaded,
// planes landed, levels 0-15 / 16-31 / 32-47 done, stored, energy done); 0 when the handle does not use that kernel.
int pa_debug_sweep_reg_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_blocks) {
    if (!sim->d.reg_ok) return 0;
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t n_blocks = static_cast<size_t>(sim->d.n_real) * sim->d.W;
    if (n_blocks > capacity_blocks) return fail(sim, PA_ERR_INVALID, "stamp buffer too small");
    unsigned long long *dev = nullptr;
    PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), n_blocks * 8 * sizeof(unsigned long long)));
    PA_HIP(sim, hipMemset(dev, 0, n_blocks * 8 * sizeof(unsigned long long)));
    PA_HIP(sim, launch_rng(sim->d, sim->p, false, sim->stream));
    PA_HIP(sim, launch_sweep_reg(sim->d, sim->p, true, sim->stream, dev));
    sim->swaps_pending = false;
    toggle_spin_buffers(sim);
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    PA_HIP(sim, hipMemcpy(out, dev, n_blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return static_cast<int>(n_blocks);
}

// Diagnostic only: s_memrealtime (100 MHz) at every level barrier of every workgroup of ONE k_sweep_slab launch, after
// `settle` ordinary sweeps; out[workgroups][n_levels + 4] = {first level, last level, stamp per level ...}.
int pa_debug_slab_stamps(pa_sim *sim, unsigned long long *out, uint64_t capacity_words, uint32_t settle) {
    const SimDims &d = sim->d;
    if (!d.pipe_K) return 0;
    PA_HIP(sim, hipSetDevice(sim->device));
    const uint32_t n_tw = d.n_real * d.W, n_tw_pad = (n_tw + 7u) & ~7u;
    const size_t n_blocks = static_cast<size_t>(n_tw_pad) * d.pipe_K, words = n_blocks * (d.n_levels + 4);
    if (words > capacity_words) return fail(sim, PA_ERR_INVALID, "stamp buffer too small");
    unsigned long long *dev = nullptr;
    PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), words * sizeof(unsigned long long)));
    PA_HIP(sim, hipMemset(dev, 0, words * sizeof(unsigned long long)));
    for (uint32_t k = 0; k <= settle; ++k) {
        DevPtrs q = sim->p;
        q.slab_stamps = k == settle ? dev : nullptr;
        PA_HIP(sim, launch_rng(d, q, false, sim->stream));
        PA_HIP(sim, enqueue_sweep(sim, d, q, false, false, sim->stream));
        sim->swaps_pending = false;
        toggle_spin_buffers(sim);
    }
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    PA_HIP(sim, hipMemcpy(out, dev, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return static_cast<int>(n_blocks);
}

// Diagnostic only.  enable != 0: every union-find cluster launch of the handle's sampling calls from now on leaves
// s_memrealtime (100 MHz) stamps of its phases, [workgroup][16] (kernels_cluster.hip: stamp(k)); meant for a handle with one
// stream group (two groups would write the same rows).  enable == 0: copy the rows of the last launch to out, stop stamping;
// returns the number of workgroups.
int pa_debug_cluster_stamps(pa_sim *sim, int enable, unsigned long long *out, uint64_t capacity_blocks) {
    const SimDims &d = sim->d;
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t n_blocks = static_cast<size_t>(d.n_real) * std::max(d.n_temps * std::max(d.n_pairs, 1u), d.n_sys);
    if (enable) {
        if (!sim->cluster_stamps) {
            PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->cluster_stamps), n_blocks * 16 * sizeof(unsigned long long)));
            PA_HIP(sim, hipMemset(sim->cluster_stamps, 0, n_blocks * 16 * sizeof(unsigned long long)));
        }
        return static_cast<int>(n_blocks);
    }
    if (!sim->cluster_stamps) return 0;
    if (capacity_blocks < n_blocks) return fail(sim, PA_ERR_INVALID, "stamp buffer too small");
    PA_HIP(sim, hipDeviceSynchronize());
    PA_HIP(sim, hipMemcpy(out, sim->cluster_stamps, n_blocks * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(sim->cluster_stamps);
    sim->cluster_stamps = nullptr;
    return static_cast<int>(n_blocks);
}

int pa_debug_measure_stamps(pa_sim *sim, unsigned long long out[7]) {
    PA_HIP(sim, hipSetDevice(sim->device));
    unsigned long long *dev = nullptr;
    PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&dev), 7 * sizeof(unsigned long long)));
    PA_HIP(sim, hipMemset(dev, 0, 7 * sizeof(unsigned long long)));
    PA_HIP(sim, launch_rng(sim->d, sim->p, false, sim->stream));
    PA_HIP(sim, launch_sweep(sim->d, sim->p, true, true, false, sim->stream));
    toggle_spin_buffers(sim);
    PA_HIP(sim, overlap_launch(sim->d, sim->p, sim->stream));
    PtParams pt{0, sim->next_parity, sim->hot_slot, sim->cold_slot};
    PA_HIP(sim, launch_measure_stamped(sim->d, sim->p, pt, dev, sim->stream));
    sim->swaps_pending = true;
    if (int rc = flush_swaps(sim)) return rc;
    PA_HIP(sim, hipStreamSynchronize(sim->stream));
    PA_HIP(sim, hipMemcpy(out, dev, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return PA_OK;
}

static int sweep_only(pa_sim *sim, uint64_t n_sweeps, bool gibbs) {
    PA_HIP(sim, hipSetDevice(sim->device));
    if (gibbs)
        if (int rc = ensure_gibbs(sim)) return rc;
    // (experiment behind the developer gate: one flagged launch per sweep, see k_stream_sweep_flagged)
    const bool flagged = !gibbs && dev_env("PEAPODS_FLAGGED") && flagged_sweep_applies(sim->d) && !sim->swaps_pending;
    if (flagged && !sim->flag_words) {
        const uint32_t groups = ((sim->d.n_real * sim->d.W + 1) / 2 + 7) / 8;
        PA_HIP(sim, hipMalloc(reinterpret_cast<void **>(&sim->flag_words), (groups + 1) * sizeof(uint32_t)));
        PA_HIP(sim, hipMemset(sim->flag_words, 0, (groups + 1) * sizeof(uint32_t)));
        sim->flag_groups = groups;
    }
    for (uint64_t k = 0; k < n_sweeps; ++k) {
        if (flagged) {
            PA_HIP(sim, launch_stream_sweep_flagged(sim->d, sim->p, sim->flag_words, ++sim->flag_epoch,
                                                    sim->flag_words + sim->flag_groups, sim->stream));
        } else {
            PA_HIP(sim, launch_rng(sim->d, sim->p, gibbs, sim->stream));
            PA_HIP(sim, enqueue_sweep(sim, sim->d, sim->p, false, gibbs, sim->stream));
        }
        sim->swaps_pending = false;
        toggle_spin_buffers(sim);
    }
    if (int rc = pa_sim_synchronize(sim)) return rc;
    if (flagged) {
        uint32_t timed_out = 0;
        PA_HIP(sim, hipMemcpy(&timed_out, sim->flag_words + sim->flag_groups, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (timed_out) return fail(sim, PA_ERR_DEVICE, "flagged sweep: a release counter never arrived");
    }
    return PA_OK;
}

int pa_sim_metropolis_sweep(pa_sim *sim, uint64_t n_sweeps) { return sweep_only(sim, n_sweeps, false); }
int pa_sim_gibbs_sweep(pa_sim *sim, uint64_t n_sweeps) { return sweep_only(sim, n_sweeps, true); }

int pa_sim_compute_energies(pa_sim *sim) {
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    PA_HIP(sim, launch_sweep(sim->d, sim->p, true, false, false, sim->stream));
    return pa_sim_synchronize(sim);
}

int pa_sim_parallel_tempering(pa_sim *sim, int32_t schedule) {

    if (schedule != PA_PT_SINGLE_RANDOM_EDGE && schedule != PA_PT_FULL_LADDER)
        return fail(sim, PA_ERR_INVALID,
                    "unknown pt_schedule, expected 'single_random_edge' or 'full_ladder'");
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    PA_HIP(sim, launch_sweep(sim->d, sim->p, true, false, false, sim->stream));
    PtParams pt{schedule, sim->next_parity, sim->hot_slot, sim->cold_slot};
    PA_HIP(sim, measure_launch(sim->fresh, sim->d, sim->p, pt, false, false, true, sim->stream));
    if (schedule == PA_PT_FULL_LADDER) sim->next_parity = 1 - sim->next_parity;
    sim->swaps_pending = true;
    if (int rc = flush_swaps(sim)) return rc;
    return pa_sim_synchronize(sim);
}

int pa_sim_overlap_dots(pa_sim *sim, int64_t *dot_spin, int64_t *dot_link) {
    const SimDims &d = sim->d;
    if (d.n_pairs == 0) return fail(sim, PA_ERR_INVALID, "overlap needs n_replicas >= 2");
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    PA_HIP(sim, overlap_launch(d, sim->p, sim->stream));
    const size_t count = static_cast<size_t>(d.n_real) * d.n_pairs * d.n_temps;
    std::vector<int32_t> a, b;
    int rc;
    if ((rc = download(sim, a, sim->p.dot_spin_prev, count)) || (rc = download(sim, b, sim->p.dot_link_prev, count)))
        return rc;
    for (size_t i = 0; i < count; ++i) {
        dot_spin[i] = a[i];
        dot_link[i] = b[i];
    }
    return PA_OK;
}

uint64_t pa_sim_n_spins(const pa_sim *sim) { return sim->d.N; }
uint64_t pa_sim_n_neighbors(const pa_sim *sim) { return sim->d.nn; }
uint64_t pa_sim_n_systems(const pa_sim *sim) { return sim->d.n_sys; }
uint64_t pa_sim_n_realizations(const pa_sim *sim) { return sim->d.n_real; }
uint64_t pa_sim_sweeps_completed(const pa_sim *sim) { return sim->n_sweeps_run; }

uint64_t pa_abi_layout(uint64_t *out, uint64_t capacity) {
    std::vector<uint64_t> v;
#define PA_FIELD(type, field) v.push_back(offsetof(type, field))
    v.push_back(sizeof(pa_sim_desc));
    v.push_back(13);
    PA_FIELD(pa_sim_desc, n_dims);
    PA_FIELD(pa_sim_desc, shape);
    PA_FIELD(pa_sim_desc, n_neighbors);
    PA_FIELD(pa_sim_desc, neighbor_offsets);
    PA_FIELD(pa_sim_desc, couplings);
    PA_FIELD(pa_sim_desc, n_realizations);
    PA_FIELD(pa_sim_desc, temperatures);
    PA_FIELD(pa_sim_desc, n_temps);
    PA_FIELD(pa_sim_desc, n_replicas);
    PA_FIELD(pa_sim_desc, seed);
    PA_FIELD(pa_sim_desc, device);
    PA_FIELD(pa_sim_desc, first_realization);
    PA_FIELD(pa_sim_desc, stream_groups);
    v.push_back(sizeof(pa_sample_config));
    v.push_back(19);
    PA_FIELD(pa_sample_config, n_sweeps);
    PA_FIELD(pa_sample_config, warmup_sweeps);
    PA_FIELD(pa_sample_config, sweep_mode);
    PA_FIELD(pa_sample_config, pt_interval);
    PA_FIELD(pa_sample_config, pt_schedule);
    PA_FIELD(pa_sample_config, overlap_cluster_interval);
    PA_FIELD(pa_sample_config, overlap_cluster_modes);
    PA_FIELD(pa_sample_config, n_overlap_cluster_modes);
    PA_FIELD(pa_sample_config, overlap_cluster_mode);
    PA_FIELD(pa_sample_config, measure_overlap);
    PA_FIELD(pa_sample_config, cluster_interval);
    PA_FIELD(pa_sample_config, cluster_mode);
    PA_FIELD(pa_sample_config, equilibration_diagnostic);
    PA_FIELD(pa_sample_config, autocorrelation_max_lag);
    PA_FIELD(pa_sample_config, collect_cluster_stats);
    PA_FIELD(pa_sample_config, cluster_action);
    PA_FIELD(pa_sample_config, overlap_cluster_action);
    PA_FIELD(pa_sample_config, autocorrelation_backend);
    PA_FIELD(pa_sample_config, snapshot_interval);
#undef PA_FIELD
    for (uint64_t k = 0; k < v.size() && k < capacity; ++k) out[k] = v[k];
    return v.size();
}

#define PA_CHECK_REAL(sim, r)                                                                      \
    if ((r) >= (sim)->d.n_real) return fail(sim, PA_ERR_INVALID, "realization index out of range")

int pa_sim_get_spins(pa_sim *sim, uint64_t realization, int8_t *out) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    const SimDims &d = sim->d;
    std::vector<uint32_t> S, slot_of;
    int rc;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    if ((rc = download(sim, S, sim->p.S + tile_words * realization, tile_words))) return rc;
    if ((rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys))) return rc;
    for (uint32_t sys = 0; sys < d.n_sys; ++sys) {
        const uint32_t slot = slot_of[sys], word = slot >> 5, bit = slot & 31u;
        int8_t *dst = out + static_cast<size_t>(sys) * d.N;
        for (uint32_t site = 0; site < d.N; ++site)
            dst[site] = ((S[static_cast<size_t>(word) * d.N + site] >> bit) & 1u) ? -1 : 1;
    }
    return PA_OK;
}

int pa_sim_get_system_ids(pa_sim *sim, uint64_t realization, uint64_t *out) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    std::vector<uint32_t> ids;
    int rc = download(sim, ids, sim->p.system_ids + sim->d.n_sys * realization, sim->d.n_sys);
    if (rc) return rc;
    for (uint32_t k = 0; k < sim->d.n_sys; ++k) out[k] = ids[k];
    return PA_OK;
}

int pa_sim_get_energies(pa_sim *sim, uint64_t realization, float *out) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<int32_t> E;
    std::vector<uint32_t> slot_of;
    int rc;
    if ((rc = download(sim, E, sim->p.Esum + d.n_sys * realization, d.n_sys))) return rc;
    if ((rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys))) return rc;
    for (uint32_t sys = 0; sys < d.n_sys; ++sys) { // energy.rs:108: total / n_spins as f32
        const int32_t raw = E[slot_of[sys]];
        float total = static_cast<float>(raw);
        if (d.general) std::memcpy(&total, &raw, sizeof(float)); // raw bits of the f32 sum
        out[sys] = total / static_cast<float>(d.N);
    }
    return PA_OK;
}

int pa_sim_get_magnetization_sums(pa_sim *sim, uint64_t realization, int64_t *out) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<int32_t> M;
    std::vector<uint32_t> slot_of;
    int rc;
    if ((rc = download(sim, M, sim->p.Msum + d.n_sys * realization, d.n_sys))) return rc;
    if ((rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys))) return rc;
    for (uint32_t sys = 0; sys < d.n_sys; ++sys) out[sys] = M[slot_of[sys]];
    return PA_OK;
}

int pa_sim_get_rng_state(pa_sim *sim, uint64_t realization, uint64_t system, uint64_t out[4]) {
    PA_CHECK_REAL(sim, realization);
    if (system >= sim->d.n_sys) return fail(sim, PA_ERR_INVALID, "system index out of range");
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<uint32_t> slot_of, cur;
    int rc;
    if ((rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys))) return rc;
    if ((rc = download(sim, cur, sim->p.cursor + static_cast<size_t>(realization) * 8 * d.SP,
                       static_cast<size_t>(8) * d.SP)))
        return rc;
    const uint32_t slot = slot_of[system];
    for (int w = 0; w < 4; ++w)
        out[w] = (static_cast<uint64_t>(cur[(2 * w + 1) * d.SP + slot]) << 32) | cur[(2 * w) * d.SP + slot];
    return PA_OK;
}

// ---- state setters: the inner boundary in the other direction ------------------------------------------------
// The reference's kernels run on slices a caller owns (mcmc/sweep.rs:220-229: spins, system_ids and rngs of a
// Realization are plain borrowed buffers), so a caller may arrange them freely before a sweep -- its own test of the
// permuted case does (sweep.rs:382-442: system_ids = [2, 0, 1]).  These three calls give a handle the same freedom.
// All validate before touching anything.

int pa_sim_set_spins(pa_sim *sim, uint64_t realization, const int8_t *spins) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    for (size_t i = 0; i < static_cast<size_t>(d.n_sys) * d.N; ++i)
        if (spins[i] != 1 && spins[i] != -1) return fail(sim, PA_ERR_INVALID, "spins must be +1 or -1");
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    std::vector<uint32_t> S, slot_of;
    int rc;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W;
    if ((rc = download(sim, S, sim->p.S + tile_words * realization, tile_words))) return rc;
    if ((rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys))) return rc;
    for (uint32_t sys = 0; sys < d.n_sys; ++sys) { // system-major in, bit column of the system's slot out
        const uint32_t slot = slot_of[sys], bit = 1u << (slot & 31u);
        uint32_t *word = S.data() + static_cast<size_t>(slot >> 5) * d.N;
        const int8_t *src = spins + static_cast<size_t>(sys) * d.N;
        for (uint32_t site = 0; site < d.N; ++site) word[site] = src[site] < 0 ? (word[site] | bit) : (word[site] & ~bit);
    }
    if ((rc = upload(sim, sim->p.S + tile_words * realization, S))) return rc;
    // the cached energies and magnetisation sums follow the new state (Realization::new does the same, realization.rs:196-197)
    PA_HIP(sim, launch_sweep(sim->d, sim->p, true, false, false, sim->stream));
    return pa_sim_synchronize(sim);
}

int pa_sim_set_system_ids(pa_sim *sim, uint64_t realization, const uint64_t *system_ids) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    std::vector<uint32_t> new_slot_of(d.n_sys, 0xFFFFFFFFu);
    for (uint32_t slot = 0; slot < d.n_sys; ++slot) {
        const uint64_t sys = system_ids[slot];
        if (sys >= d.n_sys || new_slot_of[sys] != 0xFFFFFFFFu)
            return fail(sim, PA_ERR_INVALID, "system_ids must be a permutation of 0..n_systems");
        // a replica's ladder keeps its own systems (tempering.rs exchanges neighbours of one ladder only)
        if (sys / d.n_temps != slot / d.n_temps)
            return fail(sim, PA_ERR_INVALID, "system_ids must permute the systems of each replica among that replica's slots");
        new_slot_of[sys] = slot;
    }
    PA_HIP(sim, hipSetDevice(sim->device));
    if (int rc = flush_swaps(sim)) return rc;
    // spins, generators and cached sums belong to systems: every system's bit column, cursor and sums move to its new slot
    std::vector<uint32_t> S, old_slot_of, cur;
    std::vector<int32_t> E, M;
    int rc;
    const size_t tile_words = static_cast<size_t>(d.N) * d.W, cur_words = static_cast<size_t>(8) * d.SP;
    if ((rc = download(sim, S, sim->p.S + tile_words * realization, tile_words)) ||
        (rc = download(sim, old_slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys)) ||
        (rc = download(sim, cur, sim->p.cursor + cur_words * realization, cur_words)) ||
        (rc = download(sim, E, sim->p.Esum + d.n_sys * realization, d.n_sys)) ||
        (rc = download(sim, M, sim->p.Msum + d.n_sys * realization, d.n_sys)))
        return rc;
    std::vector<uint32_t> S2(S), cur2(cur), ids(d.n_sys);
    std::vector<int32_t> E2(E), M2(M);
    for (uint32_t sys = 0; sys < d.n_sys; ++sys) {
        const uint32_t from = old_slot_of[sys], to = new_slot_of[sys];
        ids[to] = sys;
        E2[to] = E[from];
        M2[to] = M[from];
        for (int w = 0; w < 8; ++w) cur2[static_cast<size_t>(w) * d.SP + to] = cur[static_cast<size_t>(w) * d.SP + from];
        const uint32_t *src = S.data() + static_cast<size_t>(from >> 5) * d.N;
        uint32_t *dst = S2.data() + static_cast<size_t>(to >> 5) * d.N;
        const uint32_t sb = from & 31u, bit = 1u << (to & 31u);
        for (uint32_t site = 0; site < d.N; ++site) dst[site] = ((src[site] >> sb) & 1u) ? (dst[site] | bit) : (dst[site] & ~bit);
    }
    if ((rc = upload(sim, sim->p.S + tile_words * realization, S2)) ||
        (rc = upload(sim, sim->p.cursor + cur_words * realization, cur2)) ||
        (rc = upload(sim, sim->p.Esum + d.n_sys * realization, E2)) ||
        (rc = upload(sim, sim->p.Msum + d.n_sys * realization, M2)) ||
        (rc = upload(sim, sim->p.system_ids + d.n_sys * realization, ids)) ||
        (rc = upload(sim, sim->p.slot_of + d.n_sys * realization, new_slot_of)))
        return rc;
    return PA_OK;
}

int pa_sim_set_rng_state(pa_sim *sim, uint64_t realization, uint64_t system, const uint64_t state[4]) {
    PA_CHECK_REAL(sim, realization);
    if (system >= sim->d.n_sys) return fail(sim, PA_ERR_INVALID, "system index out of range");
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<uint32_t> slot_of;
    if (int rc = download(sim, slot_of, sim->p.slot_of + d.n_sys * realization, d.n_sys)) return rc;
    const uint32_t slot = slot_of[system];
    uint32_t *cur = sim->p.cursor + static_cast<size_t>(realization) * 8 * d.SP;
    for (int w = 0; w < 4; ++w) {
        const uint32_t lo = static_cast<uint32_t>(state[w]), hi = static_cast<uint32_t>(state[w] >> 32);
        PA_HIP(sim, hipMemcpyAsync(cur + static_cast<size_t>(2 * w) * d.SP + slot, &lo, 4, hipMemcpyHostToDevice, sim->stream));
        PA_HIP(sim, hipMemcpyAsync(cur + static_cast<size_t>(2 * w + 1) * d.SP + slot, &hi, 4, hipMemcpyHostToDevice, sim->stream));
        PA_HIP(sim, hipStreamSynchronize(sim->stream)); // (lo / hi live on this frame)
    }
    return PA_OK;
}

int pa_sim_get_pt_counters(pa_sim *sim, uint64_t realization, uint64_t *edge_attempts,
                           uint64_t *edge_acceptances, uint64_t *round_trips) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<uint64_t> a, b, c;
    int rc;
    if ((rc = download(sim, a, sim->p.edge_att + d.n_edges * realization, d.n_edges)) ||
        (rc = download(sim, b, sim->p.edge_acc + d.n_edges * realization, d.n_edges)) ||
        (rc = download(sim, c, sim->p.round_trips + d.n_sys * realization, d.n_sys)))
        return rc;
    for (uint32_t k = 0; k + 1 < d.n_temps; ++k) {
        edge_attempts[k] = a[k];
        edge_acceptances[k] = b[k];
    }
    std::memcpy(round_trips, c.data(), sizeof(uint64_t) * d.n_sys);
    return PA_OK;
}

// Statistics::average (stats.rs:29-35): sum / count, or the raw sums when nothing was recorded.
static void average_into(double *dst, const double *sums, size_t n, uint64_t count) {
    if (count == 0) {
        std::memcpy(dst, sums, sizeof(double) * n);
        return;
    }
    const double c = static_cast<double>(count);
    for (size_t i = 0; i < n; ++i) dst[i] = sums[i] / c;
}

int pa_sim_get_result_moments(pa_sim *sim, uint64_t realization, double *out) {
    PA_CHECK_REAL(sim, realization);
    PA_HIP(sim, hipSetDevice(sim->device));
    const SimDims &d = sim->d;
    std::vector<double> sums;
    int rc = download(sim, sums, sim->p.mom + static_cast<size_t>(realization) * 5 * d.n_temps,
                      static_cast<size_t>(5) * d.n_temps);
    if (rc) return rc;
    if (nothing_accumulated(sim, 1u)) std::fill(sums.begin(), sums.end(), 0.0); // the sums were never started
    average_into(out, sums.data(), sums.size(), sim->n_recorded * d.n_replicas);
    return PA_OK;
}

int pa_sim_get_result_overlap(pa_sim *sim, uint64_t realization, double *out) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    if (d.n_pairs == 0) return fail(sim, PA_ERR_INVALID, "overlap needs n_replicas >= 2");
    PA_HIP(sim, hipSetDevice(sim->device));
    std::vector<double> sums;
    int rc = download(sim, sums, sim->p.ovm + static_cast<size_t>(realization) * 6 * d.n_temps,
                      static_cast<size_t>(6) * d.n_temps);
    if (rc) return rc;
    if (nothing_accumulated(sim, 2u)) std::fill(sums.begin(), sums.end(), 0.0);
    average_into(out, sums.data(), sums.size(),
                 sim->overlap_collected ? sim->n_recorded * d.n_pairs : 0);
    return PA_OK;
}

int pa_sim_get_result_histograms(pa_sim *sim, uint64_t realization, uint64_t *histogram,
                                 double *ql_at_q_sum, double *ql2_at_q_sum) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    if (d.n_pairs == 0) return fail(sim, PA_ERR_INVALID, "overlap needs n_replicas >= 2");
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t bins = static_cast<size_t>(d.n_temps) * (d.N + 1);
    if (nothing_accumulated(sim, 2u)) { // no recorded overlap since the accumulators were reset
        std::fill(histogram, histogram + bins, uint64_t(0));
        std::fill(ql_at_q_sum, ql_at_q_sum + bins, 0.0);
        std::fill(ql2_at_q_sum, ql2_at_q_sum + bins, 0.0);
        return PA_OK;
    }
    PA_HIP(sim, hipMemcpyAsync(histogram, sim->p.hist + bins * realization, bins * sizeof(uint64_t),
                               hipMemcpyDeviceToHost, sim->stream));
    PA_HIP(sim, hipMemcpyAsync(ql_at_q_sum, sim->p.qlq + bins * realization, bins * sizeof(double),
                               hipMemcpyDeviceToHost, sim->stream));
    PA_HIP(sim, hipMemcpyAsync(ql2_at_q_sum, sim->p.ql2q + bins * realization, bins * sizeof(double),
                               hipMemcpyDeviceToHost, sim->stream));
    // bins no sweep of the last run reached hold whatever an earlier run left: they read as zero
    const size_t row_words = (d.N + 32) / 32;
    std::vector<uint32_t> touched;
    if (int rc = download(sim, touched, sim->p.hist_touched + row_words * d.n_temps * realization, row_words * d.n_temps))
        return rc;
    for (uint32_t t = 0; t < d.n_temps; ++t)
        for (uint32_t b = 0; b <= d.N; ++b)
            if (!((touched[t * row_words + (b >> 5)] >> (b & 31u)) & 1u)) {
                const size_t at = static_cast<size_t>(t) * (d.N + 1) + b;
                histogram[at] = 0;
                ql_at_q_sum[at] = 0.0;
                ql2_at_q_sum[at] = 0.0;
            }
    return PA_OK;
}

int pa_sim_get_autocorrelation_tau(pa_sim *sim, uint64_t realization, double *mags2_tau, double *overlap2_tau) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    if (!sim->ac_lag) return fail(sim, PA_ERR_INVALID, "the last sample() did not collect autocorrelations");
    PA_HIP(sim, hipSetDevice(sim->device));
    std::vector<float> series;
    const size_t n = static_cast<size_t>(sim->ac_samples) * d.n_temps;
    const size_t base = static_cast<size_t>(realization) * sim->ac_capacity * d.n_temps;
    int rc = download(sim, series, sim->p.ac_m2 + base, n);
    if (rc) return rc;
    autocorrelation_tau(series.data(), sim->ac_samples, d.n_temps, sim->ac_lag, mags2_tau, sim->ac_fft);
    if (d.n_pairs) {
        if ((rc = download(sim, series, sim->p.ac_q2 + base, n))) return rc;
        autocorrelation_tau(series.data(), sim->ac_samples, d.n_temps, sim->ac_lag, overlap2_tau, sim->ac_fft);
    }
    return PA_OK;
}

int pa_sim_get_cluster_stats(pa_sim *sim, uint64_t realization, uint64_t *fk_csd, uint64_t *overlap_csd,
                             double *top_sizes) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    if (!sim->cs_fk && !sim->cs_overlap)
        return fail(sim, PA_ERR_INVALID, "the last sample() did not collect cluster statistics");
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t bins = static_cast<size_t>(d.n_temps) * (d.N + 1);
    if (sim->cs_fk && fk_csd)
        PA_HIP(sim, hipMemcpy(fk_csd, sim->p.cs_fk_csd + bins * realization, sizeof(uint64_t) * bins, hipMemcpyDeviceToHost));
    if (sim->cs_overlap) {
        const uint32_t modes = sim->cs_n_modes;
        if (overlap_csd)
            PA_HIP(sim, hipMemcpy(overlap_csd, sim->p.cs_ov_csd + bins * sim->cs_modes * realization,
                                  sizeof(uint64_t) * bins * modes, hipMemcpyDeviceToHost));
        if (top_sizes) { // mod.rs:799-823: sums over the recorded calls of a mode / (calls * n_pairs)
            std::vector<double> acc;
            int rc = download(sim, acc, sim->p.cs_top4_acc + static_cast<size_t>(4) * d.n_temps * sim->cs_modes * realization,
                              static_cast<size_t>(4) * d.n_temps * modes);
            if (rc) return rc;
            for (uint32_t m = 0; m < modes; ++m) {
                const double denom = static_cast<double>(sim->top4_calls[m] * d.n_pairs);
                for (size_t k = 0; k < static_cast<size_t>(4) * d.n_temps; ++k)
                    top_sizes[m * 4 * d.n_temps + k] = sim->top4_calls[m] ? acc[m * 4 * d.n_temps + k] / denom : 0.0;
            }
        }
    }
    return (sim->cs_fk ? 1 : 0) | (sim->cs_overlap ? 2 : 0);
}

int pa_sim_get_observation(pa_sim *sim, uint64_t realization, int32_t kind, uint64_t *count, uint64_t *sizes,
                           double *top4_frac, double *bond_density, double *large_mean, double *winding) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    if (kind < 0 || kind > 3) return fail(sim, PA_ERR_INVALID, "observation kind out of range");
    if (!sim->obs_kind[kind]) return 0;
    PA_HIP(sim, hipSetDevice(sim->device));
    const size_t row = kObsWords + d.N + 1;
    std::vector<uint64_t> acc;
    int rc = download(sim, acc, sim->p.obs_acc + (static_cast<size_t>(realization) * 4 + kind) * d.n_temps * row,
                      static_cast<size_t>(d.n_temps) * row);
    if (rc) return rc;
    bool any = false;
    for (uint32_t t = 0; t < d.n_temps; ++t) any = any || acc[t * row] != 0;
    if (!any) return 0; // GraphObservationAccum::finish, mod.rs:63-65
    for (uint32_t t = 0; t < d.n_temps; ++t) { // mod.rs:67-97
        const uint64_t *a = &acc[t * row];
        count[t] = a[0];
        std::memcpy(sizes + static_cast<size_t>(t) * (d.N + 1), a + kObsWords, sizeof(uint64_t) * (d.N + 1));
        for (int k = 0; k < 4; ++k) top4_frac[t * 4 + k] = winding[t * 4 + k] = 0.0;
        bond_density[t] = large_mean[t] = 0.0;
        if (a[0] == 0) continue;
        const double c = static_cast<double>(a[0]);
        for (int k = 0; k < 4; ++k) {
            top4_frac[t * 4 + k] = static_cast<double>(a[1 + k]) / (c * static_cast<double>(d.N));
            if (sim->square) winding[t * 4 + k] = static_cast<double>(a[6 + k]) / c;
        }
        bond_density[t] = static_cast<double>(a[5]) / (c * static_cast<double>(d.N) * static_cast<double>(d.nn));
        large_mean[t] = static_cast<double>(a[10]) / c;
    }
    return sim->square ? 3 : 1;
}

uint64_t pa_sim_n_snapshots(const pa_sim *sim) { return sim->snapshots.size(); }

int pa_sim_get_snapshot(pa_sim *sim, uint64_t index, uint64_t *sweep_id, int32_t *mode_idx, uint32_t *cluster_ids,
                        uint32_t *blue_ids, int8_t *spins, uint64_t *system_ids) {
    if (!sim) return PA_ERR_INVALID;
    if (index >= sim->snapshots.size()) return fail(sim, PA_ERR_INVALID, "snapshot index out of range");
    const pa_sim::Snapshot &snap = sim->snapshots[index];
    *sweep_id = snap.sweep_id;
    *mode_idx = snap.mode_idx;
    std::memcpy(cluster_ids, snap.ids.data(), sizeof(uint32_t) * snap.ids.size());
    if (snap.has_blue && blue_ids) std::memcpy(blue_ids, snap.blue.data(), sizeof(uint32_t) * snap.blue.size());
    std::memcpy(spins, snap.spins.data(), snap.spins.size());
    std::memcpy(system_ids, snap.sids.data(), sizeof(uint64_t) * snap.sids.size());
    return snap.has_blue ? 1 : 0;
}

uint64_t pa_sim_n_equil_checkpoints(const pa_sim *sim) { return sim->equil_sweeps.size(); }

int pa_sim_get_equilibration(pa_sim *sim, uint64_t realization, uint64_t *sweeps, double *energy_avg,
                             double *link_overlap_avg) {
    PA_CHECK_REAL(sim, realization);
    const SimDims &d = sim->d;
    const size_t n_ck = sim->equil_sweeps.size();
    if (!n_ck) return fail(sim, PA_ERR_INVALID, "the last sample() did not run the equilibration diagnostic");
    PA_HIP(sim, hipSetDevice(sim->device));
    std::vector<double> ck;
    int rc = download(sim, ck, sim->p.eq_ckpt + static_cast<size_t>(realization) * 64 * 2 * d.n_temps,
                      n_ck * 2 * d.n_temps);
    if (rc) return rc;
    for (size_t k = 0; k < n_ck; ++k) {
        sweeps[k] = sim->equil_sweeps[k];
        std::memcpy(energy_avg + k * d.n_temps, &ck[(k * 2 + 0) * d.n_temps], sizeof(double) * d.n_temps);
        std::memcpy(link_overlap_avg + k * d.n_temps, &ck[(k * 2 + 1) * d.n_temps], sizeof(double) * d.n_temps);
    }
    return PA_OK;
}

uint64_t pa_sim_result_block_len(const pa_sim *sim) {
    return static_cast<uint64_t>(sim->d.n_temps) * (sim->d.n_pairs ? 11 : 5);
}

int pa_sim_get_result_blocks(pa_sim *sim, double *out) {
    const SimDims &d = sim->d;
    const uint64_t len = pa_sim_result_block_len(sim);
    for (uint32_t real = 0; real < d.n_real; ++real) {
        int rc = pa_sim_get_result_moments(sim, real, out + real * len);
        if (rc) return rc;
        if (d.n_pairs && (rc = pa_sim_get_result_overlap(sim, real, out + real * len + 5 * d.n_temps)))
            return rc;
    }
    return PA_OK;
}

} // extern "C"
