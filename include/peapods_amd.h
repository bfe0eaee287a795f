/*
 * peapods_amd.h -- C ABI of the MI355X-native spin-sim hot path.
 *
 * Drop-in boundary for ONE path of PeaBrane/peapods: the (disorder x temperature
 * x replica) ensemble Metropolis sweep + parallel-tempering energy/swap step +
 * replica-overlap reductions (+ Houdayer/Joerg overlap-cluster move).  The entry
 * points are what a Rust/C/Python FFI for that path binds; they take plain
 * pointers and sizes only.  Reference interfaces replaced (paths relative to the
 * reference root, ss/ = spin-sim/src/):
 *
 *   pa_sim_create / destroy / reset / sample / get_spins
 *        <- PyO3 class IsingSimulation {new, reset, sample, get_spins}, src/lib.rs:104-634
 *           (which wraps spin_sim::{Lattice, Realization, run_sweep_parallel}, src/lib.rs:9-10)
 *   pa_sim_metropolis_sweep      <- mcmc::sweep::metropolis_sweep,          ss/mcmc/sweep.rs:220-229
 *   pa_sim_gibbs_sweep           <- mcmc::sweep::gibbs_sweep,               ss/mcmc/sweep.rs:262-284
 *   pa_sim_compute_energies      <- spins::energy::compute_energies_and_magnetizations_into,
 *                                                                          ss/spins/energy.rs:59-65
 *   pa_sim_parallel_tempering    <- mcmc::tempering::parallel_tempering(_full_ladder) as driven by
 *                                   ss/simulation/mod.rs:748-796 (one event for every replica ladder)
 *   (sample with cluster_interval) <- clusters::fk_update, ss/clusters/fk.rs:27-200 (update action)
 *   pa_sim_overlap_dots          <- the integer part of OverlapAccum::collect, ss/statistics/overlap.rs:259-282
 *   pa_sim_get_result_*          <- SweepResult / OverlapStats fields, ss/statistics/results.rs:56-77,
 *                                   ss/statistics/overlap.rs:8-39 (per realisation, before aggregation)
 *   pa_aggregate_*               <- SweepResult::aggregate_impl, ss/statistics/results.rs:84-285 and
 *                                   OverlapStats::aggregate_impl, ss/statistics/overlap.rs:67-155
 *   pa_build_metropolis_table    <- UnitCouplingMetropolisLookup::new, ss/mcmc/sweep.rs:109-139
 *   pa_splitmix64 / pa_child_seed / pa_realization_seed
 *                                <- ss/simulation/realization.rs:9-19, src/lib.rs:22-32
 *
 * All state lives on the device between calls (continuation semantics of
 * repeated sample() calls, reference tests/test_sampling_interfaces.py:103-109).
 * Calls on one handle are not re-entrant; handles on different devices are
 * independent.  Every function returns PA_OK (0) or an error code; the text of
 * the last error is available from pa_sim_last_error / the create() buffer.
 * Options are validated before any state is touched (reference tests
 * :145-156,197-206 require spins unchanged after a rejected call).
 *
 * There is no CPU fallback: without a usable HIP device create() fails with
 * PA_ERR_DEVICE.
 */
#ifndef PEAPODS_AMD_H
#define PEAPODS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PA_OK = 0,
    PA_ERR_INVALID = 1,     /* bad argument / option (ValueError in the reference) */
    PA_ERR_UNSUPPORTED = 2, /* valid in the reference, not built on the device path yet */
    PA_ERR_DEVICE = 3,      /* HIP runtime error, or no device */
    PA_ERR_INTERRUPTED = 4
};

enum { PA_SWEEP_METROPOLIS = 0, PA_SWEEP_GIBBS = 1 };
enum { PA_PT_SINGLE_RANDOM_EDGE = 0, PA_PT_FULL_LADDER = 1 };
enum { PA_OC_HOUDAYER = 1, PA_OC_JORG = 2, PA_OC_CMR = 3 };
enum { PA_CLUSTER_WOLFF = 0, PA_CLUSTER_SW = 1 };
enum { PA_ACTION_UPDATE = 0, PA_ACTION_OBSERVE = 1 }; /* ClusterAction, ss/config.rs:41-59 */
enum { PA_AUTOCORR_RING = 0, PA_AUTOCORR_FFT = 1 };   /* AutocorrelationBackend, ss/config.rs */
/* rows of pa_sim_get_observation (per_disorder["cluster_observations"], src/lib.rs:416-445) */
enum { PA_OBS_FK = 0, PA_OBS_HOUDAYER = 1, PA_OBS_JORG = 2, PA_OBS_CMR_BLUE = 3 };

typedef struct pa_sim pa_sim;

/* Constructor arguments of IsingSimulation (src/lib.rs:106-115). */
typedef struct {
    uint32_t n_dims;
    const uint64_t *shape;           /* [n_dims] */
    uint32_t n_neighbors;            /* forward directions z'; 0 = hypercubic (n_dims) */
    const int64_t *neighbor_offsets; /* [n_neighbors][n_dims] or NULL for hypercubic */
    const float *couplings;          /* [n_realizations][n_spins][n_neighbors], forward bonds */
    uint64_t n_realizations;
    const float *temperatures; /* [n_temps] */
    uint64_t n_temps;
    uint64_t n_replicas;
    uint64_t seed;  /* dynamics root seed (src/lib.rs:155) */
    int32_t device; /* HIP device ordinal */
    /* Global index of this handle's realisation 0.  Realisation seeds depend only on the global
     * index (src/lib.rs:30-32), so a disorder ensemble sharded over several handles / GPUs gives
     * the same per-realisation results as one handle holding all of them. */
    uint64_t first_realization;
    /* Stream groups the realisations are split into (each group's launches go to a stream of its own, so that one
     * group's random stream overlaps another's sweep); 0 = the default (2).  1 = every launch covers the whole
     * ensemble on one stream: what pa_sim_profile_step needs.  The trajectory does not depend on it. */
    uint32_t stream_groups;
} pa_sim_desc;

/* Keyword arguments of IsingSimulation.sample that are on the path
 * (src/lib.rs:176-284 -> SimConfig, ss/config.rs:251-263). */
typedef struct {
    uint64_t n_sweeps;
    uint64_t warmup_sweeps;
    int32_t sweep_mode;                /* PA_SWEEP_* */
    uint64_t pt_interval;              /* 0 = no parallel tempering */
    int32_t pt_schedule;               /* PA_PT_* */
    uint64_t overlap_cluster_interval; /* 0 = none */
    int32_t overlap_cluster_modes[4];  /* PA_OC_*, round-robin per call */
    int32_t n_overlap_cluster_modes;
    int32_t overlap_cluster_mode; /* PA_CLUSTER_* */
    int32_t measure_overlap;      /* 1 = collect overlap statistics on recorded sweeps (reference
                                     behaviour whenever n_replicas >= 2); 0 only for timing runs */
    uint64_t cluster_interval;    /* FK cluster update of every system each k sweeps
                                     (cluster_update_interval, ss/clusters/fk.rs); 0 = none */
    int32_t cluster_mode;         /* PA_CLUSTER_*: Swendsen-Wang or Wolff */
    int32_t equilibration_diagnostic; /* 1 = running means of the replica-averaged energy and the
                                     pair-averaged link overlap at sweeps 128, 256, ..., n_sweeps
                                     (ss/statistics/equilibration.rs, driver ss/simulation/mod.rs:511-541) */
    uint64_t autocorrelation_max_lag; /* 0 = none; else the integrated autocorrelation times of m^2 and
                                     q^2 are collected (ring backend, ss/statistics/autocorrelation.rs,
                                     driver ss/simulation/mod.rs:342-371,580-595) */
    int32_t collect_cluster_stats;  /* 1 = cluster-size histograms of the FK / overlap clusters and the four
                                     largest overlap clusters on recorded sweeps (collect_cluster_stats,
                                     src/lib.rs:243,266; driver ss/simulation/mod.rs:442-483,604-735).  As in
                                     the reference, single-cluster moves then take their union-find form. */
    int32_t cluster_action;         /* PA_ACTION_*: FK update or observation (ss/config.rs:41-59) */
    int32_t overlap_cluster_action; /* PA_ACTION_* for the overlap move */
    int32_t autocorrelation_backend; /* PA_AUTOCORR_*: exact ring sums, or the lagged products by an FFT of the
                                     retained series (ss/statistics/autocorrelation.rs:126-163; equal up to
                                     rounding -- the transform is a radix-2 one, not rustfft's) */
    uint64_t snapshot_interval;     /* 0 = none; cluster ids, spins and systems of the first realisation's
                                     first pair at every temperature, every k sweeps (mod.rs:302-335,628-714) */
} pa_sample_config;

/* ---- lifecycle ----------------------------------------------------------- */
int pa_sim_create(const pa_sim_desc *desc, pa_sim **out, char *err, size_t err_len);
void pa_sim_destroy(pa_sim *sim);
int pa_sim_reset(pa_sim *sim, uint64_t seed);
const char *pa_sim_last_error(const pa_sim *sim);

/* Checks a sample configuration without touching any state (ss/config.rs:180-247).
 * Usable without a device; `err` receives the reference's message text. */
int pa_validate_sample_config(const pa_sample_config *cfg, uint64_t n_replicas, char *err,
                              size_t err_len);

/* ---- the driver loop (ss/simulation/mod.rs:405-797 for every realisation) -- */
int pa_sim_sample(pa_sim *sim, const pa_sample_config *cfg);
/* Asynchronous variant for timing: returns once the last sweeps are enqueued (the host runs at most two
 * batches of 256 sweeps ahead of the device); pa_sim_synchronize waits for them. */
int pa_sim_sample_async(pa_sim *sim, const pa_sample_config *cfg);
int pa_sim_synchronize(pa_sim *sim);
/* Cooperative cancel (the reference's AtomicBool polled once per sweep, src/lib.rs:304-308,
 * ss/simulation/mod.rs:406-408).  May be called from another thread while sample() runs: the run stops
 * within two batches of 256 queued sweeps, sample() returns PA_ERR_INTERRUPTED ("interrupted"), and the
 * ensemble is left in the state after the last sweep that ran -- statistics hold whole sweeps only and a
 * following sample() continues from there.  A request made while no sample() is running is dropped. */
void pa_sim_request_interrupt(pa_sim *sim);
/* Sweeps the last sample() ran: n_sweeps, or fewer when it was interrupted (always whole sweeps). */
uint64_t pa_sim_sweeps_completed(const pa_sim *sim);

/* Device time of EVERY launch of one step of `cfg`'s protocol, cluster moves included, by kernel (the name is the kernel
 * expression of the launch site, e.g. "k_rng_relay<P, R, G>").  sample()'s own launch sequence runs for `n_sweeps`
 * sweeps with an event in front of every launch, then again without events; each kernel's share of the traced run is
 * applied to the untraced step time *step_ms_out.  names_out: the kernel names, one per line; ms_per_step_out[k] and
 * launches_per_step_out[k] belong to line k.  The handle must have one stream group (pa_sim_desc.stream_groups = 1).
 * State advances by 2 n_sweeps sweeps; the results of the handle are those of the second run. */
int pa_sim_profile_step(pa_sim *sim, const pa_sample_config *cfg, uint64_t n_sweeps, char *names_out, uint64_t names_capacity,
                        double *ms_per_step_out, double *launches_per_step_out, uint64_t capacity, uint64_t *n_kernels_out,
                        double *step_ms_out);

/* ---- phase-level entry points (inner boundary; state stays on the device) -- */
int pa_sim_metropolis_sweep(pa_sim *sim, uint64_t n_sweeps);
int pa_sim_gibbs_sweep(pa_sim *sim, uint64_t n_sweeps);
int pa_sim_compute_energies(pa_sim *sim);
int pa_sim_parallel_tempering(pa_sim *sim, int32_t schedule);
/* dot_spin/dot_link: [n_realizations][n_pairs][n_temps] i64, pairs (2p, 2p+1) at slot t */
int pa_sim_overlap_dots(pa_sim *sim, int64_t *dot_spin, int64_t *dot_link);

/* ---- geometry / sizes ---------------------------------------------------- */
uint64_t pa_sim_n_spins(const pa_sim *sim);
uint64_t pa_sim_n_neighbors(const pa_sim *sim);
uint64_t pa_sim_n_systems(const pa_sim *sim);

/* ---- state getters (caller-allocated; reference layouts) ------------------ */
/* spins: i8 +-1, [n_systems][n_spins] system-major for one realisation (src/lib.rs:620-622) */
int pa_sim_get_spins(pa_sim *sim, uint64_t realization, int8_t *out);
/* system_ids: [n_systems] slot -> system (Realization::system_ids) */
int pa_sim_get_system_ids(pa_sim *sim, uint64_t realization, uint64_t *out);
/* energies: f32 [n_systems] indexed by system (Realization::energies), as of the last computation */
int pa_sim_get_energies(pa_sim *sim, uint64_t realization, float *out);
/* magnetisation sums: i64 [n_systems] indexed by system, as of the last computation */
int pa_sim_get_magnetization_sums(pa_sim *sim, uint64_t realization, int64_t *out);
/* xoshiro256** state of one system's generator (Realization::rngs[system]) */
int pa_sim_get_rng_state(pa_sim *sim, uint64_t realization, uint64_t system, uint64_t out[4]);
/* ---- state setters (caller-owned buffers in the reference's layouts; validated before anything changes) ----
 * The reference's kernels borrow the slices of a Realization (ss/mcmc/sweep.rs:220-229, ss/spins/energy.rs:43-48,
 * ss/mcmc/tempering.rs:20-27), so a caller may hand over any state; these are that hand-over for a handle.
 * spins: i8 +-1 [n_systems][n_spins] by system (the cached energies / magnetisation sums are recomputed);
 * system_ids: [n_systems] slot -> system, a permutation inside every replica's ladder (spins, generators and cached
 * sums stay with their systems, as in the reference, where only this array is permuted);
 * rng state: the four words of Realization::rngs[system]. */
int pa_sim_set_spins(pa_sim *sim, uint64_t realization, const int8_t *spins);
int pa_sim_set_system_ids(pa_sim *sim, uint64_t realization, const uint64_t *system_ids);
int pa_sim_set_rng_state(pa_sim *sim, uint64_t realization, uint64_t system, const uint64_t state[4]);
/* PtState counters: edge_* [n_temps-1], round_trips [n_replicas*n_temps] by system */
int pa_sim_get_pt_counters(pa_sim *sim, uint64_t realization, uint64_t *edge_attempts,
                           uint64_t *edge_acceptances, uint64_t *round_trips);

/* ---- per-realisation results of the last sample() ------------------------- */
/* moments: f64 [5][n_temps] = mags, mags2, mags4, energies, energies2 */
int pa_sim_get_result_moments(pa_sim *sim, uint64_t realization, double *out);
/* overlap: f64 [6][n_temps] = q, q2, q4, ql, ql2, ql4 (requires n_replicas >= 2) */
int pa_sim_get_result_overlap(pa_sim *sim, uint64_t realization, double *out);
/* histogram u64, ql_at_q_sum f64, ql2_at_q_sum f64: each [n_temps][n_spins + 1] */
int pa_sim_get_result_histograms(pa_sim *sim, uint64_t realization, uint64_t *histogram,
                                 double *ql_at_q_sum, double *ql2_at_q_sum);
/* All realisations at once, packed f64 block per realisation:
 * [5][n_temps] moments then (if pairs) [6][n_temps] overlap moments. Used for the
 * multi-GPU all-gather of observables. */
uint64_t pa_sim_result_block_len(const pa_sim *sim);
/* Equilibration checkpoints of the last sample() (EquilCheckpoint, ss/statistics/equilibration.rs:1-5):
 * sweeps [n], energy_avg [n][n_temps], link_overlap_avg [n][n_temps] of one realisation. */
/* Sokal-window integrated autocorrelation times per temperature of the last sample() with
 * autocorrelation_max_lag > 0 (sokal_tau over AutocorrAccum::finish, autocorrelation.rs:116-208):
 * mags2_tau [n_temps], overlap2_tau [n_temps] (untouched without replica pairs). */
int pa_sim_get_autocorrelation_tau(pa_sim *sim, uint64_t realization, double *mags2_tau, double *overlap2_tau);
/* Cluster statistics of the last sample() for one realisation (ClusterStats, ss/statistics/results.rs:12-20;
 * filled by ss/simulation/mod.rs:471-478,716-735,799-823).  fk_csd u64 [n_temps][n_spins + 1];
 * overlap_csd u64 [n_modes][n_temps][n_spins + 1]; top_sizes f64 [n_modes][n_temps][4] (mean relative size of
 * the four largest overlap clusters).  Any pointer may be NULL.  Returns bit 0 = FK histogram present,
 * bit 1 = overlap statistics present, or a negative PA_ERR_*. */
int pa_sim_get_cluster_stats(pa_sim *sim, uint64_t realization, uint64_t *fk_csd, uint64_t *overlap_csd,
                             double *top_sizes);
/* Graph observation summary `kind` (PA_OBS_*) of one realisation (GraphObservationSummary, results.rs:22-31;
 * GraphObservationAccum::finish, mod.rs:57-97): observation_count u64 [n_temps], cluster_size_counts u64
 * [n_temps][n_spins + 1], top_four_component_fractions f64 [n_temps][4], active_bond_density f64 [n_temps],
 * large_component_count f64 [n_temps], winding f64 [n_temps][4] = x, y, either, both.  Returns 0 = absent,
 * 1 = present, 3 = present with winding (2D hypercubic lattices), or a negative PA_ERR_*. */
int pa_sim_get_observation(pa_sim *sim, uint64_t realization, int32_t kind, uint64_t *count, uint64_t *sizes,
                           double *top4_frac, double *bond_density, double *large_mean, double *winding);
/* Cluster snapshots of the last sample() (ClusterSnapshot, results.rs:47-54; src/lib.rs:576-614):
 * cluster_ids / blue_ids u32 [n_temps][n_spins] (union-find roots in the reference's convention), spins i8
 * [n_temps][2][n_spins], system_ids u64 [n_temps][2].  Returns 1 when blue_ids are present (CMR). */
uint64_t pa_sim_n_snapshots(const pa_sim *sim);
int pa_sim_get_snapshot(pa_sim *sim, uint64_t index, uint64_t *sweep_id, int32_t *mode_idx, uint32_t *cluster_ids,
                        uint32_t *blue_ids, int8_t *spins, uint64_t *system_ids);
uint64_t pa_sim_n_equil_checkpoints(const pa_sim *sim);
int pa_sim_get_equilibration(pa_sim *sim, uint64_t realization, uint64_t *sweeps, double *energy_avg,
                             double *link_overlap_avg);
int pa_sim_get_result_blocks(pa_sim *sim, double *out /* [n_realizations][block_len] */);

/* ---- multi-GPU: disorder shards + ONE collective (RCCL over xGMI) ---------------------------------
 * The reference has a single process whose outer rayon loop runs the realisations
 * (ss/simulation/mod.rs:887-903) and aggregates them serially (ss/statistics/results.rs:165-282).  Here
 * rank g of G owns a contiguous block of realisations (pa_sim_desc.first_realization), runs with no
 * data-path collective, and the fixed-size per-realisation result blocks are all-gathered once at the
 * end; every rank then applies pa_aggregate_* in realisation order, which reproduces a single-GPU run
 * bit for bit.  One communicator rank per process / GPU.  The 128-byte id is created by one rank and
 * handed to the others by the caller over any channel.  RCCL is bound at run time: without librccl
 * these calls return PA_ERR_DEVICE and a message, the rest of the library is unaffected. */
#define PA_COMM_ID_BYTES 128
typedef struct pa_comm pa_comm;
int pa_comm_unique_id(uint8_t id_out[PA_COMM_ID_BYTES], char *err, size_t err_len); /* ncclGetUniqueId */
int pa_comm_init_rank(const uint8_t id[PA_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int32_t device,
                      pa_comm **out, char *err, size_t err_len);                   /* ncclCommInitRank */
void pa_comm_destroy(pa_comm *comm);
const char *pa_comm_last_error(const pa_comm *comm);
int32_t pa_comm_n_ranks(const pa_comm *comm);
int32_t pa_comm_rank(const pa_comm *comm);
/* Host buffers, staged through device memory: recv_host receives n_ranks * bytes_per_rank in rank order. */
int pa_comm_allgather(pa_comm *comm, const void *send_host, void *recv_host, uint64_t bytes_per_rank);
int pa_comm_allreduce_max_f64(pa_comm *comm, double *inout, uint64_t count); /* e.g. max-over-ranks timing */
int pa_comm_allreduce_sum_u64(pa_comm *comm, uint64_t *inout, uint64_t count); /* exact; e.g. summed P(q) counts */
int pa_comm_barrier(pa_comm *comm);
/* Variable-length gather in rank order: every rank contributes send_bytes host bytes (0 allowed); recv_host
 * receives the concatenation on `root`, or on every rank when root < 0; bytes_per_rank_out [n_ranks] is filled on
 * every rank.  The per-realisation result arrays of a sharded run travel through it (peapods_amd.distributed.sample:
 * statistics/results.rs:165-282 and overlap.rs:106-152 then run on the concatenation, as for one process). */
int pa_comm_gatherv(pa_comm *comm, const void *send_host, uint64_t send_bytes, void *recv_host, uint64_t recv_capacity,
                    int32_t root, uint64_t *bytes_per_rank_out);
/* All ranks' pa_sim_get_result_blocks in rank (= global realisation) order; ranks may own different numbers
 * of realisations.  out: [out_capacity][pa_sim_result_block_len] f64; *n_total_out = blocks gathered. */
int pa_comm_gather_result_blocks(pa_comm *comm, pa_sim *sim, double *out, uint64_t out_capacity,
                                 uint64_t *n_total_out);
uint64_t pa_sim_n_realizations(const pa_sim *sim);

/* ---- measurement helpers -------------------------------------------------------------------------
 * Attainable HBM bandwidth of `device` beside the 8 TB/s specification: a float4 copy of
 * `bytes` (read + write both counted), best of `iterations` launches, HIP-event timed. */
int pa_device_copy_bandwidth(int32_t device, uint64_t bytes, int32_t iterations, double *gbytes_per_s_out,
                             char *err, size_t err_len);
/* sizeof and field offsets of the two argument structs as this library was compiled:
 * out = { sizeof(pa_sim_desc), n_fields, offsets..., sizeof(pa_sample_config), n_fields, offsets... } in
 * declaration order; returns the number of entries (written up to `capacity`).  Bindings assert their
 * own layout against it (tests/test_host_logic.py does so for the ctypes binding, INTEGRATION.md's Rust
 * structs are generated from it). */
uint64_t pa_abi_layout(uint64_t *out, uint64_t capacity);

/* ---- host-only helpers (no device needed) --------------------------------- */
uint64_t pa_splitmix64(uint64_t value);
uint64_t pa_child_seed(uint64_t root, uint64_t domain, uint64_t index);
uint64_t pa_realization_seed(uint64_t root, uint64_t realization);
/* xoshiro256** as seeded by the reference: state from seed, then `n_skip` steps. */
void pa_xoshiro_state(uint64_t seed, uint64_t n_skip, uint64_t out[4]);
/* Jump-ahead used by the device generator: state advanced by `n_steps` through the
 * GF(2) matrix machinery (must equal n_steps sequential steps). */
void pa_xoshiro_jump(const uint64_t in[4], uint64_t n_steps, uint64_t out[4]);
/* Acceptance table, [n_temperatures][4*n_neighbors + 1] u32; returns PA_ERR_UNSUPPORTED when the
 * reference would decline the table (non-unit couplings / bad temperature). */
int pa_build_metropolis_table(const float *couplings, uint64_t n_couplings,
                              const float *temperatures, uint64_t n_temperatures,
                              uint64_t n_neighbors, uint32_t *out);
/* Gibbs (heat-bath) counterpart for unit couplings: out[t * (4z'+1) + (e + 2z')] = number of 24-bit
 * draws k with e >= (T/2) ln(u / (1 - u)), u = k * 2^-24 (ss/mcmc/sweep.rs:262-284 on the grid of
 * gen::<f32>()).  PA_ERR_UNSUPPORTED outside the table's domain. */
int pa_build_gibbs_table(const float *couplings, uint64_t n_couplings, const float *temperatures,
                         uint64_t n_temperatures, uint64_t n_neighbors, uint32_t *out);
/* Dependency levels of the typewriter sweep: level[i] for every site; returns the level count.
 * Sites of one level are mutually independent; executing levels in order reproduces the
 * reference's sequential site order exactly. */
uint64_t pa_build_sweep_levels(uint32_t n_dims, const uint64_t *shape, uint32_t n_neighbors,
                               const int64_t *neighbor_offsets, uint32_t *level_out);
/* Mean over realisations of per-realisation f64 vectors, summed sequentially in realisation
 * order then divided once (results.rs:165-196,262-282). */
void pa_aggregate_mean(const double *per_realization, uint64_t n_realizations, uint64_t len,
                       double *out);
void pa_aggregate_sum_u64(const uint64_t *per_realization, uint64_t n_realizations, uint64_t len,
                          uint64_t *out);
void pa_aggregate_sum_f64(const double *per_realization, uint64_t n_realizations, uint64_t len,
                          double *out);
/* Integrated autocorrelation time per column of a per-sweep series [n_samples][n_cols] (f32, as pushed):
 * AutocorrAccum::push/finish + sokal_tau (ss/statistics/autocorrelation.rs:70-208), in f64 on the host.
 * backend = PA_AUTOCORR_RING (the exact lagged sums) or PA_AUTOCORR_FFT (:126-163).  This is what
 * pa_sim_get_autocorrelation_tau applies to the series the device wrote; exported for tests. */
void pa_autocorrelation_tau(const float *series, uint64_t n_samples, uint64_t n_cols, uint64_t max_lag,
                            double *tau_out, int32_t backend);
/* Number of visible HIP devices (0 when there is none / the runtime is unusable). */
int pa_device_count(void);
const char *pa_version(void);
/* Developer switches honoured so far by this process, space separated ("NAME=value ..."), or "".  Kernel-selection
 * and tuning variables (PEAPODS_NO_RELAY, PEAPODS_STREAMS, PEAPODS_RNG_BLOCK, ...) are ignored unless
 * PEAPODS_DEVELOPER=1 is set with them: a stray variable cannot change which kernels a user runs.  (The reference has
 * the same kind of switches in its own benchmark only: spin-sim/examples/bench.rs:19-21.)  The pointer stays valid
 * until the next call. */
const char *pa_active_switches(void);

#ifdef __cplusplus
}
#endif
#endif
