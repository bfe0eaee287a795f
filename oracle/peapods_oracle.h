/*
 * peapods_oracle.h -- CPU restatement of the PeaBrane/peapods spin-sim hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C restatement
 * of the reference's algorithm (typewriter site order, one Xoshiro256** stream
 * per system, SplitMix64 seed tree, 24-bit integer Metropolis acceptance table,
 * system_ids-permutation parallel tempering, i64 overlap dot products).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped product (peapods_amd/, libpeapods_amd.so) never links or calls it.
 *
 * PARITY UNPINNED at the bit level against upstream binaries: the reference is
 * a Rust crate (no cargo/rustc in this image) and its tests hold no absolute
 * trajectory or RNG vector; the oracle is pinned by the reference's own
 * known-answer facts (neighbour tables, energy KAT, PT edge order, acceptance
 * table boundary property, table==log equivalence, union-find clusters) and by
 * the published xoshiro256** / SplitMix64 vectors.  See DESIGN.md section (c).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root; spin-sim/src/ abbreviated as ss/).
 */
#ifndef PEAPODS_ORACLE_H
#define PEAPODS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- seeds and RNG (ss/simulation/realization.rs:6-19, src/lib.rs:22-32,
 *      rand 0.8.5 / rand_core 0.6.4 / rand_xoshiro 0.6.0 semantics) ---------- */
#define ORC_SYSTEM_SEED_DOMAIN 0x53A917E14C2D8B6FULL
#define ORC_PAIR_SEED_DOMAIN 0xA1475EED91C302D7ULL

typedef struct {
    uint64_t s[4];
} orc_rng;

uint64_t orc_splitmix64(uint64_t value);
uint64_t orc_child_seed(uint64_t root, uint64_t domain, uint64_t index);
uint64_t orc_realization_seed(uint64_t root, uint64_t realization);
void orc_rng_seed_from_u64(orc_rng *rng, uint64_t seed);
uint64_t orc_rng_next_u64(orc_rng *rng);
uint32_t orc_rng_next_u32(orc_rng *rng);
float orc_rng_gen_f32(orc_rng *rng);
uint64_t orc_rng_gen_range_usize(orc_rng *rng, uint64_t low, uint64_t high);
void orc_rng_shuffle_usize(orc_rng *rng, uint64_t *values, size_t len);
/* pinning only: the oracle's u32 range sampling / f32 conversion on a supplied word stream (rand 0.8.5 value-stability vectors) */
size_t orc_pin_sample_single_u32(uint32_t low, uint32_t high, const uint32_t *words, size_t n_words, uint32_t *out);
float orc_pin_f32_of_word(uint32_t word);
size_t orc_pin_sample_single_inclusive_u8(uint8_t low, uint8_t high, const uint32_t *words, size_t n_words, uint8_t *out);

/* ---- lattice (ss/geometry/lattice.rs:9-109, offsets.rs) ------------------- */
#define ORC_MAX_DIMS 8
typedef struct {
    int n_dims;
    size_t shape[ORC_MAX_DIMS];
    size_t strides[ORC_MAX_DIMS];
    size_t n_spins;
    size_t n_neighbors;
    uint32_t *fwd; /* [n_spins * n_neighbors] */
    uint32_t *bwd;
    int has_square_shape; /* canonical 2D flag, lattice.rs:31-38 */
} orc_lattice;

orc_lattice *orc_lattice_new(const uint64_t *shape, int n_dims);
orc_lattice *orc_lattice_with_offsets(const uint64_t *shape, int n_dims, const int64_t *offsets,
                                      size_t n_offsets);
void orc_lattice_free(orc_lattice *lat);
uint64_t orc_lattice_n_spins(const orc_lattice *lat);
uint64_t orc_lattice_n_neighbors(const orc_lattice *lat);
uint64_t orc_lattice_stride(const orc_lattice *lat, int d);
uint64_t orc_lattice_neighbor_fwd(const orc_lattice *lat, uint64_t i, uint64_t d);
uint64_t orc_lattice_neighbor_bwd(const orc_lattice *lat, uint64_t i, uint64_t d);

/* ---- Metropolis acceptance table (ss/mcmc/sweep.rs:99-167) ---------------- */
typedef struct {
    uint32_t *accepted_counts; /* [n_rows * table_width] */
    int32_t offset;
    size_t table_width;
    size_t n_rows;
} orc_lookup;

orc_lookup *orc_lookup_new(const float *couplings, size_t n_couplings, const float *temperatures,
                           size_t n_temperatures, size_t n_neighbors);
void orc_lookup_free(orc_lookup *lookup);
uint32_t orc_lookup_accepted_count(float temperature, int32_t energy_change);
int orc_lookup_legacy_accepts(float temperature, int32_t energy_change, uint32_t draw);
uint32_t orc_lookup_get(const orc_lookup *lookup, size_t temperature_id, int32_t energy_change);

/* ---- single-spin sweeps (ss/mcmc/sweep.rs:8-97,170-284, ss/parallel.rs) ---- */
void orc_metropolis_sweep(const orc_lattice *lat, int8_t *spins, const float *couplings,
                          const float *temperatures, const uint64_t *system_ids, size_t n_slots,
                          orc_rng *rngs, const orc_lookup *lookup /* may be NULL */);
void orc_gibbs_sweep(const orc_lattice *lat, int8_t *spins, const float *couplings,
                     const float *temperatures, const uint64_t *system_ids, size_t n_slots,
                     orc_rng *rngs);

/* ---- energies (ss/spins/energy.rs:78-110) --------------------------------- */
void orc_compute_energies(const orc_lattice *lat, const int8_t *spins, const float *couplings,
                          size_t n_systems, float *energies, int64_t *magnetization_sums,
                          float *interactions);

/* ---- parallel tempering (ss/mcmc/tempering.rs) ---------------------------- */
typedef struct {
    uint64_t edge;
    int accepted;
    uint64_t left_system;
    uint64_t right_system;
} orc_tempering_attempt;

/* Both return the number of attempts written to `attempts` (capacity >= n_temps). */
size_t orc_parallel_tempering(const float *energies, const float *temperatures,
                              uint64_t *system_ids, size_t n_temps, size_t n_spins, orc_rng *rng,
                              orc_tempering_attempt *attempts);
size_t orc_parallel_tempering_full_ladder(const float *energies, const float *temperatures,
                                          uint64_t *system_ids, size_t n_temps, size_t n_spins,
                                          orc_rng *rng, size_t first_parity,
                                          orc_tempering_attempt *attempts);

void orc_pt_replay(size_t n_replicas, size_t n_temps, const float *temps,
                   const orc_tempering_attempt *attempts, size_t n_attempts,
                   uint64_t *edge_attempts, uint64_t *edge_acceptances, uint64_t *round_trips);

/* ---- union-find / DFS helpers (ss/clusters/utils.rs:68-149,155-209,260-316) */
uint32_t orc_uf_find(uint32_t *parent, uint32_t x);
void orc_uf_union(uint32_t *parent, uint8_t *rank, uint32_t a, uint32_t b);
/* bonds given as (site, dim) pairs; fills parent[n], counts[n] (flattened). */
void orc_uf_from_bonds(const orc_lattice *lat, const uint64_t *bond_sites, const uint64_t *bond_dims,
                       size_t n_bonds, uint32_t *parent, uint32_t *counts);
void orc_uf_histogram(const uint32_t *counts, size_t n, uint64_t *hist /* [n+1] */);
void orc_top4_sizes(const uint32_t *counts, size_t n, uint32_t out[4]);
/* graph observation of a bond list applied in order (BondMetrics + WindingUf, utils.rs:317-455):
 * out = {active_bonds, top4[4], large_components, winding_x, winding_y, observed} */
void orc_graph_observation(const orc_lattice *lat, const uint64_t *bond_sites, const uint64_t *bond_dims,
                           size_t n_bonds, uint32_t out[9]);

/* ---- full simulation (ss/simulation/{mod.rs,realization.rs}, src/lib.rs) -- */
enum { ORC_SWEEP_METROPOLIS = 0, ORC_SWEEP_GIBBS = 1 };
enum { ORC_PT_SINGLE_RANDOM_EDGE = 0, ORC_PT_FULL_LADDER = 1 };
enum { ORC_OC_NONE = 0, ORC_OC_HOUDAYER = 1, ORC_OC_JORG = 2, ORC_OC_CMR = 3 };
enum { ORC_CLUSTER_WOLFF = 0, ORC_CLUSTER_SW = 1 };
enum { ORC_ACTION_UPDATE = 0, ORC_ACTION_OBSERVE = 1 }; /* ss/config.rs:41-59 */

typedef struct {
    uint64_t n_sweeps;
    uint64_t warmup_sweeps;
    int32_t sweep_mode;
    uint64_t pt_interval; /* 0 = no parallel tempering */
    int32_t pt_schedule;
    uint64_t overlap_cluster_interval; /* 0 = none */
    int32_t overlap_cluster_modes[4];  /* round-robin list of ORC_OC_* */
    int32_t n_overlap_cluster_modes;
    int32_t overlap_cluster_mode; /* ORC_CLUSTER_* */
    int32_t n_threads;            /* realisations are distributed over threads; <=1 = serial */
    uint64_t cluster_interval;    /* FK cluster update every k sweeps (ss/clusters/fk.rs); 0 = none */
    int32_t cluster_mode;         /* ORC_CLUSTER_* */
    uint64_t autocorrelation_max_lag; /* 0 = none; ring backend of ss/statistics/autocorrelation.rs */
    int32_t equilibration_diagnostic; /* running means of E and q_l at log-spaced checkpoints
                                         (ss/statistics/equilibration.rs) */
    int32_t collect_cluster_stats;    /* cluster size histograms + top-4 sizes (src/lib.rs:243,266) */
    int32_t cluster_action;           /* ORC_ACTION_*: FK update */
    int32_t overlap_cluster_action;   /* ORC_ACTION_*: overlap move */
    uint64_t snapshot_interval;       /* 0 = none; cluster snapshots of realisation 0 (mod.rs:302-335) */
} orc_config;

typedef struct orc_sim orc_sim;

orc_sim *orc_sim_create(const uint64_t *shape, int n_dims, const int64_t *offsets, size_t n_offsets,
                        const float *couplings, size_t n_realizations, const float *temperatures,
                        size_t n_temps, size_t n_replicas, uint64_t seed);
orc_sim *orc_sim_create_shard(const uint64_t *shape, int n_dims, const int64_t *offsets,
                              size_t n_offsets, const float *couplings, size_t n_realizations,
                              const float *temperatures, size_t n_temps, size_t n_replicas,
                              uint64_t seed, uint64_t first_realization);
void orc_sim_destroy(orc_sim *sim);
void orc_sim_reset(orc_sim *sim, uint64_t seed);
/* 0 on success; error text via orc_sim_last_error. */
int orc_sim_sample(orc_sim *sim, const orc_config *cfg);
const char *orc_sim_last_error(const orc_sim *sim);

uint64_t orc_sim_n_spins(const orc_sim *sim);
uint64_t orc_sim_n_neighbors(const orc_sim *sim);
/* state getters: realisation-major */
void orc_sim_get_spins(const orc_sim *sim, size_t realization, int8_t *out);        /* [n_sys*N] */
void orc_sim_get_system_ids(const orc_sim *sim, size_t realization, uint64_t *out); /* [n_sys] */
void orc_sim_get_energies(const orc_sim *sim, size_t realization, float *out);      /* [n_sys] */
void orc_sim_get_rng_state(const orc_sim *sim, size_t realization, size_t system, uint64_t out[4]);
void orc_sim_get_pt_counters(const orc_sim *sim, size_t realization, uint64_t *edge_attempts,
                             uint64_t *edge_acceptances, uint64_t *round_trips);
/* per-realisation results of the LAST sample() call (SweepResult, statistics/results.rs:56) */
void orc_sim_get_result_moments(const orc_sim *sim, size_t realization,
                                double *out /* [5][n_temps]: mags,mags2,mags4,energies,energies2 */);
int orc_sim_has_overlap(const orc_sim *sim);
/* equilibration checkpoints of the last sample(): count, then sweeps[n], energy_avg[n][n_temps],
 * link_overlap_avg[n][n_temps] of one realisation */
/* integrated autocorrelation times of m^2 and q^2 per temperature (Sokal window) of one realisation;
 * returns 0 when the last sample() did not collect them (overlap2_tau only with >= 2 replicas) */
int orc_sim_get_tau(const orc_sim *sim, size_t realization, double *mags2_tau, double *overlap2_tau);
size_t orc_sim_n_equil_checkpoints(const orc_sim *sim);
void orc_sim_get_equil(const orc_sim *sim, size_t realization, uint64_t *sweeps, double *energy_avg,
                       double *link_overlap_avg);
void orc_sim_get_result_overlap(const orc_sim *sim, size_t realization,
                                double *out /* [6][n_temps]: q,q2,q4,ql,ql2,ql4 */);
/* cluster statistics / graph observations / snapshots of the last sample() (results.rs:12-54) */
int orc_sim_get_cluster_stats(const orc_sim *sim, size_t realization, uint64_t *fk_csd /* [n_temps][N+1] */,
                              uint64_t *overlap_csd /* [n_modes][n_temps][N+1] */,
                              double *top_sizes /* [n_modes][n_temps][4] */);
int orc_sim_get_observation(const orc_sim *sim, size_t realization, int kind, uint64_t *count, uint64_t *sizes,
                            double *top4_frac, double *bond_density, double *large_mean, double *winding);
size_t orc_sim_n_snapshots(const orc_sim *sim);
int orc_sim_get_snapshot(const orc_sim *sim, size_t k, uint64_t *sweep_id, int32_t *mode_idx,
                         uint32_t *cluster_ids, uint32_t *blue_ids, int8_t *spins, uint64_t *system_ids);
void orc_sim_get_result_histograms(const orc_sim *sim, size_t realization,
                                   uint64_t *histogram /* [n_temps][N+1] */,
                                   double *ql_at_q_sum, double *ql2_at_q_sum);

#ifdef __cplusplus
}
#endif
#endif
