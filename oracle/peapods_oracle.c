/*
 * peapods_oracle.c -- CPU restatement of the PeaBrane/peapods spin-sim hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see peapods_oracle.h).  PARITY UNPINNED against
 * upstream binaries; pinned by the reference's own known-answer tests and the
 * published xoshiro256** / SplitMix64 vectors (tests/test_oracle_*.py); the u32 range
 * sampling and the f32 conversion also by rand 0.8.5's own value-stability vectors.
 *
 * Reference paths are relative to the reference root; "ss/" = spin-sim/src/.
 * Floating-point note: the reference is Rust; f32::ln / f32::exp lower to the
 * platform libm (logf/expf of glibc on Linux), which is what this file calls.
 * Build with -ffp-contract=off so no multiply-add is fused (rustc never fuses).
 */
#include "peapods_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ========================================================================== */
/* Seeds and RNG                                                              */
/* ========================================================================== */

/* ss/simulation/realization.rs:9-15 and src/lib.rs:22-28 (identical bodies). */
uint64_t orc_splitmix64(uint64_t value) {
    value += 0x9E3779B97F4A7C15ULL;
    uint64_t mixed = value;
    mixed = (mixed ^ (mixed >> 30)) * 0xBF58476D1CE4E5B9ULL;
    mixed = (mixed ^ (mixed >> 27)) * 0x94D049BB133111EBULL;
    return mixed ^ (mixed >> 31);
}

/* ss/simulation/realization.rs:17-19 */
uint64_t orc_child_seed(uint64_t root, uint64_t domain, uint64_t index) {
    return orc_splitmix64(root ^ domain ^ orc_splitmix64(index));
}

/* src/lib.rs:30-32 */
uint64_t orc_realization_seed(uint64_t root, uint64_t realization) {
    return orc_splitmix64(root ^ orc_splitmix64(realization));
}

/* rand_xoshiro 0.6.0 Xoshiro256StarStar::seed_from_u64: the 32 seed bytes are
 * four successive SplitMix64 outputs (little endian) starting from state
 * `seed`; an all-zero seed would be replaced by seed_from_u64(0), which cannot
 * happen for SplitMix64 output.  Call sites: realization.rs:170,189,219,234. */
void orc_rng_seed_from_u64(orc_rng *rng, uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) {
        x += 0x9E3779B97F4A7C15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        rng->s[i] = z ^ (z >> 31);
    }
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* xoshiro256** (Blackman & Vigna), as rand_xoshiro 0.6.0 next_u64. */
uint64_t orc_rng_next_u64(orc_rng *rng) {
    uint64_t *s = rng->s;
    uint64_t result = rotl64(s[1] * 5, 7) * 9;
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

/* rand_xoshiro 0.6.0: next_u32 = upper half of next_u64. */
uint32_t orc_rng_next_u32(orc_rng *rng) { return (uint32_t)(orc_rng_next_u64(rng) >> 32); }

/* rand 0.8.5 Standard f32: 24 high bits of one u32, scaled by 2^-24
 * (confirmed by the in-tree comment ss/mcmc/sweep.rs:179-181). */
static inline float f32_of_word(uint32_t value) { return (float)(value >> 8) * (1.0f / 16777216.0f); }
float orc_rng_gen_f32(orc_rng *rng) { return f32_of_word(orc_rng_next_u32(rng)); }

/* rand 0.8.5 UniformInt<usize>::sample_single (64-bit): widening multiply with
 * the "conservative zone" rejection.  Call sites: tempering.rs:33, utils.rs:113. */
uint64_t orc_rng_gen_range_usize(orc_rng *rng, uint64_t low, uint64_t high) {
    uint64_t range = high - low;
    int lz = __builtin_clzll(range);
    uint64_t zone = (range << lz) - 1;
    for (;;) {
        uint64_t v = orc_rng_next_u64(rng);
        __uint128_t m = (__uint128_t)v * (__uint128_t)range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return low + hi;
    }
}

/* rand 0.8.5 UniformInt<u32>::sample_single, used by gen_index for bounds that
 * fit a u32 (seq/mod.rs gen_index). */
static inline uint32_t u32_zone(uint32_t range) { return (range << __builtin_clz(range)) - 1; }
/* one attempt on the word v: 1 and *out = the sample if v is accepted */
static inline int u32_attempt(uint32_t v, uint32_t low, uint32_t range, uint32_t zone, uint32_t *out) {
    uint64_t m = (uint64_t)v * (uint64_t)range;
    uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
    if (lo > zone) return 0;
    *out = low + hi;
    return 1;
}
static uint32_t gen_range_u32(orc_rng *rng, uint32_t low, uint32_t high) {
    uint32_t range = high - low, zone = u32_zone(range), out;
    for (;;)
        if (u32_attempt(orc_rng_next_u32(rng), low, range, zone, &out)) return out;
}

/* Pinning only (tests/test_oracle_golden.py): the arithmetic of gen_range_u32 and orc_rng_gen_f32 on a word stream the
 * test supplies, so that rand 0.8.5's own value-stability vectors -- which that crate draws from a Pcg32, not from
 * xoshiro -- apply to the very code the oracle runs.  Returns the words consumed, 0 if the stream ran out. */
size_t orc_pin_sample_single_u32(uint32_t low, uint32_t high, const uint32_t *words, size_t n_words, uint32_t *out) {
    uint32_t range = high - low, zone = u32_zone(range);
    for (size_t k = 0; k < n_words; ++k)
        if (u32_attempt(words[k], low, range, zone, out)) return k + 1;
    return 0;
}
float orc_pin_f32_of_word(uint32_t word) { return f32_of_word(word); }

/* rand 0.8.5 UniformInt<u8>::sample_single_inclusive (uniform.rs, uniform_int_impl! with
 * $u_large = u32): zone = u32::MAX - (u32::MAX - range + 1) % range, one u32 per attempt.
 * Call sites: ss/clusters/overlap.rs:808 (1..=3), :827 (0..=3), :950 (1..=3). */
static inline uint32_t u8_range_inclusive(uint8_t low, uint8_t high) { return (uint32_t)(uint8_t)(high - low + 1); }
static inline uint32_t u8_zone(uint32_t range) { return UINT32_MAX - (UINT32_MAX - range + 1) % range; }
static uint8_t gen_range_u8_inclusive(orc_rng *rng, uint8_t low, uint8_t high) {
    uint32_t range = u8_range_inclusive(low, high), zone = u8_zone(range), out;
    for (;;)
        if (u32_attempt(orc_rng_next_u32(rng), low, range, zone, &out)) return (uint8_t)out;
}
/* pinning only, as orc_pin_sample_single_u32: rand's sample_single(low, high) on u8 = the inclusive form with high - 1 */
size_t orc_pin_sample_single_inclusive_u8(uint8_t low, uint8_t high, const uint32_t *words, size_t n_words, uint8_t *out) {
    uint32_t range = u8_range_inclusive(low, high), zone = u8_zone(range), v;
    for (size_t k = 0; k < n_words; ++k)
        if (u32_attempt(words[k], low, range, zone, &v)) {
            *out = (uint8_t)v;
            return k + 1;
        }
    return 0;
}

/* rand 0.8.5 SliceRandom::shuffle: for i in (1..len).rev() swap(i, gen_index(i+1)).
 * Call site: ss/clusters/overlap.rs:48. */
void orc_rng_shuffle_usize(orc_rng *rng, uint64_t *values, size_t len) {
    for (size_t i = len; i-- > 1;) {
        size_t j = (size_t)gen_range_u32(rng, 0, (uint32_t)(i + 1));
        uint64_t tmp = values[i];
        values[i] = values[j];
        values[j] = tmp;
    }
}

/* ========================================================================== */
/* Lattice                                                                    */
/* ========================================================================== */

/* ss/geometry/lattice.rs:44-99 */
orc_lattice *orc_lattice_with_offsets(const uint64_t *shape, int n_dims, const int64_t *offsets,
                                      size_t n_offsets) {
    if (n_dims < 1 || n_dims > ORC_MAX_DIMS) return NULL;
    orc_lattice *lat = (orc_lattice *)calloc(1, sizeof(*lat));
    lat->n_dims = n_dims;
    lat->n_neighbors = n_offsets;
    lat->n_spins = 1;
    for (int d = 0; d < n_dims; ++d) {
        lat->shape[d] = (size_t)shape[d];
        lat->n_spins *= (size_t)shape[d];
    }
    for (int d = 0; d < n_dims; ++d) lat->strides[d] = 1;
    for (int d = n_dims - 2; d >= 0; --d) lat->strides[d] = lat->strides[d + 1] * lat->shape[d + 1];

    lat->fwd = (uint32_t *)calloc(lat->n_spins * n_offsets + 1, sizeof(uint32_t));
    lat->bwd = (uint32_t *)calloc(lat->n_spins * n_offsets + 1, sizeof(uint32_t));
    for (size_t i = 0; i < lat->n_spins; ++i) {
        int64_t coords[ORC_MAX_DIMS];
        for (int d = 0; d < n_dims; ++d) coords[d] = (int64_t)((i / lat->strides[d]) % lat->shape[d]);
        for (size_t k = 0; k < n_offsets; ++k) {
            for (int pass = 0; pass < 2; ++pass) {
                int64_t sign = pass == 0 ? 1 : -1;
                size_t flat = 0;
                for (int dim = 0; dim < n_dims; ++dim) {
                    int64_t extent = (int64_t)lat->shape[dim];
                    int64_t c = (coords[dim] + sign * offsets[k * n_dims + dim]) % extent;
                    if (c < 0) c += extent; /* rem_euclid */
                    flat += (size_t)c * lat->strides[dim];
                }
                (pass == 0 ? lat->fwd : lat->bwd)[i * n_offsets + k] = (uint32_t)flat;
            }
        }
    }
    lat->has_square_shape = 0;
    return lat;
}

/* ss/geometry/lattice.rs:31-38 with offsets.rs:4-13 (hypercubic unit vectors). */
orc_lattice *orc_lattice_new(const uint64_t *shape, int n_dims) {
    int64_t offsets[ORC_MAX_DIMS * ORC_MAX_DIMS] = {0};
    for (int d = 0; d < n_dims; ++d) offsets[d * n_dims + d] = 1;
    orc_lattice *lat = orc_lattice_with_offsets(shape, n_dims, offsets, (size_t)n_dims);
    if (lat) lat->has_square_shape = (n_dims == 2);
    return lat;
}

void orc_lattice_free(orc_lattice *lat) {
    if (!lat) return;
    free(lat->fwd);
    free(lat->bwd);
    free(lat);
}

uint64_t orc_lattice_n_spins(const orc_lattice *lat) { return lat->n_spins; }
uint64_t orc_lattice_n_neighbors(const orc_lattice *lat) { return lat->n_neighbors; }
uint64_t orc_lattice_stride(const orc_lattice *lat, int d) { return lat->strides[d]; }
uint64_t orc_lattice_neighbor_fwd(const orc_lattice *lat, uint64_t i, uint64_t d) {
    return lat->fwd[i * lat->n_neighbors + d];
}
uint64_t orc_lattice_neighbor_bwd(const orc_lattice *lat, uint64_t i, uint64_t d) {
    return lat->bwd[i * lat->n_neighbors + d];
}

/* ========================================================================== */
/* Metropolis acceptance table                                                */
/* ========================================================================== */

#define F32_UNIFORM_BITS 24
#define F32_UNIFORM_VALUES (1u << F32_UNIFORM_BITS)

/* ss/mcmc/sweep.rs:155-159 */
int orc_lookup_legacy_accepts(float temperature, int32_t energy_change, uint32_t draw) {
    float uniform = (float)draw / (float)F32_UNIFORM_VALUES;
    return (float)energy_change >= (temperature / 2.0f) * logf(uniform);
}

/* ss/mcmc/sweep.rs:141-153 */
uint32_t orc_lookup_accepted_count(float temperature, int32_t energy_change) {
    uint32_t low = 0, high = F32_UNIFORM_VALUES;
    while (low < high) {
        uint32_t midpoint = low + (high - low) / 2;
        if (orc_lookup_legacy_accepts(temperature, energy_change, midpoint))
            low = midpoint + 1;
        else
            high = midpoint;
    }
    return low;
}

/* ss/mcmc/sweep.rs:109-139: fail closed unless every coupling is -1/0/+1 and
 * every temperature is finite with T/2 > 0. */
orc_lookup *orc_lookup_new(const float *couplings, size_t n_couplings, const float *temperatures,
                           size_t n_temperatures, size_t n_neighbors) {
    for (size_t i = 0; i < n_couplings; ++i) {
        float c = couplings[i];
        if (!(c == -1.0f || c == 0.0f || c == 1.0f)) return NULL;
    }
    for (size_t i = 0; i < n_temperatures; ++i) {
        float t = temperatures[i];
        if (!(isfinite(t) && t / 2.0f > 0.0f)) return NULL;
    }
    size_t offset = n_neighbors * 2;
    if (offset > F32_UNIFORM_VALUES) return NULL;
    orc_lookup *lookup = (orc_lookup *)calloc(1, sizeof(*lookup));
    lookup->offset = (int32_t)offset;
    lookup->table_width = offset * 2 + 1;
    lookup->n_rows = n_temperatures;
    lookup->accepted_counts =
        (uint32_t *)malloc(sizeof(uint32_t) * (n_temperatures * lookup->table_width + 1));
    size_t k = 0;
    for (size_t t = 0; t < n_temperatures; ++t)
        for (int32_t e = -lookup->offset; e <= lookup->offset; ++e)
            lookup->accepted_counts[k++] = orc_lookup_accepted_count(temperatures[t], e);
    return lookup;
}

void orc_lookup_free(orc_lookup *lookup) {
    if (!lookup) return;
    free(lookup->accepted_counts);
    free(lookup);
}

/* ss/mcmc/sweep.rs:161-166 */
uint32_t orc_lookup_get(const orc_lookup *lookup, size_t temperature_id, int32_t energy_change) {
    int32_t idx = energy_change + lookup->offset;
    if (idx < 0 || (size_t)idx >= lookup->table_width) {
        fprintf(stderr, "oracle: unit coupling energy change is out of range\n");
        abort();
    }
    return lookup->accepted_counts[temperature_id * lookup->table_width + (size_t)idx];
}

/* ========================================================================== */
/* Single-spin sweeps                                                         */
/* ========================================================================== */

/* ss/mcmc/sweep.rs:8-19.  The canonical-2D interior specialisation (:22-32)
 * adds the same four terms in the same order, so it is arithmetic-identical
 * (the reference's own test :293-343 asserts that); one path restates both. */
static inline float local_field(const orc_lattice *lat, const int8_t *spin_slice,
                                const float *couplings, size_t i) {
    size_t nn = lat->n_neighbors;
    float h = 0.0f;
    for (size_t d = 0; d < nn; ++d) {
        size_t j_fwd = lat->fwd[i * nn + d];
        h += (float)spin_slice[j_fwd] * couplings[i * nn + d];
        size_t j_bwd = lat->bwd[i * nn + d];
        h += (float)spin_slice[j_bwd] * couplings[j_bwd * nn + d];
    }
    return h;
}

/* ss/mcmc/sweep.rs:220-258 via ss/parallel.rs:15-41 (slot -> system spin slice,
 * system RNG, slot temperature / table row) and sweep_sites :51-97 (sites
 * 0..N-1 in order, one draw per site). */
void orc_metropolis_sweep(const orc_lattice *lat, int8_t *spins, const float *couplings,
                          const float *temperatures, const uint64_t *system_ids, size_t n_slots,
                          orc_rng *rngs, const orc_lookup *lookup) {
    size_t n = lat->n_spins;
    for (size_t slot = 0; slot < n_slots; ++slot) {
        size_t system_id = (size_t)system_ids[slot];
        int8_t *spin_slice = spins + system_id * n;
        orc_rng *rng = &rngs[system_id];
        float temp = temperatures[slot];
        if (lookup) {
            /* attempt_flip_lookup, sweep.rs:170-185 */
            for (size_t i = 0; i < n; ++i) {
                float h = local_field(lat, spin_slice, couplings, i);
                int32_t energy_change = (int32_t)((float)(-spin_slice[i]) * h);
                uint32_t draw = orc_rng_next_u32(rng) >> (32 - F32_UNIFORM_BITS);
                if (draw < orc_lookup_get(lookup, slot, energy_change)) spin_slice[i] = -spin_slice[i];
            }
        } else {
            /* attempt_flip with the log threshold, sweep.rs:35-48,256 */
            for (size_t i = 0; i < n; ++i) {
                float h = local_field(lat, spin_slice, couplings, i);
                float si = (float)spin_slice[i];
                float eng_change = -si * h;
                float threshold = (temp / 2.0f) * logf(orc_rng_gen_f32(rng));
                if (eng_change >= threshold) spin_slice[i] = -spin_slice[i];
            }
        }
    }
}

/* ss/mcmc/sweep.rs:262-284 */
void orc_gibbs_sweep(const orc_lattice *lat, int8_t *spins, const float *couplings,
                     const float *temperatures, const uint64_t *system_ids, size_t n_slots,
                     orc_rng *rngs) {
    size_t n = lat->n_spins;
    for (size_t slot = 0; slot < n_slots; ++slot) {
        size_t system_id = (size_t)system_ids[slot];
        int8_t *spin_slice = spins + system_id * n;
        orc_rng *rng = &rngs[system_id];
        float temp = temperatures[slot];
        for (size_t i = 0; i < n; ++i) {
            float h = local_field(lat, spin_slice, couplings, i);
            float si = (float)spin_slice[i];
            float eng_change = -si * h;
            float u = orc_rng_gen_f32(rng);
            float threshold = (temp / 2.0f) * logf(u / (1.0f - u));
            if (eng_change >= threshold) spin_slice[i] = -spin_slice[i];
        }
    }
}

/* ========================================================================== */
/* Energies                                                                   */
/* ========================================================================== */

/* ss/spins/energy.rs:78-110: sequential f32 accumulation over (i, d); E/N;
 * magnetisation sums as i64; optional per-bond interactions (:14-40). */
void orc_compute_energies(const orc_lattice *lat, const int8_t *spins, const float *couplings,
                          size_t n_systems, float *energies, int64_t *magnetization_sums,
                          float *interactions) {
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    for (size_t r = 0; r < n_systems; ++r) {
        size_t base = r * n;
        float total = 0.0f;
        int64_t mag = 0;
        for (size_t i = 0; i < n; ++i) {
            int8_t spin = spins[base + i];
            mag += spin;
            float si = (float)spin;
            for (size_t d = 0; d < nn; ++d) {
                size_t j = lat->fwd[i * nn + d];
                float sj = (float)spins[base + j];
                float c = couplings[i * nn + d];
                float interaction = si * sj * c;
                if (interactions) interactions[(r * n + i) * nn + d] = interaction;
                total += interaction;
            }
        }
        energies[r] = total / (float)n;
        if (magnetization_sums) magnetization_sums[r] = mag;
    }
}

/* ========================================================================== */
/* Parallel tempering                                                         */
/* ========================================================================== */

/* ss/mcmc/tempering.rs:73-102 */
static orc_tempering_attempt attempt_edge(const float *energies, const float *temperatures,
                                          uint64_t *system_ids, size_t n_spins, orc_rng *rng,
                                          size_t temp_id) {
    float temp_1 = temperatures[temp_id];
    float temp_2 = temperatures[temp_id + 1];
    float energy_1 = energies[system_ids[temp_id]];
    float energy_2 = energies[system_ids[temp_id + 1]];
    orc_tempering_attempt attempt;
    attempt.edge = temp_id;
    attempt.left_system = system_ids[temp_id];
    attempt.right_system = system_ids[temp_id + 1];

    float delta = (float)n_spins * (energy_2 - energy_1) * (1.0f / temp_1 - 1.0f / temp_2);
    float log_rand = logf(orc_rng_gen_f32(rng));
    attempt.accepted = delta >= log_rand;
    if (attempt.accepted) {
        uint64_t tmp = system_ids[temp_id];
        system_ids[temp_id] = system_ids[temp_id + 1];
        system_ids[temp_id + 1] = tmp;
    }
    return attempt;
}

/* ss/mcmc/tempering.rs:20-42 */
size_t orc_parallel_tempering(const float *energies, const float *temperatures,
                              uint64_t *system_ids, size_t n_temps, size_t n_spins, orc_rng *rng,
                              orc_tempering_attempt *attempts) {
    if (n_temps < 2) return 0;
    size_t temp_id = (size_t)orc_rng_gen_range_usize(rng, 0, n_temps - 1);
    attempts[0] = attempt_edge(energies, temperatures, system_ids, n_spins, rng, temp_id);
    return 1;
}

/* ss/mcmc/tempering.rs:45-71 */
size_t orc_parallel_tempering_full_ladder(const float *energies, const float *temperatures,
                                          uint64_t *system_ids, size_t n_temps, size_t n_spins,
                                          orc_rng *rng, size_t first_parity,
                                          orc_tempering_attempt *attempts) {
    if (n_temps < 2) return 0;
    size_t count = 0;
    size_t parities[2] = {first_parity, 1 - first_parity};
    for (int p = 0; p < 2; ++p)
        for (size_t edge = parities[p]; edge < n_temps - 1; edge += 2)
            attempts[count++] = attempt_edge(energies, temperatures, system_ids, n_spins, rng, edge);
    return count;
}

/* ========================================================================== */
/* Union-find / DFS helpers                                                   */
/* ========================================================================== */

/* ss/clusters/utils.rs:68-74 (path halving) */
uint32_t orc_uf_find(uint32_t *parent, uint32_t x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

/* ss/clusters/utils.rs:77-91 (union by rank) */
void orc_uf_union(uint32_t *parent, uint8_t *rank, uint32_t x, uint32_t y) {
    uint32_t rx = orc_uf_find(parent, x), ry = orc_uf_find(parent, y);
    if (rx == ry) return;
    if (rank[rx] < rank[ry]) {
        parent[rx] = ry;
    } else {
        parent[ry] = rx;
        if (rank[rx] == rank[ry]) rank[rx] += 1;
    }
}

/* ss/clusters/utils.rs:283-293 + :260-281 (flatten, count per root) */
static void uf_flatten_counts(uint32_t *parent, size_t n, uint32_t *counts) {
    for (size_t i = 0; i < n; ++i) parent[i] = orc_uf_find(parent, (uint32_t)i);
    memset(counts, 0, sizeof(uint32_t) * n);
    for (size_t i = 0; i < n; ++i) counts[parent[i]] += 1;
}

void orc_uf_from_bonds(const orc_lattice *lat, const uint64_t *bond_sites, const uint64_t *bond_dims,
                       size_t n_bonds, uint32_t *parent, uint32_t *counts) {
    size_t n = lat->n_spins;
    uint8_t *rank = (uint8_t *)calloc(n, 1);
    for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
    /* activate_bonds visits (site, dim) in ascending order, utils.rs:219-236 */
    for (size_t i = 0; i < n; ++i)
        for (size_t d = 0; d < lat->n_neighbors; ++d)
            for (size_t b = 0; b < n_bonds; ++b)
                if (bond_sites[b] == i && bond_dims[b] == d)
                    orc_uf_union(parent, rank, (uint32_t)i, lat->fwd[i * lat->n_neighbors + d]);
    uf_flatten_counts(parent, n, counts);
    free(rank);
}

/* ss/clusters/utils.rs:297-303 */
void orc_uf_histogram(const uint32_t *counts, size_t n, uint64_t *hist) {
    for (size_t i = 0; i < n; ++i)
        if (counts[i] > 0) hist[counts[i]] += 1;
}

/* ss/clusters/utils.rs:305-316 */
void orc_top4_sizes(const uint32_t *counts, size_t n, uint32_t out[4]) {
    uint32_t top[4] = {0, 0, 0, 0}; /* ascending */
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = counts[i];
        if (c > top[0]) {
            top[0] = c;
            for (int a = 1; a < 4; ++a) /* insertion sort of 4 == sort_unstable result */
                for (int b = a; b > 0 && top[b - 1] > top[b]; --b) {
                    uint32_t t = top[b];
                    top[b] = top[b - 1];
                    top[b - 1] = t;
                }
        }
    }
    out[0] = top[3];
    out[1] = top[2];
    out[2] = top[1];
    out[3] = top[0];
}

/* ss/clusters/utils.rs:371-455: union-find that carries every site's displacement to its root;
 * a bond that closes a cycle with a non-zero net displacement marks the component as winding
 * around that axis of the torus (2D hypercubic lattices only, lattice.rs:31-38). */
typedef struct {
    uint32_t *parent;
    uint8_t *rank;
    int64_t *disp; /* [n][2] */
    uint8_t *wrap; /* [n][2] */
    int64_t shape[2];
    size_t n;
} winding_uf;

static winding_uf *winding_new(size_t n, size_t s0, size_t s1) {
    winding_uf *w = (winding_uf *)calloc(1, sizeof(*w));
    w->n = n;
    w->parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    for (size_t i = 0; i < n; ++i) w->parent[i] = (uint32_t)i;
    w->rank = (uint8_t *)calloc(n + 1, 1);
    w->disp = (int64_t *)calloc(2 * n + 2, sizeof(int64_t));
    w->wrap = (uint8_t *)calloc(2 * n + 2, 1);
    w->shape[0] = (int64_t)s0;
    w->shape[1] = (int64_t)s1;
    return w;
}

static void winding_free(winding_uf *w) {
    if (!w) return;
    free(w->parent);
    free(w->rank);
    free(w->disp);
    free(w->wrap);
    free(w);
}

/* utils.rs:390-402: full path compression, displacements re-based on the root */
static uint32_t winding_find(winding_uf *w, uint32_t site, int64_t to_root[2]) {
    uint32_t parent = w->parent[site];
    if (parent == site) {
        to_root[0] = to_root[1] = 0;
        return site;
    }
    int64_t own0 = w->disp[2 * site], own1 = w->disp[2 * site + 1], up[2];
    uint32_t root = winding_find(w, parent, up);
    to_root[0] = own0 + up[0];
    to_root[1] = own1 + up[1];
    w->parent[site] = root;
    w->disp[2 * site] = to_root[0];
    w->disp[2 * site + 1] = to_root[1];
    return root;
}

/* utils.rs:404-438 */
static void winding_union_forward(winding_uf *w, uint32_t site, uint32_t neighbor, size_t dim) {
    int64_t ds[2], dn[2];
    uint32_t root_site = winding_find(w, site, ds);
    uint32_t root_neighbor = winding_find(w, neighbor, dn);
    int64_t rel[2] = {ds[0] - dn[0], ds[1] - dn[1]};
    rel[dim] += 1;
    if (root_site == root_neighbor) {
        for (int axis = 0; axis < 2; ++axis) w->wrap[2 * root_site + axis] |= (uint8_t)(rel[axis] != 0);
        return;
    }
    uint8_t wrap0 = w->wrap[2 * root_site] | w->wrap[2 * root_neighbor];
    uint8_t wrap1 = w->wrap[2 * root_site + 1] | w->wrap[2 * root_neighbor + 1];
    if (w->rank[root_site] < w->rank[root_neighbor]) {
        w->parent[root_site] = root_neighbor;
        w->disp[2 * root_site] = -rel[0];
        w->disp[2 * root_site + 1] = -rel[1];
        w->wrap[2 * root_neighbor] = wrap0;
        w->wrap[2 * root_neighbor + 1] = wrap1;
        return;
    }
    w->parent[root_neighbor] = root_site;
    w->disp[2 * root_neighbor] = rel[0];
    w->disp[2 * root_neighbor + 1] = rel[1];
    w->wrap[2 * root_site] = wrap0;
    w->wrap[2 * root_site + 1] = wrap1;
    if (w->rank[root_site] == w->rank[root_neighbor]) w->rank[root_site] += 1;
}

/* utils.rs:440-454 */
static void winding_result(winding_uf *w, int *x, int *y) {
    int64_t scratch[2];
    for (size_t i = 0; i < w->n; ++i) winding_find(w, (uint32_t)i, scratch);
    *x = *y = 0;
    for (size_t i = 0; i < w->n; ++i) {
        if (w->parent[i] != i) continue;
        *x |= w->wrap[2 * i];
        *y |= w->wrap[2 * i + 1];
    }
}

/* ss/clusters/utils.rs:317-369: what one observed bond graph reports */
typedef struct {
    uint32_t top4[4];
    uint32_t active_bonds;
    uint32_t large_components;
    uint8_t winding_x, winding_y, observed;
} graph_slot;

typedef struct {
    uint32_t active_bonds;
    winding_uf *winding;
} bond_metrics;

static void metrics_begin(bond_metrics *m, const orc_lattice *lat) { /* :335-342 */
    m->active_bonds = 0;
    m->winding = lat->has_square_shape ? winding_new(lat->n_spins, lat->shape[0], lat->shape[1]) : NULL;
}

static void metrics_record(bond_metrics *m, const orc_lattice *lat, size_t site, size_t dim) { /* :345-350 */
    m->active_bonds += 1;
    if (m->winding)
        winding_union_forward(m->winding, (uint32_t)site, lat->fwd[site * lat->n_neighbors + dim], dim);
}

static void metrics_finish(bond_metrics *m, const uint32_t *counts, size_t n, graph_slot *out) { /* :352-368 */
    int wx = 0, wy = 0;
    if (m->winding) winding_result(m->winding, &wx, &wy);
    uint32_t threshold = (uint32_t)ceil((double)n * 0.05);
    memset(out, 0, sizeof(*out));
    orc_top4_sizes(counts, n, out->top4);
    out->active_bonds = m->active_bonds;
    out->winding_x = (uint8_t)wx;
    out->winding_y = (uint8_t)wy;
    for (size_t i = 0; i < n; ++i) out->large_components += counts[i] >= threshold;
    out->observed = 1;
    winding_free(m->winding);
    m->winding = NULL;
}

/* Known-answer entry: bonds (site, dim) applied in the given order; out = {active_bonds, top4[4],
 * large_components, winding_x, winding_y} (utils.rs tests :736-781). */
void orc_graph_observation(const orc_lattice *lat, const uint64_t *bond_sites, const uint64_t *bond_dims,
                           size_t n_bonds, uint32_t out[9]) {
    size_t n = lat->n_spins;
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *rank = (uint8_t *)calloc(n + 1, 1);
    bond_metrics m;
    metrics_begin(&m, lat);
    for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
    for (size_t b = 0; b < n_bonds; ++b) {
        orc_uf_union(parent, rank, (uint32_t)bond_sites[b],
                     lat->fwd[bond_sites[b] * lat->n_neighbors + bond_dims[b]]);
        metrics_record(&m, lat, (size_t)bond_sites[b], (size_t)bond_dims[b]);
    }
    uf_flatten_counts(parent, n, counts);
    graph_slot g;
    metrics_finish(&m, counts, n, &g);
    out[0] = g.active_bonds;
    for (int k = 0; k < 4; ++k) out[1 + k] = g.top4[k];
    out[5] = g.large_components;
    out[6] = g.winding_x;
    out[7] = g.winding_y;
    out[8] = g.observed;
    free(parent);
    free(counts);
    free(rank);
}

/* Optional outputs of one cluster call, indexed by slot (FK: temperature slot of the system;
 * overlap moves: t * n_pairs + g).  csd / top4 are cleared by the caller before the call
 * (mod.rs:442-455,604-626); snap / blue_snap keep what an earlier call left when a slot writes
 * nothing (a single-cluster move that finds no seed, overlap.rs:246-248). */
typedef struct {
    int observe;         /* ClusterAction::Observe: measure the bond graph, leave the spins */
    uint64_t *csd;       /* [n_slots][N+1] or NULL */
    uint32_t *top4;      /* [n_slots][4] or NULL */
    graph_slot *obs;     /* [n_slots] or NULL */
    uint32_t *snap;      /* [n_slots][N] or NULL */
    uint32_t *blue_snap; /* [n_slots][N] or NULL (CMR) */
    int8_t *spin_snap;   /* [n_slots][2][N], with snap */
    uint64_t *sid_snap;  /* [n_slots][2], with snap */
} cluster_out;

/* the per-slot outputs every union-find path shares (overlap.rs:257-291) */
static void emit_counts(const cluster_out *out, size_t slot, const uint32_t *parent, const uint32_t *counts,
                        size_t n, int with_snap) {
    if (!out) return;
    if (out->csd) orc_uf_histogram(counts, n, out->csd + slot * (n + 1));
    if (out->top4) orc_top4_sizes(counts, n, out->top4 + slot * 4);
    if (with_snap && out->snap) memcpy(out->snap + slot * n, parent, sizeof(uint32_t) * n);
}

static void emit_spins(const cluster_out *out, size_t slot, const int8_t *a, const int8_t *b, uint64_t sa,
                       uint64_t sb, size_t n) { /* overlap.rs:211-221,411-419,680-688 */
    if (!out || !out->snap) return;
    memcpy(out->spin_snap + (slot * 2 + 0) * n, a, n);
    memcpy(out->spin_snap + (slot * 2 + 1) * n, b, n);
    out->sid_snap[slot * 2 + 0] = sa;
    out->sid_snap[slot * 2 + 1] = sb;
}

/* ========================================================================== */
/* Realisation state                                                          */
/* ========================================================================== */

/* ss/simulation/realization.rs:21-29 */
typedef struct {
    uint64_t *edge_attempts;
    uint64_t *edge_acceptances;
    uint64_t *round_trips;
    uint8_t *trip_state;
    size_t next_parity;
    size_t cold_slot;
    size_t hot_slot;
} pt_state;

/* ss/simulation/realization.rs:132-149 */
typedef struct {
    float *couplings;
    int8_t *spins;
    float *temperatures;
    uint64_t *system_ids;
    orc_rng *rngs;
    orc_rng *pair_rngs;
    pt_state pt;
    float *energies;
} realization;

/* statistics/stats.rs:2-35 */
typedef struct {
    size_t count;
    double *aggregate;
    uint32_t power;
} statistics;

/* GraphObservationAccum / GraphObservationSummary (simulation/mod.rs:17-98, results.rs:22-31) */
typedef struct {
    int enabled, present, has_winding;
    uint64_t *count;      /* [n_temps] observation_count */
    uint64_t *csd;        /* [n_temps][N+1] cluster_size_counts */
    uint64_t *top4_sum;   /* [n_temps][4] */
    uint64_t *bonds;      /* [n_temps] */
    uint64_t *winding;    /* [n_temps][4]: x, y, either, both */
    uint64_t *large;      /* [n_temps] */
    double *top4_frac;    /* [n_temps][4] top_four_component_fractions */
    double *bond_density; /* [n_temps] */
    double *large_mean;   /* [n_temps] */
    double *winding_frac; /* [n_temps][4] */
} graph_summary;

typedef struct {
    uint64_t sweep_id;
    int mode_idx, has_blue;
    uint32_t *cluster_ids; /* [n_temps][N] */
    uint32_t *blue_ids;    /* [n_temps][N] */
    int8_t *spins;         /* [n_temps][2][N] */
    uint64_t *system_ids;  /* [n_temps][2] */
} cluster_snapshot;

static void summary_enable(graph_summary *g, size_t n_temps, size_t n) { /* mod.rs:27-36 */
    if (g->enabled) return;
    g->enabled = 1;
    g->count = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    g->csd = (uint64_t *)calloc(n_temps * (n + 1) + 1, sizeof(uint64_t));
    g->top4_sum = (uint64_t *)calloc(4 * n_temps + 1, sizeof(uint64_t));
    g->bonds = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    g->winding = (uint64_t *)calloc(4 * n_temps + 1, sizeof(uint64_t));
    g->large = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    g->top4_frac = (double *)calloc(4 * n_temps + 1, sizeof(double));
    g->bond_density = (double *)calloc(n_temps + 1, sizeof(double));
    g->large_mean = (double *)calloc(n_temps + 1, sizeof(double));
    g->winding_frac = (double *)calloc(4 * n_temps + 1, sizeof(double));
}

static void summary_push(graph_summary *g, size_t temp, const uint64_t *sizes, size_t n, const graph_slot *slot) {
    if (!slot->observed) return; /* mod.rs:38-55 */
    g->count[temp] += 1;
    for (size_t k = 0; k <= n; ++k) g->csd[temp * (n + 1) + k] += sizes[k];
    for (int k = 0; k < 4; ++k) g->top4_sum[temp * 4 + k] += slot->top4[k];
    g->bonds[temp] += slot->active_bonds;
    g->winding[temp * 4 + 0] += slot->winding_x;
    g->winding[temp * 4 + 1] += slot->winding_y;
    g->winding[temp * 4 + 2] += (uint64_t)(slot->winding_x || slot->winding_y);
    g->winding[temp * 4 + 3] += (uint64_t)(slot->winding_x && slot->winding_y);
    g->large[temp] += slot->large_components;
}

static void summary_finish(graph_summary *g, size_t n_temps, size_t n, size_t nn, int has_winding) {
    if (!g->enabled) return; /* mod.rs:57-97 */
    g->has_winding = has_winding;
    for (size_t t = 0; t < n_temps; ++t) g->present |= g->count[t] != 0;
    for (size_t t = 0; t < n_temps; ++t) {
        if (g->count[t] == 0) continue;
        double c = (double)g->count[t];
        for (int k = 0; k < 4; ++k) {
            g->top4_frac[t * 4 + k] = (double)g->top4_sum[t * 4 + k] / (c * (double)n);
            if (has_winding) g->winding_frac[t * 4 + k] = (double)g->winding[t * 4 + k] / c;
        }
        g->bond_density[t] = (double)g->bonds[t] / (c * (double)n * (double)nn);
        g->large_mean[t] = (double)g->large[t] / c;
    }
}

static void summary_free(graph_summary *g) {
    free(g->count);
    free(g->csd);
    free(g->top4_sum);
    free(g->bonds);
    free(g->winding);
    free(g->large);
    free(g->top4_frac);
    free(g->bond_density);
    free(g->large_mean);
    free(g->winding_frac);
    memset(g, 0, sizeof(*g));
}

/* SweepResult fields on the path (statistics/results.rs:56-77, overlap.rs:8-39) */
typedef struct {
    double *moments;      /* [5][n_temps] mags, mags2, mags4, energies, energies2 */
    double *overlap;      /* [6][n_temps] q, q2, q4, ql, ql2, ql4 (NULL if n_pairs == 0) */
    uint64_t *histogram;  /* [n_temps][N+1] */
    double *ql_at_q_sum;  /* [n_temps][N+1] */
    double *ql2_at_q_sum; /* [n_temps][N+1] */
    double *mags2_tau;    /* [n_temps] or NULL */
    double *overlap2_tau; /* [n_temps] or NULL */
    size_t n_equil;       /* equilibration checkpoints (statistics/equilibration.rs) */
    uint64_t *equil_sweeps;
    double *equil_energy; /* [n_equil][n_temps] */
    double *equil_ql;     /* [n_equil][n_temps] */
    /* ClusterStats (statistics/results.rs:12-20) */
    uint64_t *fk_csd;      /* [n_temps][N+1] or NULL */
    uint64_t *overlap_csd; /* [n_modes][n_temps][N+1] or NULL */
    double *top_sizes;     /* [n_modes][n_temps][4] or NULL */
    size_t n_modes;
    /* per-disorder graph observations (results.rs:22-45): fk, houdayer, jorg, cmr_blue */
    graph_summary observations[4];
    /* ClusterSnapshot list (results.rs:47-54), realisation 0 only */
    size_t n_snapshots;
    cluster_snapshot *snapshots;
} sweep_result;

struct orc_sim {
    orc_lattice *lattice;
    size_t n_replicas, n_temps, n_realizations;
    uint64_t constructor_seed;
    uint64_t first_realization; /* global index of realisation 0 (disorder sharding) */
    realization *realizations;
    sweep_result *results;
    char error[256];
};

/* ss/simulation/realization.rs:92-107 */
static void extreme_temperature_slots(const float *temperatures, size_t n_temps, size_t *cold,
                                      size_t *hot) {
    *cold = 0;
    *hot = 0;
    for (size_t slot = 1; slot < n_temps; ++slot) {
        if (temperatures[slot] < temperatures[*cold]) *cold = slot;
        if (temperatures[slot] > temperatures[*hot]) *hot = slot;
    }
}

/* ss/simulation/realization.rs:46-71 */
static void pt_reset(pt_state *pt, size_t n_replicas, size_t n_temps, const uint64_t *system_ids,
                     const float *temperatures) {
    size_t n_edges = n_temps > 0 ? n_temps - 1 : 0;
    memset(pt->edge_attempts, 0, sizeof(uint64_t) * (n_edges + 1));
    memset(pt->edge_acceptances, 0, sizeof(uint64_t) * (n_edges + 1));
    memset(pt->round_trips, 0, sizeof(uint64_t) * n_replicas * n_temps);
    memset(pt->trip_state, 0, n_replicas * n_temps);
    pt->next_parity = 0;
    extreme_temperature_slots(temperatures, n_temps, &pt->cold_slot, &pt->hot_slot);
    if (n_temps == 0) return;
    for (size_t replica = 0; replica < n_replicas; ++replica)
        pt->trip_state[system_ids[replica * n_temps + pt->hot_slot]] = 1;
}

/* ss/simulation/realization.rs:109-120 */
static void pt_record_arrival(pt_state *pt, size_t system, size_t slot) {
    if (slot == pt->hot_slot) {
        if (pt->trip_state[system] == 2) pt->round_trips[system] += 1;
        pt->trip_state[system] = 1;
        return;
    }
    if (slot == pt->cold_slot && pt->trip_state[system] == 1) pt->trip_state[system] = 2;
}

/* ss/simulation/realization.rs:73-82 */
static void pt_record_attempt(pt_state *pt, orc_tempering_attempt attempt) {
    pt->edge_attempts[attempt.edge] += 1;
    if (!attempt.accepted) return;
    pt->edge_acceptances[attempt.edge] += 1;
    pt_record_arrival(pt, (size_t)attempt.left_system, (size_t)attempt.edge + 1);
    pt_record_arrival(pt, (size_t)attempt.right_system, (size_t)attempt.edge);
}

/* Replays a list of attempts through PtState (realization.rs:46-120) starting
 * from the identity permutation; used to pin the round-trip bookkeeping on the
 * reference's own test (realization.rs:285-302). */
void orc_pt_replay(size_t n_replicas, size_t n_temps, const float *temps,
                   const orc_tempering_attempt *attempts, size_t n_attempts,
                   uint64_t *edge_attempts, uint64_t *edge_acceptances, uint64_t *round_trips) {
    size_t n_systems = n_replicas * n_temps;
    pt_state pt;
    pt.edge_attempts = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    pt.edge_acceptances = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    pt.round_trips = (uint64_t *)calloc(n_systems + 1, sizeof(uint64_t));
    pt.trip_state = (uint8_t *)calloc(n_systems + 1, 1);
    uint64_t *system_ids = (uint64_t *)malloc(sizeof(uint64_t) * (n_systems + 1));
    float *temperatures = (float *)malloc(sizeof(float) * (n_systems + 1));
    for (size_t i = 0; i < n_systems; ++i) {
        system_ids[i] = i;
        temperatures[i] = temps[i % n_temps];
    }
    pt_reset(&pt, n_replicas, n_temps, system_ids, temperatures);
    for (size_t a = 0; a < n_attempts; ++a) pt_record_attempt(&pt, attempts[a]);
    size_t n_edges = n_temps > 0 ? n_temps - 1 : 0;
    memcpy(edge_attempts, pt.edge_attempts, sizeof(uint64_t) * n_edges);
    memcpy(edge_acceptances, pt.edge_acceptances, sizeof(uint64_t) * n_edges);
    memcpy(round_trips, pt.round_trips, sizeof(uint64_t) * n_systems);
    free(pt.edge_attempts);
    free(pt.edge_acceptances);
    free(pt.round_trips);
    free(pt.trip_state);
    free(system_ids);
    free(temperatures);
}

/* ss/simulation/realization.rs:213-246 (reset); new() :155-211 does the same
 * after allocating, so both go through here. */
static void realization_reset(const orc_lattice *lat, realization *real, size_t n_replicas,
                              size_t n_temps, uint64_t base_seed) {
    size_t n = lat->n_spins, n_systems = n_replicas * n_temps;
    for (size_t i = 0; i < n_systems; ++i) {
        orc_rng_seed_from_u64(&real->rngs[i], orc_child_seed(base_seed, ORC_SYSTEM_SEED_DOMAIN, i));
        for (size_t j = 0; j < n; ++j)
            real->spins[i * n + j] = orc_rng_gen_f32(&real->rngs[i]) < 0.5f ? -1 : 1;
    }
    for (size_t i = 0; i < n_systems; ++i) real->system_ids[i] = i;
    size_t n_pairs = n_replicas / 2;
    for (size_t i = 0; i < n_temps * n_pairs; ++i)
        orc_rng_seed_from_u64(&real->pair_rngs[i], orc_child_seed(base_seed, ORC_PAIR_SEED_DOMAIN, i));
    orc_compute_energies(lat, real->spins, real->couplings, n_systems, real->energies, NULL, NULL);
    pt_reset(&real->pt, n_replicas, n_temps, real->system_ids, real->temperatures);
}

static void realization_alloc(const orc_lattice *lat, realization *real, const float *couplings,
                              const float *temps, size_t n_replicas, size_t n_temps) {
    size_t n = lat->n_spins, nn = lat->n_neighbors, n_systems = n_replicas * n_temps;
    real->couplings = (float *)malloc(sizeof(float) * (n * nn + 1));
    memcpy(real->couplings, couplings, sizeof(float) * n * nn);
    real->spins = (int8_t *)malloc(n_systems * n + 1);
    real->temperatures = (float *)malloc(sizeof(float) * (n_systems + 1));
    for (size_t r = 0; r < n_replicas; ++r) /* temps.repeat(n_replicas), realization.rs:166 */
        memcpy(real->temperatures + r * n_temps, temps, sizeof(float) * n_temps);
    real->system_ids = (uint64_t *)malloc(sizeof(uint64_t) * (n_systems + 1));
    real->rngs = (orc_rng *)malloc(sizeof(orc_rng) * (n_systems + 1));
    real->pair_rngs = (orc_rng *)malloc(sizeof(orc_rng) * (n_temps * (n_replicas / 2) + 1));
    real->energies = (float *)malloc(sizeof(float) * (n_systems + 1));
    real->pt.edge_attempts = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    real->pt.edge_acceptances = (uint64_t *)calloc(n_temps + 1, sizeof(uint64_t));
    real->pt.round_trips = (uint64_t *)calloc(n_systems + 1, sizeof(uint64_t));
    real->pt.trip_state = (uint8_t *)calloc(n_systems + 1, 1);
}

static void realization_free(realization *real) {
    free(real->couplings);
    free(real->spins);
    free(real->temperatures);
    free(real->system_ids);
    free(real->rngs);
    free(real->pair_rngs);
    free(real->energies);
    free(real->pt.edge_attempts);
    free(real->pt.edge_acceptances);
    free(real->pt.round_trips);
    free(real->pt.trip_state);
}

/* ========================================================================== */
/* Statistics                                                                 */
/* ========================================================================== */

static void stat_init(statistics *st, size_t n_temps, uint32_t power) {
    st->count = 0;
    st->aggregate = (double *)calloc(n_temps + 1, sizeof(double));
    st->power = power;
}

/* statistics/stats.rs:17-27; powi(2) of an f64 is the exact product v*v. */
static void stat_update(statistics *st, const float *values, size_t n_temps) {
    st->count += 1;
    for (size_t t = 0; t < n_temps; ++t) {
        double v = (double)values[t];
        st->aggregate[t] += (st->power == 1) ? v : v * v;
    }
}

/* statistics/stats.rs:29-35 */
static void stat_average(const statistics *st, double *out, size_t n_temps) {
    if (st->count == 0) {
        memcpy(out, st->aggregate, sizeof(double) * n_temps);
        return;
    }
    double c = (double)st->count;
    for (size_t t = 0; t < n_temps; ++t) out[t] = st->aggregate[t] / c;
}

/* ========================================================================== */
/* Overlap-cluster moves (Houdayer group_size 2, Joerg)                        */
/* ========================================================================== */

/* ss/clusters/utils.rs:107-119 */
static int64_t find_seed(size_t n_spins, orc_rng *rng, const int8_t *a, const int8_t *b, int joerg) {
    for (int probe = 0; probe < 64; ++probe) {
        size_t i = (size_t)orc_rng_gen_range_usize(rng, 0, n_spins);
        int active = joerg ? (a[i] != b[i]) : ((int)a[i] + (int)b[i] == 0);
        if (active) return (int64_t)i;
    }
    return -1;
}

/* ss/clusters/overlap.rs:35-56 (build_tasks), :153-339 (houdayer_step, group
 * size 2) and :348-553 (jorg_step).  Without outputs the single-cluster variant is find_seed +
 * dfs_cluster (utils.rs:125-149); the SW variant, the Observe action and every call that collects
 * sizes or a snapshot take the union-find path (`use_uf`, :179-183,:376-380), whose
 * single-cluster form draws all bonds first and looks for the seed afterwards. */
static void overlap_update(const orc_lattice *lat, realization *real, size_t n_replicas,
                           size_t n_temps, int mode, int cluster_mode, const cluster_out *out) {
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    size_t n_pairs = n_replicas / 2, group_size = 2, n_groups = n_replicas / group_size;
    int joerg = (mode == ORC_OC_JORG);
    int wolff = cluster_mode == ORC_CLUSTER_WOLFF;
    int observe = out && out->observe;
    int collects = out && (out->csd || out->top4 || out->snap);
    int use_uf = observe || !wolff || collects;
    uint64_t *systems = (uint64_t *)malloc(sizeof(uint64_t) * (n_temps * n_replicas + 1));
    for (size_t t = 0; t < n_temps; ++t) {
        for (size_t k = 0; k < n_replicas; ++k)
            systems[t * n_replicas + k] = real->system_ids[k * n_temps + t];
        orc_rng_shuffle_usize(&real->pair_rngs[t * n_pairs], systems + t * n_replicas, n_replicas);
    }
    uint8_t *in_cluster = (uint8_t *)malloc(n + 1);
    size_t *stack = (size_t *)malloc(sizeof(size_t) * (n + 1));
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *rank = (uint8_t *)malloc(n + 1);

    for (size_t task = 0; task < n_temps * n_groups; ++task) {
        size_t t = task / n_groups, g = task % n_groups, slot = t * n_pairs + g;
        const uint64_t *grp = systems + t * n_replicas + g * group_size;
        orc_rng *rng = &real->pair_rngs[t * n_pairs + g];
        float temp = real->temperatures[t];
        int8_t *a = real->spins + (size_t)grp[0] * n;
        int8_t *b = real->spins + (size_t)grp[1] * n;
        emit_spins(out, slot, a, b, grp[0], grp[1], n);
#define ACTIVE(i) (joerg ? (a[i] != b[i]) : ((int)a[i] + (int)b[i] == 0))
        if (!use_uf) {
            int64_t seed = find_seed(n, rng, a, b, joerg);
            if (seed < 0) continue;
            memset(in_cluster, 0, n);
            size_t top = 0;
            in_cluster[seed] = 1;
            stack[top++] = (size_t)seed;
            while (top > 0) {
                size_t site = stack[--top];
                for (size_t d = 0; d < nn; ++d) {
                    for (int dir = 0; dir < 2; ++dir) {
                        size_t nb = dir == 0 ? lat->fwd[site * nn + d] : lat->bwd[site * nn + d];
                        if (in_cluster[nb]) continue;
                        int add;
                        if (!joerg) {
                            add = ACTIVE(site) && ACTIVE(nb);
                        } else {
                            add = 0;
                            if (ACTIVE(nb)) {
                                float coupling = dir == 0 ? real->couplings[site * nn + d]
                                                          : real->couplings[nb * nn + d];
                                float inter = (float)a[site] * (float)a[nb] * coupling;
                                if (inter > 0.0f)
                                    add = orc_rng_gen_f32(rng) < 1.0f - expf(-4.0f * inter / temp);
                            }
                        }
                        if (add) {
                            in_cluster[nb] = 1;
                            stack[top++] = nb;
                        }
                    }
                }
            }
            for (size_t i = 0; i < n; ++i)
                if (in_cluster[i]) {
                    a[i] = (int8_t)-a[i];
                    b[i] = (int8_t)-b[i];
                }
        } else {
            bond_metrics metrics;
            int with_metrics = out && out->obs;
            if (with_metrics) metrics_begin(&metrics, lat);
            for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
            memset(rank, 0, n);
            for (size_t i = 0; i < n; ++i)
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    int bond = 0;
                    if (ACTIVE(i) && ACTIVE(j)) {
                        if (!joerg) {
                            bond = 1;
                        } else {
                            float inter = (float)a[i] * (float)a[j] * real->couplings[i * nn + d];
                            if (inter > 0.0f)
                                bond = orc_rng_gen_f32(rng) < 1.0f - expf(-4.0f * inter / temp);
                        }
                    }
                    if (bond) {
                        orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                        if (with_metrics) metrics_record(&metrics, lat, i, d);
                    }
                }
            if (wolff) { /* overlap.rs:245-272,446-472 */
                if (with_metrics) winding_free(metrics.winding); /* not reported by this branch */
                int64_t seed = find_seed(n, rng, a, b, joerg);
                if (seed < 0) continue;
                uint32_t seed_root = orc_uf_find(parent, (uint32_t)seed);
                for (size_t i = 0; i < n; ++i)
                    if (orc_uf_find(parent, (uint32_t)i) == seed_root) {
                        a[i] = (int8_t)-a[i];
                        b[i] = (int8_t)-b[i];
                    }
                if (collects) {
                    uf_flatten_counts(parent, n, counts);
                    emit_counts(out, slot, parent, counts, n, 1);
                }
                continue;
            }
            uf_flatten_counts(parent, n, counts);
            emit_counts(out, slot, parent, counts, n, 1);
            if (with_metrics) metrics_finish(&metrics, counts, n, &out->obs[slot]);
            if (observe) continue;
            memset(rank, 0xFF, n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t root = parent[i];
                if (counts[root] > 1 && rank[root] == 0xFF)
                    rank[root] = (uint8_t)(orc_rng_gen_f32(rng) < 0.5f);
            }
            for (size_t i = 0; i < n; ++i)
                if (rank[parent[i]] == 1) {
                    a[i] = (int8_t)-a[i];
                    b[i] = (int8_t)-b[i];
                }
        }
#undef ACTIVE
    }
    free(systems);
    free(in_cluster);
    free(stack);
    free(parent);
    free(counts);
    free(rank);
}

/* ss/clusters/overlap.rs:153-339 with group_size N > 2 ("houdN", flagged experimental upstream):
 * groups of N consecutive replicas of the shuffled order, active sites = the N spins sum to
 * zero, deterministic bonds between active neighbours, every replica of the group flips. */
static void houdayer_n_update(const orc_lattice *lat, realization *real, size_t n_replicas, size_t n_temps,
                              size_t group_size, int cluster_mode, const cluster_out *out) {
    int collects = out && (out->csd || out->top4 || out->snap);
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    size_t n_pairs = n_replicas / 2, n_groups = n_replicas / group_size;
    uint64_t *systems = (uint64_t *)malloc(sizeof(uint64_t) * (n_temps * n_replicas + 1));
    for (size_t t = 0; t < n_temps; ++t) {
        for (size_t k = 0; k < n_replicas; ++k)
            systems[t * n_replicas + k] = real->system_ids[k * n_temps + t];
        orc_rng_shuffle_usize(&real->pair_rngs[t * n_pairs], systems + t * n_replicas, n_replicas);
    }
    uint8_t *active = (uint8_t *)malloc(n + 1);
    uint8_t *in_cluster = (uint8_t *)malloc(n + 1);
    size_t *stack = (size_t *)malloc(sizeof(size_t) * (n + 1));
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *rank = (uint8_t *)malloc(n + 1);
    for (size_t task = 0; task < n_temps * n_groups; ++task) {
        size_t t = task / n_groups, g = task % n_groups, slot = t * n_pairs + g;
        const uint64_t *grp = systems + t * n_replicas + g * group_size;
        orc_rng *rng = &real->pair_rngs[t * n_pairs + g];
        emit_spins(out, slot, real->spins + (size_t)grp[0] * n, real->spins + (size_t)grp[1] * n, grp[0],
                   grp[1], n); /* the group's first two systems, :211-221 */
        for (size_t i = 0; i < n; ++i) { /* is_active, :199-205 (the spins do not change before the flips) */
            int sum = 0;
            for (size_t k = 0; k < group_size; ++k) sum += real->spins[(size_t)grp[k] * n + i];
            active[i] = sum == 0;
        }
        memset(in_cluster, 0, n);
        if (cluster_mode == ORC_CLUSTER_WOLFF && collects) { /* union-find form, :231-272 */
            for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
            memset(rank, 0, n);
            for (size_t i = 0; i < n; ++i)
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    if (active[i] && active[j]) orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                }
            int64_t seed = -1;
            for (int probe = 0; probe < 64; ++probe) {
                size_t i = (size_t)orc_rng_gen_range_usize(rng, 0, n);
                if (active[i]) {
                    seed = (int64_t)i;
                    break;
                }
            }
            if (seed < 0) continue;
            uf_flatten_counts(parent, n, counts);
            for (size_t i = 0; i < n; ++i) in_cluster[i] = parent[i] == parent[seed];
            emit_counts(out, slot, parent, counts, n, 1);
        } else if (cluster_mode == ORC_CLUSTER_WOLFF) {
            int64_t seed = -1;
            for (int probe = 0; probe < 64; ++probe) { /* find_seed, utils.rs:107-119 */
                size_t i = (size_t)orc_rng_gen_range_usize(rng, 0, n);
                if (active[i]) {
                    seed = (int64_t)i;
                    break;
                }
            }
            if (seed < 0) continue;
            size_t top = 0;
            in_cluster[seed] = 1;
            stack[top++] = (size_t)seed;
            while (top > 0) {
                size_t site = stack[--top];
                for (size_t d = 0; d < nn; ++d)
                    for (int dir = 0; dir < 2; ++dir) {
                        size_t nb = dir == 0 ? lat->fwd[site * nn + d] : lat->bwd[site * nn + d];
                        if (!in_cluster[nb] && active[site] && active[nb]) {
                            in_cluster[nb] = 1;
                            stack[top++] = nb;
                        }
                    }
            }
        } else {
            for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
            memset(rank, 0, n);
            for (size_t i = 0; i < n; ++i)
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    if (active[i] && active[j]) orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                }
            uf_flatten_counts(parent, n, counts);
            emit_counts(out, slot, parent, counts, n, 1);
            memset(rank, 0xFF, n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t root = parent[i];
                if (counts[root] > 1 && rank[root] == 0xFF) rank[root] = (uint8_t)(orc_rng_gen_f32(rng) < 0.5f);
            }
            for (size_t i = 0; i < n; ++i) in_cluster[i] = rank[parent[i]] == 1;
        }
        for (size_t i = 0; i < n; ++i)
            if (in_cluster[i])
                for (size_t k = 0; k < group_size; ++k) {
                    int8_t *sp = real->spins + (size_t)grp[k] * n + i;
                    *sp = (int8_t)-*sp;
                }
    }
    free(systems);
    free(active);
    free(in_cluster);
    free(stack);
    free(parent);
    free(counts);
    free(rank);
}

/* ss/clusters/overlap.rs:555-971: CMR two-phase move (blue clusters of doubly satisfied bonds with
 * p = 1 - r^2, then grey clusters = blue + singly satisfied "red" bonds with p = 1 - r,
 * r = exp(-2|J|/T)).  Update action without observers: cluster_mode sw = union-find path
 * (:690-843), wolff = the two-phase depth-first search (:845-965). */
static void cmr_update(const orc_lattice *lat, realization *real, size_t n_replicas, size_t n_temps,
                       int cluster_mode, const cluster_out *out) {
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    size_t n_pairs = n_replicas / 2;
    int wolff = cluster_mode == ORC_CLUSTER_WOLFF;
    int observe = out && out->observe;
    int use_uf = observe || !wolff || (out && (out->csd || out->top4 || out->snap)); /* :640-644 */
    uint64_t *systems = (uint64_t *)malloc(sizeof(uint64_t) * (n_temps * n_replicas + 1));
    for (size_t t = 0; t < n_temps; ++t) { /* build_tasks, :35-56 */
        for (size_t k = 0; k < n_replicas; ++k)
            systems[t * n_replicas + k] = real->system_ids[k * n_temps + t];
        orc_rng_shuffle_usize(&real->pair_rngs[t * n_pairs], systems + t * n_replicas, n_replicas);
    }
    uint8_t *in_cluster = (uint8_t *)malloc(n + 1);
    size_t *stack = (size_t *)malloc(sizeof(size_t) * (n + 1));
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *rank = (uint8_t *)malloc(n + 1);
    uint8_t *flag = (uint8_t *)malloc(n + 1);

    for (size_t task = 0; task < n_temps * n_pairs; ++task) {
        size_t t = task / n_pairs, g = task % n_pairs, slot = t * n_pairs + g;
        const uint64_t *grp = systems + t * n_replicas + g * 2;
        orc_rng *rng = &real->pair_rngs[t * n_pairs + g];
        float temp = real->temperatures[t];
        int8_t *a = real->spins + (size_t)grp[0] * n;
        int8_t *b = real->spins + (size_t)grp[1] * n;
        emit_spins(out, slot, a, b, grp[0], grp[1], n);
#define SAT(s, i, j, c) ((float)(s)[i] * (float)(s)[j] * (c) > 0.0f)
        if (!use_uf) {
            size_t seed = (size_t)orc_rng_gen_range_usize(rng, 0, n);
            memset(in_cluster, 0, n);
            size_t top = 0;
            in_cluster[seed] = 1;
            stack[top++] = seed;
            while (top > 0) { /* phase 1: blue cluster, dfs_cluster order */
                size_t site = stack[--top];
                for (size_t d = 0; d < nn; ++d)
                    for (int dir = 0; dir < 2; ++dir) {
                        size_t nb = dir == 0 ? lat->fwd[site * nn + d] : lat->bwd[site * nn + d];
                        if (in_cluster[nb]) continue;
                        float c = dir == 0 ? real->couplings[site * nn + d] : real->couplings[nb * nn + d];
                        if (!SAT(a, site, nb, c) || !SAT(b, site, nb, c)) continue;
                        float r = expf(-2.0f * fabsf(c) / temp);
                        if (orc_rng_gen_f32(rng) < 1.0f - r * r) {
                            in_cluster[nb] = 1;
                            stack[top++] = nb;
                        }
                    }
            }
            for (size_t i = 0; i < n; ++i)
                if (in_cluster[i]) {
                    a[i] = (int8_t)-a[i];
                    b[i] = (int8_t)-b[i];
                }
            for (size_t i = 0; i < n; ++i) /* phase 2 starts from every blue site */
                if (in_cluster[i]) stack[top++] = i;
            while (top > 0) {
                size_t site = stack[--top];
                for (size_t d = 0; d < nn; ++d)
                    for (int dir = 0; dir < 2; ++dir) {
                        size_t nb = dir == 0 ? lat->fwd[site * nn + d] : lat->bwd[site * nn + d];
                        if (in_cluster[nb]) continue;
                        float c = dir == 0 ? real->couplings[site * nn + d] : real->couplings[nb * nn + d];
                        if (SAT(a, site, nb, c) == SAT(b, site, nb, c)) continue;
                        float r = expf(-2.0f * fabsf(c) / temp);
                        if (orc_rng_gen_f32(rng) < 1.0f - r) {
                            in_cluster[nb] = 1;
                            stack[top++] = nb;
                        }
                    }
            }
            uint8_t k = gen_range_u8_inclusive(rng, 1, 3);
            for (size_t i = 0; i < n; ++i)
                if (in_cluster[i]) {
                    if (k & 1) a[i] = (int8_t)-a[i];
                    if (k & 2) b[i] = (int8_t)-b[i];
                }
        } else {
            /* the single-cluster form of the union-find path draws its seed first (:691-695) */
            size_t seed = wolff ? (size_t)orc_rng_gen_range_usize(rng, 0, n) : 0;
            bond_metrics metrics;
            int with_metrics = out && out->obs;
            if (with_metrics) metrics_begin(&metrics, lat);
            for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
            memset(rank, 0, n);
            for (size_t i = 0; i < n; ++i) /* blue bonds, :582-605 */
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    float c = real->couplings[i * nn + d];
                    if (!SAT(a, i, j, c) || !SAT(b, i, j, c)) continue;
                    float r = expf(-2.0f * fabsf(c) / temp);
                    if (orc_rng_gen_f32(rng) < 1.0f - r * r) {
                        orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                        if (with_metrics) metrics_record(&metrics, lat, i, d);
                    }
                }
            uf_flatten_counts(parent, n, counts);
            emit_counts(out, slot, parent, counts, n, 0); /* sizes are those of the blue clusters, :725-733 */
            if (out && out->blue_snap) memcpy(out->blue_snap + slot * n, parent, sizeof(uint32_t) * n);
            if (with_metrics) metrics_finish(&metrics, counts, n, &out->obs[slot]);
            if (observe) continue; /* :743-745 */
            if (wolff) { /* :748-755 */
                uint32_t seed_root = parent[seed];
                for (size_t i = 0; i < n; ++i)
                    if (parent[i] == seed_root) {
                        a[i] = (int8_t)-a[i];
                        b[i] = (int8_t)-b[i];
                    }
            } else {
                memset(flag, 0xFF, n); /* :757-770 */
                for (size_t i = 0; i < n; ++i) {
                    uint32_t root = parent[i];
                    if (counts[root] > 1 && flag[root] == 0xFF)
                        flag[root] = (uint8_t)(orc_rng_gen_f32(rng) < 0.5f);
                }
                for (size_t i = 0; i < n; ++i)
                    if (flag[parent[i]] == 1) {
                        a[i] = (int8_t)-a[i];
                        b[i] = (int8_t)-b[i];
                    }
            }
            for (size_t i = 0; i < n; ++i) /* red bonds extend the blue forest, :777-795 */
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    float c = real->couplings[i * nn + d];
                    if (SAT(a, i, j, c) == SAT(b, i, j, c)) continue;
                    float r = expf(-2.0f * fabsf(c) / temp);
                    if (orc_rng_gen_f32(rng) < 1.0f - r)
                        orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                }
            uf_flatten_counts(parent, n, counts);
            if (out && out->snap) memcpy(out->snap + slot * n, parent, sizeof(uint32_t) * n); /* :799-803 */
            if (wolff) { /* :806-820 */
                uint32_t seed_root = parent[seed];
                uint8_t k = gen_range_u8_inclusive(rng, 1, 3);
                for (size_t i = 0; i < n; ++i)
                    if (parent[i] == seed_root) {
                        if (k & 1) a[i] = (int8_t)-a[i];
                        if (k & 2) b[i] = (int8_t)-b[i];
                    }
                continue;
            }
            memset(flag, 0xFF, n); /* :822-841 */
            for (size_t i = 0; i < n; ++i) {
                uint32_t root = parent[i];
                if (counts[root] > 1 && flag[root] == 0xFF) flag[root] = gen_range_u8_inclusive(rng, 0, 3);
            }
            for (size_t i = 0; i < n; ++i) {
                uint8_t k = flag[parent[i]];
                if (k == 0 || k == 0xFF) continue;
                if (k & 1) a[i] = (int8_t)-a[i];
                if (k & 2) b[i] = (int8_t)-b[i];
            }
        }
#undef SAT
    }
    free(systems);
    free(in_cluster);
    free(stack);
    free(parent);
    free(counts);
    free(rank);
    free(flag);
}

/* ========================================================================== */
/* Fortuin-Kasteleyn cluster update of every system (ss/clusters/fk.rs:27-200) */
/* ========================================================================== */

/* Update action without CSD collection: Wolff takes the DFS fast path (:42-86), SW the
 * union-find path (:102-196).  Slot k carries temperature temperatures[k] and system
 * system_ids[k]; the system's own generator is used (par_over_replicas / rngs[system_id]). */
static void fk_update(const orc_lattice *lat, realization *real, size_t n_systems, int wolff,
                      const cluster_out *out) {
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    int observe = out && out->observe, has_csd = out && out->csd, with_metrics = out && out->obs;
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *in_cluster = (uint8_t *)malloc(n + 1);
    size_t *stack = (size_t *)malloc(sizeof(size_t) * (n + 1));
    uint32_t *parent = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint8_t *rank = (uint8_t *)malloc(n + 1);
    for (size_t slot = 0; slot < n_systems; ++slot) {
        size_t system = (size_t)real->system_ids[slot];
        int8_t *spin = real->spins + system * n;
        orc_rng *rng = &real->rngs[system];
        float temp = real->temperatures[slot];
        if (!observe && wolff && !has_csd) { /* fk.rs:45 */
            size_t seed = (size_t)orc_rng_gen_range_usize(rng, 0, n);
            memset(in_cluster, 0, n);
            size_t top = 0;
            in_cluster[seed] = 1;
            stack[top++] = seed;
            while (top > 0) { /* dfs_cluster, utils.rs:125-149 */
                size_t site = stack[--top];
                for (size_t d = 0; d < nn; ++d)
                    for (int dir = 0; dir < 2; ++dir) {
                        size_t nb = dir == 0 ? lat->fwd[site * nn + d] : lat->bwd[site * nn + d];
                        if (in_cluster[nb]) continue;
                        float coupling = dir == 0 ? real->couplings[site * nn + d]
                                                  : real->couplings[nb * nn + d];
                        float inter = (float)spin[site] * (float)spin[nb] * coupling;
                        if (inter <= 0.0f) continue;
                        if (orc_rng_gen_f32(rng) < 1.0f - expf(-2.0f * inter / temp)) {
                            in_cluster[nb] = 1;
                            stack[top++] = nb;
                        }
                    }
            }
            for (size_t i = 0; i < n; ++i)
                if (in_cluster[i]) spin[i] = (int8_t)-spin[i];
        } else {
            bond_metrics metrics;
            if (with_metrics) metrics_begin(&metrics, lat);
            for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
            memset(rank, 0, n);
            for (size_t i = 0; i < n; ++i) /* activate_bonds, utils.rs:225-241 */
                for (size_t d = 0; d < nn; ++d) {
                    size_t j = lat->fwd[i * nn + d];
                    float inter = (float)spin[i] * (float)spin[j] * real->couplings[i * nn + d];
                    if (inter <= 0.0f) continue;
                    if (orc_rng_gen_f32(rng) < 1.0f - expf(-2.0f * inter / temp)) {
                        orc_uf_union(parent, rank, (uint32_t)i, (uint32_t)j);
                        if (with_metrics) metrics_record(&metrics, lat, i, d);
                    }
                }
            uf_flatten_counts(parent, n, counts); /* fk.rs:126-138 */
            if (has_csd) orc_uf_histogram(counts, n, out->csd + slot * (n + 1));
            if (with_metrics) metrics_finish(&metrics, counts, n, &out->obs[slot]);
            if (observe) continue; /* fk.rs:140-142 */
            if (wolff) { /* fk.rs:144-151: the seed is drawn after the bonds */
                size_t seed = (size_t)orc_rng_gen_range_usize(rng, 0, n);
                uint32_t seed_root = parent[seed];
                for (size_t i = 0; i < n; ++i)
                    if (parent[i] == seed_root) spin[i] = (int8_t)-spin[i];
                continue;
            }
            memset(rank, 2, n); /* 2 = undecided (fk.rs:182-192): every cluster draws, singletons too */
            for (size_t i = 0; i < n; ++i) {
                uint32_t root = parent[i];
                if (rank[root] == 2) rank[root] = (uint8_t)(orc_rng_gen_f32(rng) < 0.5f);
                if (rank[root] == 1) spin[i] = (int8_t)-spin[i];
            }
        }
    }
    free(in_cluster);
    free(stack);
    free(parent);
    free(rank);
    free(counts);
}

/* ========================================================================== */
/* Autocorrelation, ring backend (ss/statistics/autocorrelation.rs:22-124,165-208) */
/* ========================================================================== */

typedef struct {
    size_t max_lag, n_temps, ring_len, ring_pos, n_recorded;
    double *sum_o, *sum_o2; /* [n_temps] */
    float *ring;            /* [n_temps][ring_len] */
    double *sum_prod;       /* [n_temps][max_lag + 1] */
} autocorr;

static void autocorr_init(autocorr *a, size_t max_lag, size_t n_temps) {
    memset(a, 0, sizeof(*a));
    a->max_lag = max_lag;
    a->n_temps = n_temps;
    a->ring_len = max_lag + 1;
    a->sum_o = (double *)calloc(n_temps + 1, sizeof(double));
    a->sum_o2 = (double *)calloc(n_temps + 1, sizeof(double));
    a->ring = (float *)calloc(n_temps * a->ring_len + 1, sizeof(float));
    a->sum_prod = (double *)calloc(n_temps * (max_lag + 1) + 1, sizeof(double));
}

static void autocorr_free(autocorr *a) {
    free(a->sum_o);
    free(a->sum_o2);
    free(a->ring);
    free(a->sum_prod);
}

/* :70-124 */
static void autocorr_push(autocorr *a, const double *values) {
    for (size_t t = 0; t < a->n_temps; ++t) {
        float o = (float)values[t];
        a->sum_o[t] += (double)o;
        a->sum_o2[t] += (double)o * (double)o;
    }
    size_t pos = a->ring_pos;
    size_t n_back = a->n_recorded < a->max_lag ? a->n_recorded : a->max_lag;
    for (size_t t = 0; t < a->n_temps; ++t) {
        float o = (float)values[t];
        float *ring = a->ring + t * a->ring_len;
        double *sp = a->sum_prod + t * (a->max_lag + 1);
        ring[pos] = o;
        size_t no_wrap = pos < n_back ? pos : n_back;
        for (size_t delta = 0; delta <= no_wrap; ++delta) sp[delta] += (double)o * (double)ring[pos - delta];
        for (size_t delta = pos + 1; delta <= n_back; ++delta)
            sp[delta] += (double)o * (double)ring[pos + a->ring_len - delta];
    }
    a->ring_pos = (pos + 1) % a->ring_len;
    a->n_recorded += 1;
}

/* finish -> normalize_products (:165-189) -> sokal_tau (:199-208) per temperature */
static void autocorr_tau(const autocorr *a, double *tau_out) {
    double *gamma = (double *)malloc(sizeof(double) * (a->max_lag + 1));
    for (size_t t = 0; t < a->n_temps; ++t) {
        for (size_t delta = 0; delta <= a->max_lag; ++delta) gamma[delta] = delta == 0 ? 1.0 : 0.0;
        if (a->n_recorded > 0) {
            double m = (double)a->n_recorded;
            double mean = a->sum_o[t] / m;
            double var = a->sum_o2[t] / m - mean * mean;
            if (var > 0.0)
                for (size_t delta = 0; delta <= a->max_lag; ++delta) {
                    double count = a->n_recorded > delta ? (double)(a->n_recorded - delta) : 0.0;
                    if (count <= 0.0)
                        gamma[delta] = delta == 0 ? 1.0 : 0.0;
                    else
                        gamma[delta] = (a->sum_prod[t * (a->max_lag + 1) + delta] / count - mean * mean) / var;
                }
        }
        double tau = 0.5;
        for (size_t w = 1; w <= a->max_lag; ++w) {
            tau += gamma[w];
            if ((double)w >= 5.0 * tau) break;
        }
        tau_out[t] = tau;
    }
    free(gamma);
}

/* ========================================================================== */
/* Driver loop                                                                */
/* ========================================================================== */

static void result_alloc(sweep_result *res, size_t n_temps, size_t n_spins, size_t n_pairs) {
    memset(res, 0, sizeof(*res));
    res->moments = (double *)calloc(5 * n_temps + 1, sizeof(double));
    if (n_pairs > 0) {
        res->overlap = (double *)calloc(6 * n_temps + 1, sizeof(double));
        res->histogram = (uint64_t *)calloc(n_temps * (n_spins + 1) + 1, sizeof(uint64_t));
        res->ql_at_q_sum = (double *)calloc(n_temps * (n_spins + 1) + 1, sizeof(double));
        res->ql2_at_q_sum = (double *)calloc(n_temps * (n_spins + 1) + 1, sizeof(double));
    }
}

static void result_free(sweep_result *res) {
    free(res->moments);
    free(res->overlap);
    free(res->histogram);
    free(res->ql_at_q_sum);
    free(res->ql2_at_q_sum);
    free(res->mags2_tau);
    free(res->overlap2_tau);
    free(res->equil_sweeps);
    free(res->equil_energy);
    free(res->equil_ql);
    free(res->fk_csd);
    free(res->overlap_csd);
    free(res->top_sizes);
    for (int k = 0; k < 4; ++k) summary_free(&res->observations[k]);
    for (size_t k = 0; k < res->n_snapshots; ++k) {
        free(res->snapshots[k].cluster_ids);
        free(res->snapshots[k].blue_ids);
        free(res->snapshots[k].spins);
        free(res->snapshots[k].system_ids);
    }
    free(res->snapshots);
    memset(res, 0, sizeof(*res));
}

/* ss/config.rs:180-247 (the checks that apply to the options on the path). */
static const char *validate_config(const orc_config *cfg) {
    if (cfg->n_sweeps < 1) return "n_sweeps must be >= 1";
    if (cfg->warmup_sweeps > cfg->n_sweeps) return "warmup_sweeps must be <= n_sweeps";
    if (cfg->cluster_interval > 0 && cfg->cluster_action == ORC_ACTION_OBSERVE &&
        cfg->cluster_mode == ORC_CLUSTER_WOLFF)
        return "cluster_action='observe' requires cluster_mode='sw'";
    if (cfg->overlap_cluster_interval > 0) {
        if (cfg->snapshot_interval > 0 && cfg->snapshot_interval % cfg->overlap_cluster_interval != 0)
            return "snapshot_interval must be a positive multiple of overlap_cluster interval";
        if (cfg->n_overlap_cluster_modes < 1) return "overlap_cluster modes must not be empty";
        if (cfg->overlap_cluster_action == ORC_ACTION_OBSERVE) {
            if (cfg->overlap_cluster_mode == ORC_CLUSTER_WOLFF)
                return "overlap_cluster_action='observe' requires overlap_cluster_mode='sw'";
            if (cfg->snapshot_interval > 0)
                return "snapshot_interval is not supported with overlap_cluster_action='observe'";
            for (int k = 0; k < cfg->n_overlap_cluster_modes; ++k)
                if ((cfg->overlap_cluster_modes[k] & 0xFF) == ORC_OC_HOUDAYER &&
                    (cfg->overlap_cluster_modes[k] >> 8) > 2)
                    return "overlap_cluster_action='observe' does not support experimental houdN with N > 2";
        }
    }
    return NULL;
}

/* ss/simulation/mod.rs:177-858, restricted to the options on the path:
 * sweep -> [energies(+mags)] -> [overlap collect] -> [stats] -> [overlap
 * cluster move] -> [parallel tempering]. */
static void run_sweep_loop(const orc_lattice *lat, realization *real, size_t n_replicas,
                           size_t n_temps, const orc_config *cfg, sweep_result *res,
                           size_t realization_idx) {
    size_t n = lat->n_spins, nn = lat->n_neighbors;
    size_t n_systems = n_replicas * n_temps, n_pairs = n_replicas / 2;
    size_t n_bonds = n * nn;

    /* mod.rs:205-335: what the cluster calls report besides moving spins */
    size_t n_modes = cfg->overlap_cluster_interval > 0 ? (size_t)cfg->n_overlap_cluster_modes : 0;
    size_t n_pair_slots = n_temps * n_pairs;
    int has_fk = cfg->cluster_interval > 0, has_oc = cfg->overlap_cluster_interval > 0;
    int observe_fk = has_fk && cfg->cluster_action == ORC_ACTION_OBSERVE;
    int observe_overlap = has_oc && cfg->overlap_cluster_action == ORC_ACTION_OBSERVE;
    int collect_fk = has_fk && (cfg->collect_cluster_stats || observe_fk);
    int collect_overlap = has_oc && (cfg->collect_cluster_stats || observe_overlap) && n_pairs > 0;
    static const int kind_of_mode[4] = {-1, 1, 2, 3}; /* ORC_OC_* -> houdayer, jorg, cmr_blue */
    if (observe_fk) summary_enable(&res->observations[0], n_temps, n);
    if (observe_overlap)
        for (size_t m = 0; m < n_modes; ++m)
            summary_enable(&res->observations[kind_of_mode[cfg->overlap_cluster_modes[m] & 0xFF]], n_temps, n);
    uint64_t *sw_csd_buf = NULL, *overlap_csd_buf = NULL;
    graph_slot *fk_obs_buf = NULL, *overlap_obs_buf = NULL;
    uint32_t *top4_buf = NULL;
    double *top4_accum = NULL;
    size_t *top4_n = NULL;
    if (collect_fk) {
        res->fk_csd = (uint64_t *)calloc(n_temps * (n + 1) + 1, sizeof(uint64_t));
        sw_csd_buf = (uint64_t *)calloc(n_systems * (n + 1) + 1, sizeof(uint64_t));
    }
    if (observe_fk) fk_obs_buf = (graph_slot *)calloc(n_systems + 1, sizeof(graph_slot));
    if (collect_overlap) {
        res->n_modes = n_modes;
        res->overlap_csd = (uint64_t *)calloc(n_modes * n_temps * (n + 1) + 1, sizeof(uint64_t));
        res->top_sizes = (double *)calloc(n_modes * n_temps * 4 + 1, sizeof(double));
        overlap_csd_buf = (uint64_t *)calloc(n_pair_slots * (n + 1) + 1, sizeof(uint64_t));
        top4_buf = (uint32_t *)calloc(n_pair_slots * 4 + 1, sizeof(uint32_t));
        top4_accum = (double *)calloc(n_modes * n_temps * 4 + 1, sizeof(double));
        top4_n = (size_t *)calloc(n_modes + 1, sizeof(size_t));
    }
    if (observe_overlap) overlap_obs_buf = (graph_slot *)calloc(n_pair_slots + 1, sizeof(graph_slot));
    /* snapshots: first realisation only (mod.rs:302-335) */
    uint64_t snapshot_interval = (realization_idx == 0 && has_oc) ? cfg->snapshot_interval : 0;
    uint32_t *snap_buf = NULL, *blue_snap_buf = NULL;
    int8_t *spin_snap_buf = NULL;
    uint64_t *sid_snap_buf = NULL;
    if (snapshot_interval > 0) {
        snap_buf = (uint32_t *)calloc(n_pair_slots * n + 1, sizeof(uint32_t));
        blue_snap_buf = (uint32_t *)calloc(n_pair_slots * n + 1, sizeof(uint32_t));
        spin_snap_buf = (int8_t *)calloc(n_pair_slots * 2 * n + 1, 1);
        sid_snap_buf = (uint64_t *)calloc(n_pair_slots * 2 + 1, sizeof(uint64_t));
    }

    /* mod.rs:190-198 */
    orc_lookup *lookup = NULL;
    if (cfg->sweep_mode == ORC_SWEEP_METROPOLIS)
        lookup = orc_lookup_new(real->couplings, n * nn, real->temperatures, n_systems, nn);

    statistics mags, mags2, mags4, energies, energies2; /* mod.rs:337-341 */
    stat_init(&mags, n_temps, 1);
    stat_init(&mags2, n_temps, 1);
    stat_init(&mags4, n_temps, 1);
    stat_init(&energies, n_temps, 1);
    stat_init(&energies2, n_temps, 2);
    statistics ov[6]; /* overlap.rs:216-221 */
    for (int k = 0; k < 6; ++k) stat_init(&ov[k], n_temps, 1);

    float *mags_buf = (float *)calloc(n_temps + 1, sizeof(float));
    float *mags2_buf = (float *)calloc(n_temps + 1, sizeof(float));
    float *mags4_buf = (float *)calloc(n_temps + 1, sizeof(float));
    float *energies_buf = (float *)calloc(n_temps + 1, sizeof(float));
    float *ovbuf = (float *)calloc(6 * n_temps + 1, sizeof(float));
    int64_t *magnetization_sums = (int64_t *)calloc(n_systems + 1, sizeof(int64_t));
    orc_tempering_attempt *attempts =
        (orc_tempering_attempt *)malloc(sizeof(orc_tempering_attempt) * (n_temps + 1));
    size_t overlap_call_count = 0;

    /* mod.rs:342-371: max_lag capped at a quarter of the measurement sweeps, at least 1 */
    int collect_ac = cfg->autocorrelation_max_lag > 0;
    autocorr m2_acc, q2_acc;
    double *m2_ac_buf = (double *)calloc(n_temps + 1, sizeof(double));
    double *q2_ac_buf = (double *)calloc(n_temps + 1, sizeof(double));
    if (collect_ac) {
        uint64_t n_meas = cfg->n_sweeps - cfg->warmup_sweeps;
        uint64_t lag = cfg->autocorrelation_max_lag < n_meas / 4 ? cfg->autocorrelation_max_lag : n_meas / 4;
        if (lag < 1) lag = 1;
        autocorr_init(&m2_acc, (size_t)lag, n_temps);
        if (n_pairs > 0) autocorr_init(&q2_acc, (size_t)lag, n_temps);
    }

    /* EquilDiagnosticAccum::new (equilibration.rs:18-38): checkpoints 128, 256, ... < n_sweeps, n_sweeps */
    int equil_diag = cfg->equilibration_diagnostic != 0;
    size_t equil_next = 0, equil_count = 0;
    double *equil_sum_e = (double *)calloc(n_temps + 1, sizeof(double));
    double *equil_sum_ql = (double *)calloc(n_temps + 1, sizeof(double));
    float *diag_e_buf = (float *)calloc(n_temps + 1, sizeof(float));
    float *diag_ql_buf = (float *)calloc(n_temps + 1, sizeof(float));
    if (equil_diag) {
        size_t cap = 70;
        res->equil_sweeps = (uint64_t *)calloc(cap, sizeof(uint64_t));
        for (uint64_t pw = 128; pw < cfg->n_sweeps; pw *= 2) res->equil_sweeps[res->n_equil++] = pw;
        if (res->n_equil == 0 || res->equil_sweeps[res->n_equil - 1] != cfg->n_sweeps)
            res->equil_sweeps[res->n_equil++] = cfg->n_sweeps;
        res->equil_energy = (double *)calloc(res->n_equil * n_temps + 1, sizeof(double));
        res->equil_ql = (double *)calloc(res->n_equil * n_temps + 1, sizeof(double));
    }

    for (uint64_t sweep_id = 0; sweep_id < cfg->n_sweeps; ++sweep_id) {
        int record = sweep_id >= cfg->warmup_sweeps; /* mod.rs:410 */

        if (cfg->sweep_mode == ORC_SWEEP_METROPOLIS) /* mod.rs:412-432 */
            orc_metropolis_sweep(lat, real->spins, real->couplings, real->temperatures,
                                 real->system_ids, n_systems, real->rngs, lookup);
        else
            orc_gibbs_sweep(lat, real->spins, real->couplings, real->temperatures, real->system_ids,
                            n_systems, real->rngs);

        if (cfg->cluster_interval > 0 && sweep_id % cfg->cluster_interval == 0) { /* mod.rs:434-484 */
            cluster_out out;
            memset(&out, 0, sizeof(out));
            out.observe = observe_fk;
            if (collect_fk && record) {
                memset(sw_csd_buf, 0, sizeof(uint64_t) * n_systems * (n + 1));
                out.csd = sw_csd_buf;
            }
            if (observe_fk && record) {
                memset(fk_obs_buf, 0, sizeof(graph_slot) * n_systems);
                out.obs = fk_obs_buf;
            }
            fk_update(lat, real, n_systems, cfg->cluster_mode == ORC_CLUSTER_WOLFF, &out);
            if (collect_fk && record)
                for (size_t slot = 0; slot < n_systems; ++slot)
                    for (size_t k = 0; k <= n; ++k)
                        res->fk_csd[(slot % n_temps) * (n + 1) + k] += sw_csd_buf[slot * (n + 1) + k];
            if (observe_fk && record)
                for (size_t slot = 0; slot < n_systems; ++slot)
                    summary_push(&res->observations[0], slot % n_temps, sw_csd_buf + slot * (n + 1), n,
                                 &fk_obs_buf[slot]);
        }

        int pt_this_sweep = cfg->pt_interval > 0 && sweep_id % cfg->pt_interval == 0; /* :486-488 */

        if (record || pt_this_sweep || equil_diag) /* mod.rs:492-509 */
            orc_compute_energies(lat, real->spins, real->couplings, n_systems, real->energies,
                                 record ? magnetization_sums : NULL, NULL);

        if (equil_diag) { /* mod.rs:511-525 */
            for (size_t t = 0; t < n_temps; ++t) diag_e_buf[t] = 0.0f;
            for (size_t r = 0; r < n_replicas; ++r)
                for (size_t t = 0; t < n_temps; ++t)
                    diag_e_buf[t] += real->energies[(size_t)real->system_ids[r * n_temps + t]];
            float inv = 1.0f / (float)n_replicas;
            for (size_t t = 0; t < n_temps; ++t) diag_e_buf[t] *= inv;
            for (size_t t = 0; t < n_temps; ++t) diag_ql_buf[t] = 0.0f;
        }

        if (collect_ac && record)
            for (size_t t = 0; t < n_temps; ++t) q2_ac_buf[t] = 0.0; /* overlap.rs:255-257 */
        if ((equil_diag || record) && n_pairs > 0) { /* mod.rs:527-529 -> overlap.rs:251-333 */
            for (size_t pair_idx = 0; pair_idx < n_pairs; ++pair_idx) {
                size_t r_a = 2 * pair_idx, r_b = 2 * pair_idx + 1;
                for (size_t t = 0; t < n_temps; ++t) {
                    size_t base_a = (size_t)real->system_ids[r_a * n_temps + t] * n;
                    size_t base_b = (size_t)real->system_ids[r_b * n_temps + t] * n;
                    int64_t dot_spin = 0, dot_link = 0;
                    for (size_t j = 0; j < n; ++j) {
                        int64_t q = (int64_t)real->spins[base_a + j] * (int64_t)real->spins[base_b + j];
                        dot_spin += q;
                        for (size_t d = 0; d < nn; ++d) {
                            size_t k = lat->fwd[j * nn + d];
                            int64_t neighbor_q =
                                (int64_t)real->spins[base_a + k] * (int64_t)real->spins[base_b + k];
                            dot_link += q * neighbor_q;
                        }
                    }
                    float ql = (float)dot_link / (float)n_bonds;
                    if (equil_diag) diag_ql_buf[t] += ql; /* overlap.rs:285-287 */
                    if (!record) continue;
                    float q = (float)dot_spin / (float)n;
                    float q2 = q * q;
                    ovbuf[0 * n_temps + t] = q;
                    ovbuf[1 * n_temps + t] = q2;
                    ovbuf[2 * n_temps + t] = q2 * q2;
                    float ql2 = ql * ql;
                    ovbuf[3 * n_temps + t] = ql;
                    ovbuf[4 * n_temps + t] = ql2;
                    ovbuf[5 * n_temps + t] = ql2 * ql2;
                    size_t idx = (size_t)((dot_spin + (int64_t)n) / 2);
                    res->histogram[t * (n + 1) + idx] += 1;
                    res->ql_at_q_sum[t * (n + 1) + idx] += (double)ql;
                    res->ql2_at_q_sum[t * (n + 1) + idx] += (double)(ql * ql);
                }
                if (record) {
                    if (collect_ac) /* overlap.rs:310-314 */
                        for (size_t t = 0; t < n_temps; ++t) q2_ac_buf[t] += (double)ovbuf[1 * n_temps + t];
                    for (int k = 0; k < 6; ++k) stat_update(&ov[k], ovbuf + k * n_temps, n_temps);
                }
            }
            if (equil_diag) { /* overlap.rs:327-332 */
                float inv = 1.0f / (float)n_pairs;
                for (size_t t = 0; t < n_temps; ++t) diag_ql_buf[t] *= inv;
            }
        }

        if (equil_diag) { /* mod.rs:531-541, equilibration.rs:40-58 */
            equil_count += 1;
            for (size_t t = 0; t < n_temps; ++t) {
                equil_sum_e[t] += (double)diag_e_buf[t];
                equil_sum_ql[t] += (double)diag_ql_buf[t];
            }
            if (equil_next < res->n_equil && equil_count == res->equil_sweeps[equil_next]) {
                double c = (double)equil_count;
                for (size_t t = 0; t < n_temps; ++t) {
                    res->equil_energy[equil_next * n_temps + t] = equil_sum_e[t] / c;
                    res->equil_ql[equil_next * n_temps + t] = equil_sum_ql[t] / c;
                }
                equil_next += 1;
            }
        }

        if (record) { /* mod.rs:543-578 */
            for (size_t t = 0; t < n_temps; ++t) m2_ac_buf[t] = 0.0;
            for (size_t r = 0; r < n_replicas; ++r) {
                size_t offset = r * n_temps;
                for (size_t t = 0; t < n_temps; ++t) {
                    size_t system_id = (size_t)real->system_ids[offset + t];
                    float mag = (float)magnetization_sums[system_id] / (float)n;
                    float m2 = mag * mag;
                    mags_buf[t] = mag;
                    mags2_buf[t] = m2;
                    mags4_buf[t] = m2 * m2;
                    energies_buf[t] = real->energies[system_id];
                }
                if (collect_ac)
                    for (size_t t = 0; t < n_temps; ++t) m2_ac_buf[t] += (double)mags2_buf[t];
                stat_update(&mags, mags_buf, n_temps);
                stat_update(&mags2, mags2_buf, n_temps);
                stat_update(&mags4, mags4_buf, n_temps);
                stat_update(&energies, energies_buf, n_temps);
                stat_update(&energies2, energies_buf, n_temps);
            }
            if (collect_ac) { /* mod.rs:580-595 */
                double inv = 1.0 / (double)n_replicas;
                for (size_t t = 0; t < n_temps; ++t) m2_ac_buf[t] *= inv;
                autocorr_push(&m2_acc, m2_ac_buf);
                if (n_pairs > 0) {
                    double inv_p = 1.0 / (double)n_pairs;
                    for (size_t t = 0; t < n_temps; ++t) q2_ac_buf[t] *= inv_p;
                    autocorr_push(&q2_acc, q2_ac_buf);
                }
            }
        }

        int did_overlap_mutate = 0; /* mod.rs:597-746 */
        if (cfg->overlap_cluster_interval > 0 && sweep_id % cfg->overlap_cluster_interval == 0) {
            did_overlap_mutate = !observe_overlap;
            size_t mode_idx = overlap_call_count % (size_t)cfg->n_overlap_cluster_modes;
            int mode = cfg->overlap_cluster_modes[mode_idx];
            cluster_out out;
            memset(&out, 0, sizeof(out));
            out.observe = observe_overlap;
            if (collect_overlap && record) { /* mod.rs:604-626 */
                memset(overlap_csd_buf, 0, sizeof(uint64_t) * n_pair_slots * (n + 1));
                memset(top4_buf, 0, sizeof(uint32_t) * n_pair_slots * 4);
                out.csd = overlap_csd_buf;
                out.top4 = top4_buf;
            }
            if (observe_overlap && record) {
                memset(overlap_obs_buf, 0, sizeof(graph_slot) * n_pair_slots);
                out.obs = overlap_obs_buf;
            }
            int take_snapshot = snapshot_interval > 0 && sweep_id % snapshot_interval == 0 && record;
            int is_cmr = (mode & 0xFF) == ORC_OC_CMR;
            if (take_snapshot) { /* mod.rs:628-658 */
                memset(spin_snap_buf, 0, n_pair_slots * 2 * n);
                memset(sid_snap_buf, 0, sizeof(uint64_t) * n_pair_slots * 2);
                out.snap = snap_buf;
                out.spin_snap = spin_snap_buf;
                out.sid_snap = sid_snap_buf;
                if (is_cmr) out.blue_snap = blue_snap_buf;
            }
            /* mode = kind | group_size << 8 (group size 0 or 2: pairs) */
            if (is_cmr)
                cmr_update(lat, real, n_replicas, n_temps, cfg->overlap_cluster_mode, &out);
            else if ((mode & 0xFF) == ORC_OC_HOUDAYER && (mode >> 8) > 2)
                houdayer_n_update(lat, real, n_replicas, n_temps, (size_t)(mode >> 8), cfg->overlap_cluster_mode,
                                  &out);
            else
                overlap_update(lat, real, n_replicas, n_temps, mode & 0xFF, cfg->overlap_cluster_mode, &out);
            if (take_snapshot) { /* mod.rs:682-714: pair 0 of every temperature */
                res->snapshots = (cluster_snapshot *)realloc(res->snapshots,
                                                             sizeof(cluster_snapshot) * (res->n_snapshots + 1));
                cluster_snapshot *snap = &res->snapshots[res->n_snapshots++];
                memset(snap, 0, sizeof(*snap));
                snap->sweep_id = sweep_id;
                snap->mode_idx = (int)mode_idx;
                snap->has_blue = is_cmr;
                snap->cluster_ids = (uint32_t *)calloc(n_temps * n + 1, sizeof(uint32_t));
                snap->blue_ids = (uint32_t *)calloc(n_temps * n + 1, sizeof(uint32_t));
                snap->spins = (int8_t *)calloc(n_temps * 2 * n + 1, 1);
                snap->system_ids = (uint64_t *)calloc(n_temps * 2 + 1, sizeof(uint64_t));
                for (size_t t = 0; t < n_temps; ++t) {
                    size_t slot = t * n_pairs;
                    memcpy(snap->cluster_ids + t * n, snap_buf + slot * n, sizeof(uint32_t) * n);
                    if (is_cmr) memcpy(snap->blue_ids + t * n, blue_snap_buf + slot * n, sizeof(uint32_t) * n);
                    memcpy(snap->spins + t * 2 * n, spin_snap_buf + slot * 2 * n, 2 * n);
                    snap->system_ids[t * 2] = sid_snap_buf[slot * 2];
                    snap->system_ids[t * 2 + 1] = sid_snap_buf[slot * 2 + 1];
                }
            }
            if (collect_overlap && record) { /* mod.rs:716-735 */
                for (size_t slot = 0; slot < n_pair_slots; ++slot)
                    for (size_t k = 0; k <= n; ++k)
                        res->overlap_csd[(mode_idx * n_temps + slot / n_pairs) * (n + 1) + k] +=
                            overlap_csd_buf[slot * (n + 1) + k];
                for (size_t t = 0; t < n_temps; ++t)
                    for (size_t pr = 0; pr < n_pairs; ++pr)
                        for (int k = 0; k < 4; ++k)
                            top4_accum[(mode_idx * n_temps + t) * 4 + k] +=
                                (double)top4_buf[(t * n_pairs + pr) * 4 + k] / (double)n;
                top4_n[mode_idx] += 1;
            }
            if (record && observe_overlap) /* mod.rs:737-742 */
                for (size_t slot = 0; slot < n_pair_slots; ++slot)
                    summary_push(&res->observations[kind_of_mode[mode & 0xFF]], slot / n_pairs,
                                 overlap_csd_buf + slot * (n + 1), n, &overlap_obs_buf[slot]);
            overlap_call_count += 1;
        }

        if (pt_this_sweep) { /* mod.rs:748-796 */
            if (did_overlap_mutate)
                orc_compute_energies(lat, real->spins, real->couplings, n_systems, real->energies,
                                     NULL, NULL);
            size_t first_parity = real->pt.next_parity;
            for (size_t r = 0; r < n_replicas; ++r) {
                size_t offset = r * n_temps;
                size_t count;
                if (cfg->pt_schedule == ORC_PT_SINGLE_RANDOM_EDGE)
                    count = orc_parallel_tempering(real->energies, real->temperatures + offset,
                                                   real->system_ids + offset, n_temps, n,
                                                   &real->rngs[offset], attempts);
                else
                    count = orc_parallel_tempering_full_ladder(
                        real->energies, real->temperatures + offset, real->system_ids + offset,
                        n_temps, n, &real->rngs[offset], first_parity, attempts);
                for (size_t a = 0; a < count; ++a) pt_record_attempt(&real->pt, attempts[a]);
            }
            if (cfg->pt_schedule == ORC_PT_FULL_LADDER) real->pt.next_parity = 1 - real->pt.next_parity;
        }
    }

    /* mod.rs:838-857, overlap.rs:335-354 */
    stat_average(&mags, res->moments + 0 * n_temps, n_temps);
    stat_average(&mags2, res->moments + 1 * n_temps, n_temps);
    stat_average(&mags4, res->moments + 2 * n_temps, n_temps);
    stat_average(&energies, res->moments + 3 * n_temps, n_temps);
    stat_average(&energies2, res->moments + 4 * n_temps, n_temps);
    if (n_pairs > 0)
        for (int k = 0; k < 6; ++k) stat_average(&ov[k], res->overlap + k * n_temps, n_temps);

    if (collect_overlap) /* mod.rs:799-823 */
        for (size_t m = 0; m < n_modes; ++m) {
            if (top4_n[m] == 0) continue;
            double denom = (double)(top4_n[m] * n_pairs);
            for (size_t k = 0; k < n_temps * 4; ++k)
                res->top_sizes[m * n_temps * 4 + k] = top4_accum[m * n_temps * 4 + k] / denom;
        }
    for (int k = 0; k < 4; ++k) summary_finish(&res->observations[k], n_temps, n, nn, lat->has_square_shape);
    free(sw_csd_buf);
    free(overlap_csd_buf);
    free(fk_obs_buf);
    free(overlap_obs_buf);
    free(top4_buf);
    free(top4_accum);
    free(top4_n);
    free(snap_buf);
    free(blue_snap_buf);
    free(spin_snap_buf);
    free(sid_snap_buf);

    free(mags.aggregate);
    free(mags2.aggregate);
    free(mags4.aggregate);
    free(energies.aggregate);
    free(energies2.aggregate);
    for (int k = 0; k < 6; ++k) free(ov[k].aggregate);
    free(mags_buf);
    free(mags2_buf);
    free(mags4_buf);
    free(energies_buf);
    free(ovbuf);
    free(magnetization_sums);
    free(attempts);
    if (collect_ac) { /* mod.rs:825-832 */
        res->mags2_tau = (double *)calloc(n_temps + 1, sizeof(double));
        autocorr_tau(&m2_acc, res->mags2_tau);
        autocorr_free(&m2_acc);
        if (n_pairs > 0) {
            res->overlap2_tau = (double *)calloc(n_temps + 1, sizeof(double));
            autocorr_tau(&q2_acc, res->overlap2_tau);
            autocorr_free(&q2_acc);
        }
    }
    free(m2_ac_buf);
    free(q2_ac_buf);
    free(equil_sum_e);
    free(equil_sum_ql);
    free(diag_e_buf);
    free(diag_ql_buf);
    orc_lookup_free(lookup);
}

/* ========================================================================== */
/* Simulation object (src/lib.rs IsingSimulation)                             */
/* ========================================================================== */

/* src/lib.rs:108-174 */
orc_sim *orc_sim_create(const uint64_t *shape, int n_dims, const int64_t *offsets, size_t n_offsets,
                        const float *couplings, size_t n_realizations, const float *temperatures,
                        size_t n_temps, size_t n_replicas, uint64_t seed) {
    return orc_sim_create_shard(shape, n_dims, offsets, n_offsets, couplings, n_realizations,
                                temperatures, n_temps, n_replicas, seed, 0);
}

/* Same, for a shard of a larger disorder ensemble: realisation r of this object is global
 * realisation first_realization + r and is seeded as such (src/lib.rs:30-32,160). */
orc_sim *orc_sim_create_shard(const uint64_t *shape, int n_dims, const int64_t *offsets,
                              size_t n_offsets, const float *couplings, size_t n_realizations,
                              const float *temperatures, size_t n_temps, size_t n_replicas,
                              uint64_t seed, uint64_t first_realization) {
    orc_sim *sim = (orc_sim *)calloc(1, sizeof(*sim));
    sim->first_realization = first_realization;
    sim->lattice = (offsets && n_offsets > 0)
                       ? orc_lattice_with_offsets(shape, n_dims, offsets, n_offsets)
                       : orc_lattice_new(shape, n_dims);
    if (!sim->lattice) {
        free(sim);
        return NULL;
    }
    sim->n_replicas = n_replicas;
    sim->n_temps = n_temps;
    sim->n_realizations = n_realizations;
    sim->constructor_seed = seed;
    sim->realizations = (realization *)calloc(n_realizations + 1, sizeof(realization));
    sim->results = (sweep_result *)calloc(n_realizations + 1, sizeof(sweep_result));
    size_t chunk = sim->lattice->n_spins * sim->lattice->n_neighbors;
    for (size_t r = 0; r < n_realizations; ++r) {
        realization_alloc(sim->lattice, &sim->realizations[r], couplings + r * chunk, temperatures,
                          n_replicas, n_temps);
        realization_reset(sim->lattice, &sim->realizations[r], n_replicas, n_temps,
                          orc_realization_seed(seed, first_realization + r));
    }
    return sim;
}

void orc_sim_destroy(orc_sim *sim) {
    if (!sim) return;
    for (size_t r = 0; r < sim->n_realizations; ++r) {
        realization_free(&sim->realizations[r]);
        result_free(&sim->results[r]);
    }
    free(sim->realizations);
    free(sim->results);
    orc_lattice_free(sim->lattice);
    free(sim);
}

/* src/lib.rs:624-633 */
void orc_sim_reset(orc_sim *sim, uint64_t seed) {
    for (size_t r = 0; r < sim->n_realizations; ++r)
        realization_reset(sim->lattice, &sim->realizations[r], sim->n_replicas, sim->n_temps,
                          orc_realization_seed(seed, sim->first_realization + r));
}

typedef struct {
    orc_sim *sim;
    const orc_config *cfg;
    size_t begin, end;
} worker_args;

static void *worker_main(void *p) {
    worker_args *args = (worker_args *)p;
    for (size_t r = args->begin; r < args->end; ++r)
        run_sweep_loop(args->sim->lattice, &args->sim->realizations[r], args->sim->n_replicas,
                       args->sim->n_temps, args->cfg, &args->sim->results[r], r);
    return NULL;
}

/* ss/simulation/mod.rs:865-939: realisations are independent (outer rayon
 * level); here a static split over pthreads.  Aggregation over realisations
 * (results.rs:84-285) is left to the caller, which receives per-realisation
 * results in realisation order. */
int orc_sim_sample(orc_sim *sim, const orc_config *cfg) {
    const char *problem = validate_config(cfg);
    if (!problem && cfg->overlap_cluster_interval > 0 && sim->n_replicas < 2)
        problem = "overlap cluster requires n_replicas >= max group_size";
    if (problem) {
        snprintf(sim->error, sizeof(sim->error), "%s", problem);
        return 1;
    }
    size_t n_pairs = sim->n_replicas / 2;
    for (size_t r = 0; r < sim->n_realizations; ++r) {
        result_free(&sim->results[r]);
        result_alloc(&sim->results[r], sim->n_temps, sim->lattice->n_spins, n_pairs);
    }
    size_t n_threads = cfg->n_threads > 1 ? (size_t)cfg->n_threads : 1;
    if (n_threads > sim->n_realizations) n_threads = sim->n_realizations;
    if (n_threads <= 1) {
        worker_args args = {sim, cfg, 0, sim->n_realizations};
        worker_main(&args);
        return 0;
    }
    pthread_t *threads = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    worker_args *args = (worker_args *)malloc(sizeof(worker_args) * n_threads);
    for (size_t k = 0; k < n_threads; ++k) {
        args[k].sim = sim;
        args[k].cfg = cfg;
        args[k].begin = sim->n_realizations * k / n_threads;
        args[k].end = sim->n_realizations * (k + 1) / n_threads;
        pthread_create(&threads[k], NULL, worker_main, &args[k]);
    }
    for (size_t k = 0; k < n_threads; ++k) pthread_join(threads[k], NULL);
    free(threads);
    free(args);
    return 0;
}

const char *orc_sim_last_error(const orc_sim *sim) { return sim->error; }
uint64_t orc_sim_n_spins(const orc_sim *sim) { return sim->lattice->n_spins; }
uint64_t orc_sim_n_neighbors(const orc_sim *sim) { return sim->lattice->n_neighbors; }

void orc_sim_get_spins(const orc_sim *sim, size_t r, int8_t *out) {
    memcpy(out, sim->realizations[r].spins, sim->n_replicas * sim->n_temps * sim->lattice->n_spins);
}
void orc_sim_get_system_ids(const orc_sim *sim, size_t r, uint64_t *out) {
    memcpy(out, sim->realizations[r].system_ids, sizeof(uint64_t) * sim->n_replicas * sim->n_temps);
}
void orc_sim_get_energies(const orc_sim *sim, size_t r, float *out) {
    memcpy(out, sim->realizations[r].energies, sizeof(float) * sim->n_replicas * sim->n_temps);
}
void orc_sim_get_rng_state(const orc_sim *sim, size_t r, size_t system, uint64_t out[4]) {
    memcpy(out, sim->realizations[r].rngs[system].s, sizeof(uint64_t) * 4);
}
void orc_sim_get_pt_counters(const orc_sim *sim, size_t r, uint64_t *edge_attempts,
                             uint64_t *edge_acceptances, uint64_t *round_trips) {
    const pt_state *pt = &sim->realizations[r].pt;
    size_t n_edges = sim->n_temps > 0 ? sim->n_temps - 1 : 0;
    memcpy(edge_attempts, pt->edge_attempts, sizeof(uint64_t) * n_edges);
    memcpy(edge_acceptances, pt->edge_acceptances, sizeof(uint64_t) * n_edges);
    memcpy(round_trips, pt->round_trips, sizeof(uint64_t) * sim->n_replicas * sim->n_temps);
}
void orc_sim_get_result_moments(const orc_sim *sim, size_t r, double *out) {
    memcpy(out, sim->results[r].moments, sizeof(double) * 5 * sim->n_temps);
}
int orc_sim_has_overlap(const orc_sim *sim) { return sim->n_replicas / 2 > 0; }

int orc_sim_get_tau(const orc_sim *sim, size_t r, double *mags2_tau, double *overlap2_tau) {
    const sweep_result *res = &sim->results[r];
    if (!res->mags2_tau) return 0;
    memcpy(mags2_tau, res->mags2_tau, sizeof(double) * sim->n_temps);
    if (res->overlap2_tau) memcpy(overlap2_tau, res->overlap2_tau, sizeof(double) * sim->n_temps);
    return res->overlap2_tau ? 2 : 1;
}

size_t orc_sim_n_equil_checkpoints(const orc_sim *sim) { return sim->results[0].n_equil; }
void orc_sim_get_equil(const orc_sim *sim, size_t r, uint64_t *sweeps, double *energy_avg,
                       double *link_overlap_avg) {
    const sweep_result *res = &sim->results[r];
    memcpy(sweeps, res->equil_sweeps, sizeof(uint64_t) * res->n_equil);
    memcpy(energy_avg, res->equil_energy, sizeof(double) * res->n_equil * sim->n_temps);
    memcpy(link_overlap_avg, res->equil_ql, sizeof(double) * res->n_equil * sim->n_temps);
}
void orc_sim_get_result_overlap(const orc_sim *sim, size_t r, double *out) {
    memcpy(out, sim->results[r].overlap, sizeof(double) * 6 * sim->n_temps);
}
void orc_sim_get_result_histograms(const orc_sim *sim, size_t r, uint64_t *histogram,
                                   double *ql_at_q_sum, double *ql2_at_q_sum) {
    size_t count = sim->n_temps * (sim->lattice->n_spins + 1);
    memcpy(histogram, sim->results[r].histogram, sizeof(uint64_t) * count);
    memcpy(ql_at_q_sum, sim->results[r].ql_at_q_sum, sizeof(double) * count);
    memcpy(ql2_at_q_sum, sim->results[r].ql2_at_q_sum, sizeof(double) * count);
}

/* cluster size histograms and mean top-4 sizes of one realisation (ClusterStats, results.rs:12-20);
 * a pointer may be NULL; returns bit 0 = FK histogram present, bit 1 = overlap statistics present */
int orc_sim_get_cluster_stats(const orc_sim *sim, size_t r, uint64_t *fk_csd, uint64_t *overlap_csd,
                              double *top_sizes) {
    const sweep_result *res = &sim->results[r];
    size_t n = sim->lattice->n_spins, n_temps = sim->n_temps;
    if (res->fk_csd && fk_csd) memcpy(fk_csd, res->fk_csd, sizeof(uint64_t) * n_temps * (n + 1));
    if (res->overlap_csd && overlap_csd)
        memcpy(overlap_csd, res->overlap_csd, sizeof(uint64_t) * res->n_modes * n_temps * (n + 1));
    if (res->top_sizes && top_sizes) memcpy(top_sizes, res->top_sizes, sizeof(double) * res->n_modes * n_temps * 4);
    return (res->fk_csd ? 1 : 0) | (res->overlap_csd ? 2 : 0);
}

/* graph observation summary `kind` (0 fk, 1 houdayer, 2 jorg, 3 cmr_blue) of one realisation;
 * returns 0 if absent, 1 if present, 3 if present with winding fractions */
int orc_sim_get_observation(const orc_sim *sim, size_t r, int kind, uint64_t *count, uint64_t *sizes,
                            double *top4_frac, double *bond_density, double *large_mean, double *winding) {
    const graph_summary *g = &sim->results[r].observations[kind];
    size_t n = sim->lattice->n_spins, n_temps = sim->n_temps;
    if (!g->enabled || !g->present) return 0;
    memcpy(count, g->count, sizeof(uint64_t) * n_temps);
    memcpy(sizes, g->csd, sizeof(uint64_t) * n_temps * (n + 1));
    memcpy(top4_frac, g->top4_frac, sizeof(double) * n_temps * 4);
    memcpy(bond_density, g->bond_density, sizeof(double) * n_temps);
    memcpy(large_mean, g->large_mean, sizeof(double) * n_temps);
    if (g->has_winding) memcpy(winding, g->winding_frac, sizeof(double) * n_temps * 4);
    return g->has_winding ? 3 : 1;
}

size_t orc_sim_n_snapshots(const orc_sim *sim) { return sim->results[0].n_snapshots; }

/* snapshot k of realisation 0: returns 1 when blue_ids are present (CMR) */
int orc_sim_get_snapshot(const orc_sim *sim, size_t k, uint64_t *sweep_id, int32_t *mode_idx,
                         uint32_t *cluster_ids, uint32_t *blue_ids, int8_t *spins, uint64_t *system_ids) {
    const cluster_snapshot *snap = &sim->results[0].snapshots[k];
    size_t n = sim->lattice->n_spins, n_temps = sim->n_temps;
    *sweep_id = snap->sweep_id;
    *mode_idx = snap->mode_idx;
    memcpy(cluster_ids, snap->cluster_ids, sizeof(uint32_t) * n_temps * n);
    if (snap->has_blue) memcpy(blue_ids, snap->blue_ids, sizeof(uint32_t) * n_temps * n);
    memcpy(spins, snap->spins, n_temps * 2 * n);
    memcpy(system_ids, snap->system_ids, sizeof(uint64_t) * n_temps * 2);
    return snap->has_blue;
}
