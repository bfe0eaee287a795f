// Host-side building blocks of the MI355X spin-sim engine: seed tree, the
// xoshiro256** generator (stepping and GF(2) jump-ahead), lattice neighbour
// tables, the Metropolis acceptance table and the dependency levels that turn
// the reference's sequential typewriter sweep into parallel wavefront steps.
// Pure C++ (no device code); unit-tested on CPU through the C ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pa {

constexpr uint64_t kSystemSeedDomain = 0x53A917E14C2D8B6FULL; // realization.rs:6
constexpr uint64_t kPairSeedDomain = 0xA1475EED91C302D7ULL;   // realization.rs:7

uint64_t splitmix64(uint64_t value);
uint64_t child_seed(uint64_t root, uint64_t domain, uint64_t index);
uint64_t realization_seed(uint64_t root, uint64_t realization);

// xoshiro256** with the rand / rand_xoshiro sampling rules the reference relies on.
struct Xoshiro {
    uint64_t s[4];
    static Xoshiro from_seed_u64(uint64_t seed);
    inline uint64_t next_u64() {
        const uint64_t mul5 = s[1] * 5;
        const uint64_t result = ((mul5 << 7) | (mul5 >> 57)) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = (s[3] << 45) | (s[3] >> 19);
        return result;
    }
    inline uint32_t next_u32() { return static_cast<uint32_t>(next_u64() >> 32); }
    inline uint32_t next_draw24() { return static_cast<uint32_t>(next_u64() >> 40); }
};

// 256x256 matrix over GF(2) acting on the xoshiro state, stored as the images of
// the 256 unit vectors (column j = M * e_j, 4 x u64 each).
struct JumpMatrix {
    std::vector<uint64_t> cols; // [256][4]
    static JumpMatrix identity();
    static JumpMatrix steps(uint64_t n_steps); // T^n by repeated squaring
    JumpMatrix compose(const JumpMatrix &first) const; // this * first
    void apply(const uint64_t in[4], uint64_t out[4]) const;
    // Nibble tables used on the device: entry [p][v] (p = 0..63, v = 0..15) is
    // M * (v << 4p) as 8 little-endian dwords.
    void nibble_table(uint32_t *out /* [64 positions][2 halves][16 values][4] */) const;
};

struct LatticeTables {
    uint32_t n_dims = 0;
    std::vector<uint64_t> shape, strides;
    uint64_t n_spins = 0;
    uint32_t n_neighbors = 0;
    std::vector<uint32_t> fwd, bwd; // [n_spins][n_neighbors]
    // offsets == nullptr -> hypercubic unit vectors (lattice.rs:31-38, offsets.rs:4-13)
    static LatticeTables build(uint32_t n_dims, const uint64_t *shape, uint32_t n_neighbors,
                               const int64_t *offsets);
};

// UnitCouplingMetropolisLookup (sweep.rs:99-167).  Returns false when the
// reference declines to build the table.
bool build_metropolis_table(const float *couplings, uint64_t n_couplings, const float *temps,
                            uint64_t n_temps, uint64_t n_neighbors, std::vector<uint32_t> &out);
uint32_t accepted_count(float temperature, int32_t energy_change);

// ln(u) and ln(u / (1 - u)) for every 24-bit uniform u = k * 2^-24, computed once per process
// with the host libm (the reference's f32::ln; sweep.rs:256,279-282, tempering.rs:89).
// *monotone tells whether the table never decreases with k (then "e >= (T/2) * table[k]" is a
// prefix of the draws and a threshold describes it exactly).
const std::vector<float> &uniform_log_table();
const std::vector<float> &uniform_logit_table(bool *monotone);

// Gibbs counterpart of the table for unit couplings: out[t * (4z'+1) + (e + 2z')] = number of
// draws k with e >= (T/2) ln(u / (1 - u)) (sweep.rs:262-284 evaluated on the 2^24 grid).
// Returns false when the couplings / temperatures are outside the table's domain or the
// predicate is not a prefix of the draws.
bool build_gibbs_table(const float *couplings, uint64_t n_couplings, const float *temps,
                       uint64_t n_temps, uint64_t n_neighbors, std::vector<uint32_t> &out);

// Integrated autocorrelation time of every column of series[n_samples][n_cols] (f32 values as
// pushed): AutocorrAccum ring backend + normalize_products + sokal_tau
// (ss/statistics/autocorrelation.rs:70-124,165-208) with the same f64 operation order.
void autocorrelation_tau(const float *series, size_t n_samples, size_t n_cols, size_t max_lag, double *tau_out,
                         bool fft = false);

// Developer switches.  Kernel selection and tuning knobs (PEAPODS_NO_RELAY, PEAPODS_STREAMS, PEAPODS_RNG_BLOCK, ...) are
// read through dev_env only, and dev_env answers null unless PEAPODS_DEVELOPER=1 is set as well: a stray variable in a
// user's environment cannot change which kernels run.  Every switch that was honoured is remembered and reported by
// pa_active_switches() (include/peapods_amd.h) and in bench.py's line.
const char *dev_env(const char *name);

// level[i] = 1 + max(level[j]) over neighbours j < i (0 if none).  Returns level count.
uint32_t build_levels(const LatticeTables &lat, std::vector<uint32_t> &level);

} // namespace pa
