#include "host_util.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

#include <mutex>
#include <set>
#include <string>

namespace pa {

// reference: spin-sim/src/simulation/realization.rs:9-15 (same body as src/lib.rs:22-28)
uint64_t splitmix64(uint64_t value) {
    value += 0x9E3779B97F4A7C15ULL;
    uint64_t z = value;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// realization.rs:17-19
uint64_t child_seed(uint64_t root, uint64_t domain, uint64_t index) {
    return splitmix64(root ^ domain ^ splitmix64(index));
}

// src/lib.rs:30-32
uint64_t realization_seed(uint64_t root, uint64_t realization) {
    return splitmix64(root ^ splitmix64(realization));
}

// rand_xoshiro 0.6.0 seed_from_u64: four SplitMix64 outputs from state `seed`.
Xoshiro Xoshiro::from_seed_u64(uint64_t seed) {
    Xoshiro rng;
    uint64_t state = seed;
    for (int i = 0; i < 4; ++i) {
        rng.s[i] = splitmix64(state);
        state += 0x9E3779B97F4A7C15ULL;
    }
    return rng;
}

// ---------------------------------------------------------------------------
// GF(2) jump-ahead.  One xoshiro step is linear over GF(2) in the 256 state
// bits, so n steps are a 256x256 bit matrix T^n; the device splits each
// system's stream into blocks and enters block b with T^(b*block) * cursor.
// ---------------------------------------------------------------------------

JumpMatrix JumpMatrix::identity() {
    JumpMatrix m;
    m.cols.assign(256 * 4, 0);
    for (int j = 0; j < 256; ++j) m.cols[j * 4 + j / 64] = 1ULL << (j % 64);
    return m;
}

void JumpMatrix::apply(const uint64_t in[4], uint64_t out[4]) const {
    uint64_t acc[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) {
        uint64_t bits = in[w];
        while (bits) {
            int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint64_t *col = &cols[(w * 64 + b) * 4];
            acc[0] ^= col[0];
            acc[1] ^= col[1];
            acc[2] ^= col[2];
            acc[3] ^= col[3];
        }
    }
    std::memcpy(out, acc, sizeof(acc));
}

JumpMatrix JumpMatrix::compose(const JumpMatrix &first) const {
    JumpMatrix r;
    r.cols.resize(256 * 4);
    for (int j = 0; j < 256; ++j) apply(&first.cols[j * 4], &r.cols[j * 4]);
    return r;
}

JumpMatrix JumpMatrix::steps(uint64_t n_steps) {
    // single-step matrix: images of the unit vectors under one generator step
    JumpMatrix one;
    one.cols.resize(256 * 4);
    for (int j = 0; j < 256; ++j) {
        Xoshiro g;
        g.s[0] = g.s[1] = g.s[2] = g.s[3] = 0;
        g.s[j / 64] = 1ULL << (j % 64);
        g.next_u64();
        std::memcpy(&one.cols[j * 4], g.s, 32);
    }
    JumpMatrix result = identity();
    JumpMatrix power = one;
    while (n_steps) {
        if (n_steps & 1) result = power.compose(result);
        n_steps >>= 1;
        if (n_steps) power = power.compose(power);
    }
    return result;
}

void JumpMatrix::nibble_table(uint32_t *out) const {
    for (int p = 0; p < 64; ++p) {
        for (int v = 0; v < 16; ++v) {
            uint64_t acc[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4; ++i)
                if (v & (1 << i)) {
                    const uint64_t *col = &cols[(p * 4 + i) * 4];
                    acc[0] ^= col[0];
                    acc[1] ^= col[1];
                    acc[2] ^= col[2];
                    acc[3] ^= col[3];
                }
            // layout [position][half][value][4 dwords] (device_util.hpp jump_state_lds: bank-conflict-free gathers);
            // dword j of the 256-bit row = half j / 4, entry j % 4
            const uint32_t row[8] = {static_cast<uint32_t>(acc[0]), static_cast<uint32_t>(acc[0] >> 32),
                                     static_cast<uint32_t>(acc[1]), static_cast<uint32_t>(acc[1] >> 32),
                                     static_cast<uint32_t>(acc[2]), static_cast<uint32_t>(acc[2] >> 32),
                                     static_cast<uint32_t>(acc[3]), static_cast<uint32_t>(acc[3] >> 32)};
            for (int j = 0; j < 8; ++j)
                out[(static_cast<size_t>(p) * 2 + j / 4) * 64 + static_cast<size_t>(v) * 4 + j % 4] = row[j];
        }
    }
}

// ---------------------------------------------------------------------------
// Lattice: row-major sites, periodic wrap with Euclidean remainder
// (spin-sim/src/geometry/lattice.rs:44-99).
// ---------------------------------------------------------------------------

LatticeTables LatticeTables::build(uint32_t n_dims, const uint64_t *shape, uint32_t n_neighbors,
                                   const int64_t *offsets) {
    LatticeTables lat;
    lat.n_dims = n_dims;
    lat.shape.assign(shape, shape + n_dims);
    lat.strides.assign(n_dims, 1);
    for (int d = static_cast<int>(n_dims) - 2; d >= 0; --d)
        lat.strides[d] = lat.strides[d + 1] * lat.shape[d + 1];
    lat.n_spins = 1;
    for (uint32_t d = 0; d < n_dims; ++d) lat.n_spins *= lat.shape[d];

    std::vector<int64_t> unit;
    if (!offsets) {
        n_neighbors = n_dims;
        unit.assign(static_cast<size_t>(n_dims) * n_dims, 0);
        for (uint32_t d = 0; d < n_dims; ++d) unit[d * n_dims + d] = 1;
        offsets = unit.data();
    }
    lat.n_neighbors = n_neighbors;
    lat.fwd.resize(lat.n_spins * n_neighbors);
    lat.bwd.resize(lat.n_spins * n_neighbors);
    std::vector<int64_t> coords(n_dims);
    for (uint64_t site = 0; site < lat.n_spins; ++site) {
        for (uint32_t d = 0; d < n_dims; ++d)
            coords[d] = static_cast<int64_t>((site / lat.strides[d]) % lat.shape[d]);
        for (uint32_t k = 0; k < n_neighbors; ++k) {
            uint64_t plus = 0, minus = 0;
            for (uint32_t d = 0; d < n_dims; ++d) {
                const int64_t extent = static_cast<int64_t>(lat.shape[d]);
                int64_t up = (coords[d] + offsets[k * n_dims + d]) % extent;
                int64_t down = (coords[d] - offsets[k * n_dims + d]) % extent;
                if (up < 0) up += extent;
                if (down < 0) down += extent;
                plus += static_cast<uint64_t>(up) * lat.strides[d];
                minus += static_cast<uint64_t>(down) * lat.strides[d];
            }
            lat.fwd[site * n_neighbors + k] = static_cast<uint32_t>(plus);
            lat.bwd[site * n_neighbors + k] = static_cast<uint32_t>(minus);
        }
    }
    return lat;
}

// ---------------------------------------------------------------------------
// Acceptance table (spin-sim/src/mcmc/sweep.rs:99-167): number of 24-bit draws
// accepted at (T, dE), found by bisection on the reference's f32 predicate.
// logf is the platform libm, as for the reference's f32::ln.
// ---------------------------------------------------------------------------

static inline bool log_rule_accepts(float temperature, int32_t energy_change, uint32_t draw) {
    const float uniform = static_cast<float>(draw) / 16777216.0f;
    return static_cast<float>(energy_change) >= (temperature / 2.0f) * logf(uniform);
}

uint32_t accepted_count(float temperature, int32_t energy_change) {
    uint32_t lo = 0, hi = 1u << 24;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (log_rule_accepts(temperature, energy_change, mid))
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

bool build_metropolis_table(const float *couplings, uint64_t n_couplings, const float *temps,
                            uint64_t n_temps, uint64_t n_neighbors, std::vector<uint32_t> &out) {
    for (uint64_t i = 0; i < n_couplings; ++i) {
        const float c = couplings[i];
        if (!(c == -1.0f || c == 0.0f || c == 1.0f)) return false;
    }
    for (uint64_t t = 0; t < n_temps; ++t)
        if (!(std::isfinite(temps[t]) && temps[t] / 2.0f > 0.0f)) return false;
    const int32_t offset = static_cast<int32_t>(2 * n_neighbors);
    const uint64_t width = 4 * n_neighbors + 1;
    out.resize(n_temps * width);
    for (uint64_t t = 0; t < n_temps; ++t)
        for (int32_t e = -offset; e <= offset; ++e)
            out[t * width + static_cast<uint64_t>(e + offset)] = accepted_count(temps[t], e);
    return true;
}

namespace {

template <typename F> std::vector<float> fill_uniform_table(F fn) {
    std::vector<float> table(1u << 24);
    const unsigned n_threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<std::thread> workers;
    for (unsigned k = 0; k < n_threads; ++k)
        workers.emplace_back([&, k] {
            const uint32_t lo = static_cast<uint32_t>((1ull << 24) * k / n_threads);
            const uint32_t hi = static_cast<uint32_t>((1ull << 24) * (k + 1) / n_threads);
            for (uint32_t v = lo; v < hi; ++v) table[v] = fn(static_cast<float>(v) / 16777216.0f);
        });
    for (auto &t : workers) t.join();
    return table;
}

} // namespace

const std::vector<float> &uniform_log_table() {
    static const std::vector<float> table = fill_uniform_table([](float u) { return logf(u); });
    return table;
}

const std::vector<float> &uniform_logit_table(bool *monotone) {
    static const std::vector<float> table =
        fill_uniform_table([](float u) { return logf(u / (1.0f - u)); });
    static const bool rising = [] {
        for (size_t k = 1; k < table.size(); ++k)
            if (table[k] < table[k - 1]) return false;
        return true;
    }();
    if (monotone) *monotone = rising;
    return table;
}

bool build_gibbs_table(const float *couplings, uint64_t n_couplings, const float *temps,
                       uint64_t n_temps, uint64_t n_neighbors, std::vector<uint32_t> &out) {
    for (uint64_t i = 0; i < n_couplings; ++i) {
        const float c = couplings[i];
        if (!(c == -1.0f || c == 0.0f || c == 1.0f)) return false;
    }
    for (uint64_t t = 0; t < n_temps; ++t)
        if (!(std::isfinite(temps[t]) && temps[t] / 2.0f > 0.0f)) return false;
    bool monotone = false;
    const std::vector<float> &logit = uniform_logit_table(&monotone);
    if (!monotone) return false;
    const int32_t offset = static_cast<int32_t>(2 * n_neighbors);
    const uint64_t width = 4 * n_neighbors + 1;
    out.resize(n_temps * width);
    for (uint64_t t = 0; t < n_temps; ++t) {
        const float half_t = temps[t] / 2.0f;
        for (int32_t e = -offset; e <= offset; ++e) {
            uint32_t lo = 0, hi = 1u << 24;
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (static_cast<float>(e) >= half_t * logit[mid])
                    lo = mid + 1;
                else
                    hi = mid;
            }
            out[t * width + static_cast<uint64_t>(e + offset)] = lo;
        }
    }
    return true;
}

namespace {

// in-place iterative radix-2 FFT of re/im (length a power of two); inverse = conjugate twiddles, unscaled
void fft_radix2(std::vector<double> &re, std::vector<double> &im, bool inverse) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) { // bit reversal
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    const double pi = 3.14159265358979323846;
    for (size_t len = 2; len <= n; len <<= 1) {
        const double angle = (inverse ? 2.0 : -2.0) * pi / static_cast<double>(len);
        for (size_t start = 0; start < n; start += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = std::cos(angle * static_cast<double>(k)), wi = std::sin(angle * static_cast<double>(k));
                const size_t a = start + k, b = a + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr;
                im[b] = im[a] - xi;
                re[a] += xr;
                im[a] += xi;
            }
    }
}

} // namespace

void autocorrelation_tau(const float *series, size_t n_samples, size_t n_cols, size_t max_lag, double *tau_out, bool fft) {
    std::vector<double> sum_prod(max_lag + 1), gamma(max_lag + 1);
    std::vector<double> re, im;
    for (size_t t = 0; t < n_cols; ++t) {
        double sum_o = 0.0, sum_o2 = 0.0;
        std::fill(sum_prod.begin(), sum_prod.end(), 0.0);
        // push(): every accumulator receives its terms in sample order; the ring only bounds memory
        for (size_t i = 0; i < n_samples; ++i) {
            const double o = static_cast<double>(series[i * n_cols + t]);
            sum_o += o;
            sum_o2 += o * o;
            if (fft) continue;
            const size_t n_back = std::min(i, max_lag);
            for (size_t delta = 0; delta <= n_back; ++delta)
                sum_prod[delta] += o * static_cast<double>(series[(i - delta) * n_cols + t]);
        }
        if (fft && n_samples > 0) {
            // finish_fft (autocorrelation.rs:126-163): zero-pad to a power of two >= 2 n, power spectrum, inverse
            // transform / fft_len = the lagged products.  Equal to the ring sums up to rounding (the reference's
            // own test allows 1e-10 between its two backends); the transform here is a plain radix-2 one, not rustfft's.
            size_t fft_len = 1;
            while (fft_len < 2 * n_samples) fft_len <<= 1;
            re.assign(fft_len, 0.0);
            im.assign(fft_len, 0.0);
            for (size_t i = 0; i < n_samples; ++i) re[i] = static_cast<double>(series[i * n_cols + t]);
            fft_radix2(re, im, false);
            for (size_t i = 0; i < fft_len; ++i) {
                re[i] = re[i] * re[i] + im[i] * im[i];
                im[i] = 0.0;
            }
            fft_radix2(re, im, true);
            for (size_t delta = 0; delta <= max_lag && delta < fft_len; ++delta) sum_prod[delta] = re[delta] / static_cast<double>(fft_len);
        }
        for (size_t delta = 0; delta <= max_lag; ++delta) gamma[delta] = delta == 0 ? 1.0 : 0.0;
        if (n_samples > 0) {
            const double m = static_cast<double>(n_samples);
            const double mean = sum_o / m;
            const double var = sum_o2 / m - mean * mean;
            if (var > 0.0)
                for (size_t delta = 0; delta <= max_lag; ++delta) {
                    const double count = n_samples > delta ? static_cast<double>(n_samples - delta) : 0.0;
                    gamma[delta] = count <= 0.0 ? (delta == 0 ? 1.0 : 0.0) : (sum_prod[delta] / count - mean * mean) / var;
                }
        }
        double tau = 0.5; // sokal_tau
        for (size_t w = 1; w <= max_lag; ++w) {
            tau += gamma[w];
            if (static_cast<double>(w) >= 5.0 * tau) break;
        }
        tau_out[t] = tau;
    }
}

// ---------------------------------------------------------------------------
// Dependency levels.  In the reference a sweep visits sites 0..N-1 in order
// (sweep.rs:58-61); site i therefore sees the NEW value of every neighbour
// j < i and the OLD value of every neighbour j > i.  Any schedule in which
// each site runs after all its smaller neighbours and before all its larger
// ones gives the identical result; the longest-path layering below is the
// shortest such schedule (hyperplanes x+y+z = const on a hypercubic lattice).
// ---------------------------------------------------------------------------

uint32_t build_levels(const LatticeTables &lat, std::vector<uint32_t> &level) {
    level.assign(lat.n_spins, 0);
    uint32_t n_levels = lat.n_spins ? 1 : 0;
    for (uint64_t site = 0; site < lat.n_spins; ++site) {
        uint32_t lvl = 0;
        for (uint32_t k = 0; k < lat.n_neighbors; ++k) {
            const uint32_t a = lat.fwd[site * lat.n_neighbors + k];
            const uint32_t b = lat.bwd[site * lat.n_neighbors + k];
            if (a < site && level[a] + 1 > lvl) lvl = level[a] + 1;
            if (b < site && level[b] + 1 > lvl) lvl = level[b] + 1;
        }
        level[site] = lvl;
        if (lvl + 1 > n_levels) n_levels = lvl + 1;
    }
    return n_levels;
}

} // namespace pa

// The library's version string carries the hash of the sources it was built from (peapods_amd/build.py
// source_hash): bench.py prints counter-derived figures only when profiles/counters.json was collected from the
// same kernels.
namespace pa {
namespace {
std::mutex g_switch_mutex;
std::set<std::string> g_switches;   // honoured so far
std::string g_switch_text;          // what pa_active_switches last returned (the pointer stays valid)
} // namespace

const char *dev_env(const char *name) {
    const char *gate = std::getenv("PEAPODS_DEVELOPER"); // (read every time: a test may open the gate for one handle)
    if (!gate || gate[0] != '1' || gate[1] != '\0') return nullptr;
    const char *value = std::getenv(name);
    if (value) {
        std::lock_guard<std::mutex> lock(g_switch_mutex);
        g_switches.insert(std::string(name) + "=" + value);
    }
    return value;
}
} // namespace pa

extern "C" const char *pa_active_switches(void) {
    std::lock_guard<std::mutex> lock(pa::g_switch_mutex);
    pa::g_switch_text.clear();
    for (const auto &s : pa::g_switches) pa::g_switch_text += (pa::g_switch_text.empty() ? "" : " ") + s;
    return pa::g_switch_text.c_str();
}

#ifndef PA_SOURCE_HASH
#define PA_SOURCE_HASH "unknown"
#endif
extern "C" const char *pa_version(void) { return "peapods_amd 0.3 (gfx950) src:" PA_SOURCE_HASH; }
