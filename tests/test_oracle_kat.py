"""Pins the parity oracle on the reference's own known-answer tests and on the
published xoshiro256** / SplitMix64 vectors.  CPU only.

Each test names the reference test it restates (paths relative to the reference
root, ss/ = spin-sim/src/).
"""

import ctypes as C

import numpy as np
import pytest

M64 = (1 << 64) - 1
GOLDEN_GAMMA = 0x9E3779B97F4A7C15


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- published generator vectors (not under the reference tree) ------------

def test_splitmix64_published_vectors(oracle):
    lib = oracle.load()
    # splitmix64.c reference outputs, seed 1234567 and seed 1477776061723855037
    for seed, expected in [
        (1234567, [6457827717110365317, 3203168211198807973, 9817491932198370423,
                   4593380528125082431, 16408922859458223821]),
        (1477776061723855037, [1985237415132408290, 2979275885539914483, 13511426838097143398,
                               8488337342461049707, 15141737807933549159, 17093170987380407015,
                               16389528042912955399, 13177319091862933652, 10841969400225389492,
                               17094824097954834098]),
    ]:
        x = seed
        got = []
        for _ in expected:
            got.append(lib.orc_splitmix64(x))
            x = (x + GOLDEN_GAMMA) & M64
        assert got == expected


def test_xoshiro256starstar_published_vector(oracle):
    lib = oracle.load()
    rng = oracle.OrcRng()
    for i in range(4):
        rng.s[i] = i + 1
    expected = [11520, 0, 1509978240, 1215971899390074240, 1216172134540287360,
                607988272756665600, 16172922978634559625, 8476171486693032832,
                10595114339597558777, 2904607092377533576]
    assert [lib.orc_rng_next_u64(C.byref(rng)) for _ in expected] == expected


def test_seed_from_u64_is_four_splitmix_outputs(oracle):
    lib = oracle.load()
    rng = oracle.OrcRng()
    lib.orc_rng_seed_from_u64(C.byref(rng), 1234567)
    assert list(rng.s) == [6457827717110365317, 3203168211198807973, 9817491932198370423,
                           4593380528125082431]


def test_f32_and_u32_share_the_high_bits(oracle):
    lib = oracle.load()
    ra, rb = oracle.make_rngs([99])[0], oracle.make_rngs([99])[0]
    for _ in range(200):
        word = lib.orc_rng_next_u32(C.byref(ra))
        f = lib.orc_rng_gen_f32(C.byref(rb))
        assert f == np.float32(word >> 8) * np.float32(2.0 ** -24)
        assert 0.0 <= f < 1.0


class _Pcg32:
    """rand_pcg 0.3 Lcg64Xsh32 -- the generator rand 0.8.5's own tests draw from (`crate::test::rng(seed)` =
    `Pcg32::new(seed, 11634580027462260723)`), restated here to feed the oracle's sampling code the words those tests saw."""
    MUL, MASK = 6364136223846793005, (1 << 64) - 1

    def __init__(self, state, stream=11634580027462260723):
        self.inc = ((stream << 1) | 1) & self.MASK
        self.state = (state + self.inc) & self.MASK
        self._step()

    def _step(self):
        self.state = (self.state * self.MUL + self.inc) & self.MASK

    def next_u32(self):
        st = self.state
        self._step()
        rot, xsh = st >> 59, (((st >> 18) ^ st) >> 27) & 0xFFFFFFFF
        return ((xsh >> rot) | (xsh << ((32 - rot) & 31))) & 0xFFFFFFFF


def test_range_sampling_reproduces_rand_085_value_stability_vectors(oracle):
    """rand 0.8.5 pins its own outputs: `distributions/uniform.rs`, test `value_stability` --
    `sample_single(11u32, 219)` three times on `test::rng(897)` gives [17, 66, 214]; `distributions/float.rs`, test
    `value_stability` -- `Standard` f32 on `test::rng(0x6f44f5646c2a7334)` gives [0.0035963655, 0.7346052, 0.09778172].
    The oracle's range sampling (gen_range_u32: the shuffles of build_tasks, overlap.rs:48) and its f32 conversion
    (orc_rng_gen_f32: every Metropolis draw, sweep.rs:179-181) run on those word streams and must give those values."""
    lib = oracle.load()
    rng = _Pcg32(897)
    words = np.array([rng.next_u32() for _ in range(64)], dtype=np.uint32)
    got, at = [], 0
    for _ in range(3):
        out = C.c_uint32()
        used = lib.orc_pin_sample_single_u32(11, 219, words[at:].ctypes.data, len(words) - at, C.byref(out))
        assert used >= 1
        at += used
        got.append(out.value)
    assert got == [17, 66, 214]
    # the same test's u8 line (`test_samples(11u8, 219, &[17, 66, 214], ..)`): small integer types take another zone formula
    # (uniform.rs, uniform_int_impl!) -- the one the oracle's gen_range_u8_inclusive uses for CMR's 1..=3 / 0..=3 draws
    got, at = [], 0
    for _ in range(3):
        out8 = C.c_uint8()
        used = lib.orc_pin_sample_single_inclusive_u8(11, 218, words[at:].ctypes.data, len(words) - at, C.byref(out8))
        assert used >= 1
        at += used
        got.append(out8.value)
    assert got == [17, 66, 214]
    rng = _Pcg32(0x6F44F5646C2A7334)
    floats = [lib.orc_pin_f32_of_word(rng.next_u32()) for _ in range(3)]
    assert [np.float32(f) for f in floats] == [np.float32(0.0035963655), np.float32(0.7346052), np.float32(0.09778172)]


def test_gen_range_and_shuffle_are_in_range_and_permute(oracle):
    lib = oracle.load()
    rng = oracle.make_rngs([7])[0]
    seen = set()
    for _ in range(2000):
        v = lib.orc_rng_gen_range_usize(C.byref(rng), 0, 23)
        assert 0 <= v < 23
        seen.add(v)
    assert seen == set(range(23))
    values = np.arange(9, dtype=np.uint64)
    lib.orc_rng_shuffle_usize(C.byref(rng), ptr(values), 9)
    assert sorted(values.tolist()) == list(range(9))


# ---- ss/geometry/lattice.rs:116-184 ---------------------------------------

def test_2d_neighbors(oracle):
    lat = oracle.Lattice([3, 4])
    assert lat.n_spins == 12 and lat.strides() == [4, 1]
    assert (lat.neighbor_fwd(0, 0), lat.neighbor_fwd(0, 1)) == (4, 1)
    assert (lat.neighbor_bwd(0, 0), lat.neighbor_bwd(0, 1)) == (8, 3)
    assert (lat.neighbor_fwd(11, 0), lat.neighbor_fwd(11, 1)) == (3, 8)


def test_3d_neighbors(oracle):
    lat = oracle.Lattice([2, 3, 4])
    assert lat.n_spins == 24 and lat.strides() == [12, 4, 1]
    assert [lat.neighbor_fwd(0, d) for d in range(3)] == [12, 4, 1]


def test_triangular_neighbors(oracle):
    lat = oracle.Lattice([4, 4], offsets=[[1, 0], [0, 1], [1, -1]])
    assert lat.n_neighbors == 3 and lat.n_spins == 16
    assert [lat.neighbor_fwd(0, d) for d in range(3)] == [4, 1, 7]
    assert [lat.neighbor_bwd(0, d) for d in range(3)] == [12, 3, 13]
    assert lat.neighbor_fwd(5, 2) == 8 and lat.neighbor_bwd(5, 2) == 2
    assert [lat.neighbor_fwd(15, d) for d in range(3)] == [3, 12, 2]


# ---- ss/spins/energy.rs:117-147 -------------------------------------------

def test_energy_into_path_matches_interaction_path(oracle):
    lib = oracle.load()
    lat = oracle.Lattice([2, 3])
    couplings = np.ones(lat.n_spins * lat.n_neighbors, dtype=np.float32)
    spins = np.array([1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, -1], dtype=np.int8)
    energies = np.zeros(2, dtype=np.float32)
    mags = np.zeros(2, dtype=np.int64)
    inter = np.zeros(2 * lat.n_spins * lat.n_neighbors, dtype=np.float32)
    lib.orc_compute_energies(lat.handle, ptr(spins), ptr(couplings), 2, ptr(energies), ptr(mags),
                             ptr(inter))
    assert mags.tolist() == [6, 0]
    assert energies[0] == 2.0  # aligned ferromagnet: E/N = +z'
    per_system = lat.n_spins * lat.n_neighbors
    for r in range(2):
        total = np.float32(0)
        for v in inter[r * per_system:(r + 1) * per_system]:
            total = np.float32(total + v)
        assert energies[r] == np.float32(total / np.float32(lat.n_spins))


# ---- ss/mcmc/tempering.rs:110-138 -----------------------------------------

def test_full_ladder_attempts_every_edge_in_requested_parity_order(oracle):
    lib = oracle.load()
    energies = np.zeros(5, dtype=np.float32)
    temps = np.array([0.5, 0.8, 1.2, 2.0, 4.0], dtype=np.float32)
    system_ids = np.arange(5, dtype=np.uint64)
    rng = oracle.make_rngs([17])[0]
    attempts = (oracle.OrcAttempt * 8)()
    n = lib.orc_parallel_tempering_full_ladder(ptr(energies), ptr(temps), ptr(system_ids), 5, 64,
                                               C.byref(rng), 0, C.byref(attempts))
    assert [attempts[i].edge for i in range(n)] == [0, 2, 1, 3]
    n = lib.orc_parallel_tempering_full_ladder(ptr(energies), ptr(temps), ptr(system_ids), 5, 64,
                                               C.byref(rng), 1, C.byref(attempts))
    assert [attempts[i].edge for i in range(n)] == [1, 3, 0, 2]


# ---- ss/simulation/realization.rs:267-302 ---------------------------------

def test_tracks_hot_cold_hot_across_attempts(oracle):
    lib = oracle.load()
    temps = np.array([0.5, 1.0, 2.0], dtype=np.float32)
    seq = [(1, 1, 2), (0, 0, 2), (0, 2, 0), (1, 2, 1)]
    attempts = (oracle.OrcAttempt * len(seq))()
    for k, (edge, left, right) in enumerate(seq):
        attempts[k].edge, attempts[k].accepted = edge, 1
        attempts[k].left_system, attempts[k].right_system = left, right
    att = np.zeros(2, dtype=np.uint64)
    acc = np.zeros(2, dtype=np.uint64)
    trips = np.zeros(3, dtype=np.uint64)
    lib.orc_pt_replay(1, 3, ptr(temps), C.byref(attempts), len(seq), ptr(att), ptr(acc), ptr(trips))
    assert att.tolist() == [2, 2] and acc.tolist() == [2, 2]
    assert trips[2] == 1


def test_reset_replays_spins_and_pt_state(oracle):
    sim = oracle.OracleSimulation([3, 3], np.ones((3, 3, 2), np.float32), [1.0, 2.0], 2, seed=17)
    initial = sim.get_spins()
    sim.sample(3, "metropolis", pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0)
    assert sim.get_pt_counters()[0].tolist() == [6]
    sim.reset()
    np.testing.assert_array_equal(sim.get_spins(), initial)
    assert sim.get_system_ids().tolist() == [0, 1, 2, 3]
    assert sim.get_pt_counters()[0].tolist() == [0]


# ---- ss/mcmc/sweep.rs:345-442 ---------------------------------------------

def test_unit_lookup_is_fail_closed(oracle):
    lib = oracle.load()

    def make(c, t):
        c = np.array(c, dtype=np.float32)
        t = np.array(t, dtype=np.float32)
        h = lib.orc_lookup_new(ptr(c), len(c), ptr(t), len(t), 2)
        if h:
            lib.orc_lookup_free(h)
        return bool(h)

    assert make([-1.0, 0.0, 1.0], [0.5, 2.0])
    assert not make([-2.0, 2.0], [1.0])
    assert not make([-1.0, np.nan], [1.0])
    assert not make([-1.0, 1.0], [0.0])
    assert not make([-1.0, 1.0], [np.frombuffer(np.uint32(1).tobytes(), np.float32)[0]])


def test_unit_lookup_cutoffs_match_legacy_boundaries(oracle):
    lib = oracle.load()
    temps = np.array([0.7, 2.0, 5.0], dtype=np.float32)
    c = np.array([-1.0, 0.0, 1.0], dtype=np.float32)
    h = lib.orc_lookup_new(ptr(c), 3, ptr(temps), 3, 3)
    for tid, t in enumerate(temps):
        for e in range(-6, 7):
            count = lib.orc_lookup_get(h, tid, e)
            if count > 0:
                assert lib.orc_lookup_legacy_accepts(float(t), e, count - 1)
            if count < (1 << 24):
                assert not lib.orc_lookup_legacy_accepts(float(t), e, count)
            if e >= 0:
                assert count == 1 << 24
    lib.orc_lookup_free(h)


@pytest.mark.parametrize("offsets", [None, [[1, 0], [0, 1]]])
def test_unit_lookup_matches_log_with_permuted_systems(oracle, offsets):
    lib = oracle.load()
    lat = oracle.Lattice([8, 8], offsets=offsets)
    n = lat.n_spins
    couplings = np.array([[-1.0, 0.0, 1.0][i % 3] for i in range(n * lat.n_neighbors)], np.float32)
    temps = np.array([0.7, 2.0, 5.0], dtype=np.float32)
    system_ids = np.array([2, 0, 1], dtype=np.uint64)
    initial = np.array([-1 if i % 5 == 0 else 1 for i in range(3 * n)], dtype=np.int8)
    lookup = lib.orc_lookup_new(ptr(couplings), len(couplings), ptr(temps), 3, lat.n_neighbors)
    table_spins, log_spins = initial.copy(), initial.copy()
    table_rngs, log_rngs = oracle.make_rngs([71, 72, 73]), oracle.make_rngs([71, 72, 73])
    for _ in range(20):
        lib.orc_metropolis_sweep(lat.handle, ptr(table_spins), ptr(couplings), ptr(temps),
                                 ptr(system_ids), 3, C.byref(table_rngs), lookup)
        lib.orc_metropolis_sweep(lat.handle, ptr(log_spins), ptr(couplings), ptr(temps),
                                 ptr(system_ids), 3, C.byref(log_rngs), None)
    np.testing.assert_array_equal(table_spins, log_spins)
    assert not np.array_equal(table_spins, initial)
    for a, b in zip(table_rngs, log_rngs):
        assert lib.orc_rng_next_u64(C.byref(a)) == lib.orc_rng_next_u64(C.byref(b))
    lib.orc_lookup_free(lookup)


def test_square_specialization_matches_generic_trajectory(oracle):
    # sweep.rs:293-343: 5x7, non-unit couplings, log path; canonical 2D vs with_offsets.
    lib = oracle.load()
    spec = oracle.Lattice([5, 7])
    gen = oracle.Lattice([5, 7], offsets=[[1, 0], [0, 1]])
    n = spec.n_spins
    initial = np.array([-1 if i % 3 == 0 else 1 for i in range(2 * n)], dtype=np.int8)
    couplings = np.array([((i % 7) - 3.0) / 3.0 for i in range(n * 2)], dtype=np.float32)
    temps = np.array([0.8, 2.5], dtype=np.float32)
    ids = np.array([0, 1], dtype=np.uint64)
    a_spins, b_spins = initial.copy(), initial.copy()
    a_rngs, b_rngs = oracle.make_rngs([42, 43]), oracle.make_rngs([42, 43])
    lib.orc_metropolis_sweep(spec.handle, ptr(a_spins), ptr(couplings), ptr(temps), ptr(ids), 2,
                             C.byref(a_rngs), None)
    lib.orc_metropolis_sweep(gen.handle, ptr(b_spins), ptr(couplings), ptr(temps), ptr(ids), 2,
                             C.byref(b_rngs), None)
    np.testing.assert_array_equal(a_spins, b_spins)
    for a, b in zip(a_rngs, b_rngs):
        assert lib.orc_rng_next_u64(C.byref(a)) == lib.orc_rng_next_u64(C.byref(b))


# ---- ss/clusters/utils.rs:491-495,690-705,766-781 --------------------------

def test_uf_csd_bond_based_and_top4(oracle):
    lib = oracle.load()
    lat = oracle.Lattice([4, 4])
    bonds = [(0, 0), (0, 1), (3, 1), (10, 0), (10, 1)]
    sites = np.array([b[0] for b in bonds], dtype=np.uint64)
    dims = np.array([b[1] for b in bonds], dtype=np.uint64)
    parent = np.zeros(16, dtype=np.uint32)
    counts = np.zeros(16, dtype=np.uint32)
    lib.orc_uf_from_bonds(lat.handle, ptr(sites), ptr(dims), len(bonds), ptr(parent), ptr(counts))
    clusters = {}
    for i, p in enumerate(parent):
        clusters.setdefault(int(p), set()).add(i)
    big = sorted((c for c in clusters.values() if len(c) > 1), key=len)
    assert big == [{10, 11, 14}, {0, 1, 3, 4}]
    hist = np.zeros(17, dtype=np.uint64)
    lib.orc_uf_histogram(ptr(counts), 16, ptr(hist))
    assert (hist[1], hist[3], hist[4], hist.sum()) == (9, 1, 1, 11)
    top = np.zeros(4, dtype=np.uint32)
    lib.orc_top4_sizes(ptr(counts), 16, ptr(top))
    assert top.tolist() == [4, 3, 1, 1]


# ---- statistical fixtures of the reference: tests/binder_crossings.py:21-71, tests/utils.py:7,39-47

def test_2d_ferromagnet_binder_curves_cross_at_the_exact_tc(oracle):
    """The reference's own physics check of Metropolis + Swendsen-Wang (fk.rs) + tempering:
    2D square ferromagnet, 32 temperatures T_c +- 0.3, 2 replicas, seed 42, 10000 sweeps with 25 %
    warm-up; Binder cumulants of L = 8, 16, 32 agree at the exact T_c within 0.05.  (The couplings
    of a ferromagnet are all +1, so the Python-side seed material does not enter.)"""
    tc = 2.0 / np.log(1 + np.sqrt(2))
    temps = np.linspace(tc - 0.3, tc + 0.3, 32).astype(np.float32)
    at_tc = []
    for size in (8, 16, 32):
        sim = oracle.OracleSimulation([size, size], np.ones((size, size, 2), np.float32), temps, 2,
                                      None, 42)
        res = sim.sample(10000, "metropolis", pt_interval=1, warmup_ratio=0.25,
                         cluster_update_interval=1, cluster_mode="sw")
        binder = 1 - res["mags4"] / (3 * res["mags2"] ** 2)
        at_tc.append(float(np.interp(tc, temps, binder)))
    assert max(at_tc) - min(at_tc) < 0.05, at_tc


# ---- graph observers (reference spin-sim/src/clusters/utils.rs:736-781, simulation/mod.rs:949-1177) ----
def _observe(oracle, lattice, bonds):
    lib = oracle.load()
    sites = np.array([b[0] for b in bonds], dtype=np.uint64)
    dims = np.array([b[1] for b in bonds], dtype=np.uint64)
    out = np.zeros(9, dtype=np.uint32)
    lib.orc_graph_observation(lattice.handle, oracle._ptr(sites), oracle._ptr(dims), len(bonds), oracle._ptr(out))
    return {"active_bonds": int(out[0]), "top4": out[1:5].tolist(), "large": int(out[5]),
            "winding": (bool(out[6]), bool(out[7])), "observed": bool(out[8])}


def test_winding_distinguishes_seams_and_noncontractible_cycles(oracle):
    lat = oracle.Lattice([4, 4])
    index = lambda row, col: 4 * row + col  # noqa: E731
    assert _observe(oracle, lat, [(index(0, 3), 1)])["winding"] == (False, False)
    assert _observe(oracle, lat, [(index(1, c), 1) for c in range(4)])["winding"] == (False, True)
    both = [(index(0, c), 1) for c in range(4)] + [(index(r, 0), 0) for r in range(4)]
    assert _observe(oracle, lat, both)["winding"] == (True, True)


def test_bond_metrics_reuse_component_counts(oracle):
    got = _observe(oracle, oracle.Lattice([4, 4]), [(0, 0), (0, 1), (3, 1), (10, 0), (10, 1)])
    assert got["active_bonds"] == 5 and got["top4"] == [4, 3, 1, 1] and got["observed"]
    # a lattice built from explicit offsets has no canonical square shape: winding is not tracked
    lat = oracle.Lattice([4, 4], offsets=[[1, 0], [0, 1]])
    assert _observe(oracle, lat, [(4 * 1 + c, 1) for c in range(4)])["winding"] == (False, False)


def test_cluster_histograms_appear_only_when_collected(oracle):
    J = np.ones((4, 4, 2), dtype=np.float32)
    temps = np.array([2.0], dtype=np.float32)
    kw = dict(warmup_ratio=0.0, cluster_update_interval=1, cluster_mode="sw")
    plain = oracle.OracleSimulation([4, 4], J, temps, 1, None, 42).sample(1, "metropolis", **kw)
    assert "fk_csd" not in plain
    stats = oracle.OracleSimulation([4, 4], J, temps, 1, None, 42).sample(1, "metropolis", collect_cluster_stats=True, **kw)
    assert len(stats["fk_csd"]) == 1 and stats["fk_csd"][0].sum() > 0
    assert int((np.arange(17) * stats["fk_csd"][0]).sum()) == 16
    kw = dict(warmup_ratio=0.0, overlap_cluster_update_interval=1, overlap_cluster_mode="sw")
    plain = oracle.OracleSimulation([4, 4], J, temps, 2, None, 42).sample(1, "metropolis", **kw)
    assert "overlap_csd" not in plain and "top_cluster_sizes" not in plain
    stats = oracle.OracleSimulation([4, 4], J, temps, 2, None, 42).sample(1, "metropolis", collect_cluster_stats=True, **kw)
    assert len(stats["overlap_csd"]) == 1 and stats["overlap_csd"][0][0].sum() > 0


def test_observe_records_without_changing_spins(oracle):
    J = np.ones((4, 4, 2), dtype=np.float32)
    for n_rep, temp, seed, kw, kind in (
            (1, 2.0, 19, dict(cluster_update_interval=1, cluster_mode="sw", cluster_action="observe"), "fk"),
            (2, 1.5, 23, dict(overlap_cluster_update_interval=1, overlap_cluster_build_mode="cmr",
                              overlap_cluster_mode="sw", overlap_cluster_action="observe"), "cmr_blue")):
        temps = np.array([temp], dtype=np.float32)
        baseline = oracle.OracleSimulation([4, 4], J, temps, n_rep, None, seed)
        observed = oracle.OracleSimulation([4, 4], J, temps, n_rep, None, seed)
        baseline.sample(1, "metropolis", warmup_ratio=0.0)
        result = observed.sample(1, "metropolis", warmup_ratio=0.0, **kw)
        np.testing.assert_array_equal(observed.get_spins(), baseline.get_spins())
        np.testing.assert_array_equal(observed.get_system_ids(), baseline.get_system_ids())
        summary = result["per_disorder"]["cluster_observations"][kind]
        assert summary["observation_count"].tolist() == [[1]]
        assert int((np.arange(17) * summary["cluster_size_counts"][0, 0]).sum()) == 16
