"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer and index results must be bit-exact; floating-point
observables are compared bit for bit too (they are one IEEE division / f64 running
sums of exact integers), which is tighter than the 1e-6 the north star allows.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bimodal(shape, nn, n_dis, seed):
    rng = np.random.default_rng(seed)
    full = (n_dis,) + tuple(shape) + (nn,) if n_dis > 1 else tuple(shape) + (nn,)
    return (2 * rng.integers(0, 2, size=full) - 1).astype(np.float32)


def make_pair(oracle, shape, couplings, temps, n_replicas, seed, offsets=None):
    from peapods_amd import IsingSimulation

    temps = np.asarray(temps, dtype=np.float32)
    gpu = IsingSimulation(list(shape), couplings, temps, n_replicas, offsets, seed)
    cpu = oracle.OracleSimulation(list(shape), couplings, temps, n_replicas, offsets, seed)
    return gpu, cpu


def assert_state_equal(gpu, cpu, check_energy=True):
    for r in range(gpu.n_realizations):
        np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))
        np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r))
        for s in range(gpu.n_systems):
            np.testing.assert_array_equal(gpu.get_rng_state(r, s), cpu.get_rng_state(r, s))
        if check_energy:
            np.testing.assert_array_equal(gpu.get_energies(r), cpu.get_energies(r))


def assert_results_equal(a, b):
    assert set(a.keys()) == set(b.keys())
    for key in a:
        if key == "per_disorder":
            for k2 in a[key]["parallel_tempering"]:
                np.testing.assert_array_equal(a[key]["parallel_tempering"][k2],
                                              b[key]["parallel_tempering"][k2], err_msg=k2)
        elif key == "overlap_histogram":
            for x, y in zip(a[key], b[key]):
                np.testing.assert_array_equal(x, y)
        else:
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)


CASES = [
    # shape, n_temps, n_replicas, n_disorder, offsets
    ((4, 4), 2, 2, 1, None),
    ((3, 5), 3, 1, 2, None),            # odd extents: no checkerboard exists, levels still do
    ((8, 8), 3, 1, 1, [[1, 0], [0, 1]]),
    ((4, 4), 2, 2, 1, [[1, 0], [0, 1], [1, -1]]),  # triangular
    ((2, 3, 4), 4, 2, 2, None),         # extent-2 dimension: fwd and bwd bonds coincide in site
    ((6, 6, 6), 5, 4, 3, None),
    ((12,), 3, 2, 1, None),
]


@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis,offsets", CASES)
def test_initial_state_and_sweep_trajectory(oracle, shape, n_temps, n_rep, n_dis, offsets):
    nn = len(offsets) if offsets else len(shape)
    couplings = bimodal(shape, nn, n_dis, 5)
    temps = np.linspace(0.8, 2.4, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 1234, offsets)
    assert_state_equal(gpu, cpu)
    gpu.metropolis_sweep(7)
    cpu.sample(7, "metropolis", warmup_ratio=1.0)
    gpu.compute_energies()
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis,offsets", CASES)
def test_gibbs_sweep_trajectory(oracle, shape, n_temps, n_rep, n_dis, offsets):
    # heat-bath rule (sweep.rs:262-284): same stream, one draw per site, e >= (T/2) ln(u/(1-u))
    nn = len(offsets) if offsets else len(shape)
    couplings = bimodal(shape, nn, n_dis, 6)
    temps = np.linspace(0.6, 3.0, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 4321, offsets)
    gpu.gibbs_sweep(7)
    cpu.sample(7, "gibbs", warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)


def test_gibbs_with_zero_couplings_and_tempering(oracle):
    shape = (6, 6, 6)
    rng = np.random.default_rng(8)
    couplings = rng.integers(-1, 2, size=(2,) + shape + (3,)).astype(np.float32)  # diluted +-J
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.5, 2.5, 7), 4, 17)
    kwargs = dict(pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0.25)
    a = gpu.sample(20, "gibbs", **kwargs)
    b = cpu.sample(20, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # alternating rules on one handle keeps a single random stream per system
    a = gpu.sample(5, "metropolis", **kwargs)
    b = cpu.sample(5, "metropolis", **kwargs)
    assert_results_equal(a, b)
    a = gpu.sample(5, "gibbs", **kwargs)
    b = cpu.sample(5, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


def gaussian(shape, nn, n_dis, seed):
    rng = np.random.default_rng(seed)
    full = (n_dis,) + tuple(shape) + (nn,) if n_dis > 1 else tuple(shape) + (nn,)
    return rng.standard_normal(size=full).astype(np.float32)


@pytest.mark.parametrize("mode", ["metropolis", "gibbs"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis,offsets", CASES)
def test_general_couplings_sweep_trajectory(oracle, mode, shape, n_temps, n_rep, n_dis, offsets):
    """Gaussian couplings take the reference's logarithmic rule (sweep.rs:35-48,247-284): f32 local
    field in the reference's order, (T/2) ln(u) from the platform logf."""
    nn = len(offsets) if offsets else len(shape)
    couplings = gaussian(shape, nn, n_dis, 9)
    temps = np.linspace(0.4, 2.5, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 777, offsets)
    assert_state_equal(gpu, cpu)  # initial energies: sequential f32 sums (energy.rs:92-109)
    if mode == "gibbs":
        gpu.gibbs_sweep(6)
    else:
        gpu.metropolis_sweep(6)
    cpu.sample(6, mode, warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
def test_general_couplings_sample_matches_oracle(oracle, schedule):
    shape = (6, 6, 6)
    couplings = gaussian(shape, 3, 3, 21)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.5, 1.8, 12), 4, 5)
    kwargs = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.25)
    a = gpu.sample(20, "metropolis", **kwargs)
    b = cpu.sample(20, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    a = gpu.sample(7, "gibbs", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    b = cpu.sample(7, "gibbs", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # Houdayer needs no couplings beyond the energies recomputed after the move
    kwargs = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.0,
                  overlap_cluster_update_interval=2, overlap_cluster_build_mode="houdayer")
    a = gpu.sample(6, "metropolis", **kwargs)
    b = cpu.sample(6, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("inner", ["sw", "wolff"])
def test_cluster_moves_with_gaussian_couplings(oracle, inner):
    """Non-unit couplings: every bond has its own probability 1 - exp(-c |J| / T) (overlap.rs:430,
    601-603, 792-793; fk.rs:74,119), evaluated by the device expf (checked against the host's)."""
    shape = (6, 6, 6)
    couplings = gaussian(shape, 3, 2, 77)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.6, 2.0, 5), 4, 15)
    kwargs = dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode="jorg+cmr+houdayer",
                  overlap_cluster_mode=inner, cluster_update_interval=2, cluster_mode=inner, warmup_ratio=0.25)
    a = gpu.sample(12, "metropolis", **kwargs)
    b = cpu.sample(12, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # a custom lattice with five directions and mixed couplings
    offsets = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1]]
    couplings = gaussian((4, 4, 6), 5, 1, 78)
    couplings[np.random.default_rng(1).random(couplings.shape) < 0.2] = 0.0
    gpu, cpu = make_pair(oracle, (4, 4, 6), couplings, np.linspace(1.0, 4.0, 4), 2, 16, offsets)
    a = gpu.sample(10, "gibbs", **kwargs)
    b = cpu.sample(10, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


def test_unit_couplings_outside_the_table_domain_take_the_logarithmic_rule(oracle):
    # T = 0 and T < 0 make the reference decline the table (sweep.rs:112-116)
    shape = (6, 6)
    couplings = bimodal(shape, 2, 1, 3)
    gpu, cpu = make_pair(oracle, shape, couplings, [0.0, -1.0, 1.5], 2, 8)
    gpu.metropolis_sweep(5)
    cpu.sample(5, "metropolis", warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)
    gpu.gibbs_sweep(5)
    cpu.sample(5, "gibbs", warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)


BCC = [[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]]
FCC = [[1, 1, 0], [1, 0, 1], [0, 1, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]]
FIVE = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1]]


@pytest.mark.parametrize("offsets", [BCC, FCC, FIVE], ids=["bcc", "fcc", "five"])
@pytest.mark.parametrize("kind", ["bimodal", "gaussian", "diluted"])
def test_lattices_with_four_to_six_forward_directions(oracle, offsets, kind):
    """bcc / fcc geometries of the reference (python/peapods/spin_models.py:5-10) and any custom
    lattice with up to six offsets: sweeps (both rules), energies, overlap, tempering, Houdayer."""
    shape, nn = (4, 6, 4), len(offsets)
    rng = np.random.default_rng(7)
    full = (2,) + shape + (nn,)
    if kind == "bimodal":
        couplings = (2 * rng.integers(0, 2, size=full) - 1).astype(np.float32)
    elif kind == "gaussian":
        couplings = rng.standard_normal(size=full).astype(np.float32)
    else:
        couplings = rng.integers(-1, 2, size=full).astype(np.float32)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(1.0, 6.0, 5), 4, 11, offsets)
    assert_state_equal(gpu, cpu)
    gpu.metropolis_sweep(4)
    cpu.sample(4, "metropolis", warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)
    kwargs = dict(pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0.25)
    a = gpu.sample(12, "gibbs", **kwargs)
    b = cpu.sample(12, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    for inner in ("wolff", "sw"):
        kw = dict(pt_interval=2, overlap_cluster_update_interval=1, overlap_cluster_mode=inner, warmup_ratio=0.0)
        a = gpu.sample(6, "metropolis", **kw)
        b = cpu.sample(6, "metropolis", **kw)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("offsets", [BCC, FCC], ids=["bcc", "fcc"])
def test_cluster_moves_on_bcc_and_fcc(oracle, offsets):
    shape, nn = (4, 4, 6), len(offsets)
    rng = np.random.default_rng(9)
    couplings = (2 * rng.integers(0, 2, size=shape + (nn,)) - 1).astype(np.float32)
    for cluster_mode in ("sw", "wolff"):
        gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(2.0, 8.0, 4), 2, 13, offsets)
        kwargs = dict(cluster_update_interval=1, cluster_mode=cluster_mode, pt_interval=1, warmup_ratio=0.25,
                      overlap_cluster_update_interval=2, overlap_cluster_build_mode="jorg+cmr+houdayer",
                      overlap_cluster_mode=cluster_mode)
        a = gpu.sample(12, "metropolis", **kwargs)
        b = cpu.sample(12, "metropolis", **kwargs)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu)


def test_unit_lookup_case_with_zero_couplings(oracle):
    # the reference's own table test pattern (sweep.rs:382-442): couplings cycle -1, 0, +1
    shape = (8, 8)
    n = 64
    couplings = np.array([[-1.0, 0.0, 1.0][i % 3] for i in range(n * 2)], np.float32).reshape(8, 8, 2)
    gpu, cpu = make_pair(oracle, shape, couplings, [0.7, 2.0, 5.0], 2, 71)
    gpu.metropolis_sweep(20)
    cpu.sample(20, "metropolis", warmup_ratio=1.0)
    assert_state_equal(gpu, cpu, check_energy=False)
    gpu.compute_energies()
    cpu_e = cpu.get_energies()
    # oracle energies after sample() are those of the last PT/record sweep; recompute by a
    # recorded sweep-less call is not available, so compare through the energy definition
    spins = gpu.get_spins().reshape(gpu.n_systems, n).astype(np.float32)
    lat = oracle.Lattice(shape)
    J = couplings.reshape(n, 2)
    for s in range(gpu.n_systems):
        total = np.float32(0)
        for i in range(n):
            for d in range(2):
                total = np.float32(total + spins[s, i] * spins[s, lat.neighbor_fwd(i, d)] * J[i, d])
        assert gpu.get_energies()[s] == np.float32(total / np.float32(n))
    del cpu_e


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis", [((4, 4), 3, 2, 1), ((4, 4, 4), 6, 4, 3),
                                                       ((6, 6, 6), 24, 4, 2)])
def test_sample_matches_oracle(oracle, schedule, shape, n_temps, n_rep, n_dis):
    couplings = bimodal(shape, len(shape), n_dis, 11)
    temps = np.linspace(0.8, 1.6, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 99)
    kwargs = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.25)
    a = gpu.sample(24, "metropolis", **kwargs)
    b = cpu.sample(24, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # continuation semantics: a second call continues the chain and the PT counters
    a = gpu.sample(9, "metropolis", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    b = cpu.sample(9, "metropolis", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis", [((4, 4), 3, 2, 1), ((6, 6, 6), 24, 4, 2), ((8, 8, 8), 12, 4, 9)])
def test_heat_bath_sample_matches_oracle(oracle, schedule, shape, n_temps, n_rep, n_dis):
    """gibbs_sweep (sweep.rs:262-284) inside the full protocol: on +-J lattices the heat bath runs on the
    implicit-neighbour sweep kernel (three acceptance planes in LDS) with the overlap and the deferred measurement
    riding in its random-stream launch, like Metropolis."""
    couplings = bimodal(shape, len(shape), n_dis, 12)
    temps = np.linspace(0.8, 1.6, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 98)
    kwargs = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.25)
    a = gpu.sample(24, "gibbs", **kwargs)
    b = cpu.sample(24, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # a Metropolis call continues the same chain (the rules share generators, planes and tempering state)
    a = gpu.sample(9, "metropolis", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    b = cpu.sample(9, "metropolis", pt_interval=2, pt_schedule=schedule, warmup_ratio=0.0)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    a = gpu.sample(7, "gibbs", pt_interval=1, pt_schedule=schedule, warmup_ratio=0.0)
    b = cpu.sample(7, "gibbs", pt_interval=1, pt_schedule=schedule, warmup_ratio=0.0)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("mode", ["metropolis", "gibbs"])
def test_diluted_lattice_on_the_table_path(oracle, mode):
    """+-J with a fifth of the bonds removed, differently in every realisation: sites with an odd number of non-zero
    bonds take the second threshold set (sweep.rs:109-139), the sweep runs on the implicit-neighbour kernel with
    per-site non-zero masks, and with three tile words per realisation the two halves of a random-stream wave
    belong to different realisations (different parity words)."""
    shape, n_temps, n_rep, n_dis = (8, 8, 8), 24, 4, 3
    couplings = bimodal(shape, 3, n_dis, 21)
    couplings[np.random.default_rng(3).random(couplings.shape) < 0.2] = 0.0
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.8, 1.6, n_temps), n_rep, 97)
    kwargs = dict(pt_interval=1, warmup_ratio=0.25)
    a = gpu.sample(20, mode, **kwargs)
    b = cpu.sample(20, mode, **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("mode", ["metropolis", "gibbs"])
@pytest.mark.parametrize("lattice", ["triangular", "cubic-without-table"])
def test_record_path_protocol_matches_oracle(oracle, monkeypatch, lattice, mode):
    """Lattices whose neighbours do not follow from one wrap flag per slot (triangular: the diagonal wraps in three
    ways) keep the 16-byte site records on the sweep path; PEAPODS_NO_HC forces the same kernels on a cubic lattice.
    Full protocol: tempering every sweep, overlaps, a second call."""
    if lattice == "triangular":
        shape, offsets, nn = (6, 6), [[1, 0], [0, 1], [1, -1]], 3
    else:
        shape, offsets, nn = (6, 6, 6), None, 3
        monkeypatch.setenv("PEAPODS_DEVELOPER", "1")
        monkeypatch.setenv("PEAPODS_NO_HC", "1")
    couplings = bimodal(shape, nn, 3, 41)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.9, 2.2, 7), 4, 12, offsets)
    kwargs = dict(pt_interval=1, warmup_ratio=0.25)
    a = gpu.sample(30, mode, **kwargs)
    b = cpu.sample(30, mode, **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    a = gpu.sample(11, mode, pt_interval=2, pt_schedule="full_ladder", warmup_ratio=0.0)
    b = cpu.sample(11, mode, pt_interval=2, pt_schedule="full_ladder", warmup_ratio=0.0)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


def test_more_than_256_systems_per_realization(oracle):
    # 40 temperatures x 8 replicas = 320 systems (10 tile words): the tempering kernel moves the
    # generator cursors of every swapped slot, two slots per thread
    shape = (4, 4)
    couplings = bimodal(shape, 2, 2, 13)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.6, 3.0, 40), 8, 77)
    for schedule in ("full_ladder", "single_random_edge"):
        kwargs = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.2)
        a = gpu.sample(15, "metropolis", **kwargs)
        b = cpu.sample(15, "metropolis", **kwargs)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu)


def test_ferromagnet_without_replicas_or_pt(oracle):
    shape = (8, 8)
    couplings = np.ones(shape + (2,), np.float32)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(1.969, 2.569, 8), 1, 3)
    a = gpu.sample(40, "metropolis")
    b = cpu.sample(40, "metropolis")
    assert_results_equal(a, b)
    assert "overlap" not in a


def test_reset_replays_and_reseeds(oracle):
    shape = (4, 4)
    couplings = bimodal(shape, 2, 1, 3)
    gpu, cpu = make_pair(oracle, shape, couplings, [1.0, 2.0], 2, 41)
    initial = gpu.get_spins().copy()
    gpu.sample(2, "metropolis", warmup_ratio=0)
    gpu.reset()
    np.testing.assert_array_equal(gpu.get_spins(), initial)
    gpu.reset(seed=99)
    cpu.reset(seed=99)
    assert_state_equal(gpu, cpu)
    gpu.reset()
    np.testing.assert_array_equal(gpu.get_spins(), initial)


def test_overlap_dots_entry_point(oracle):
    shape = (4, 4, 4)
    couplings = bimodal(shape, 3, 2, 8)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.9, 1.5, 5), 4, 17)
    gpu.metropolis_sweep(3)
    dot_spin, dot_link = gpu.overlap_dots()
    n = 64
    lat = oracle.Lattice(shape)
    for r in range(2):
        spins = gpu.get_spins(r).reshape(gpu.n_systems, n).astype(np.int64)
        ids = gpu.get_system_ids(r)
        for p in range(2):
            for t in range(5):
                a = spins[int(ids[(2 * p) * 5 + t])]
                b = spins[int(ids[(2 * p + 1) * 5 + t])]
                q = a * b
                assert dot_spin[r, p, t] == q.sum()
                link = sum(int(q[i] * q[lat.neighbor_fwd(i, d)]) for i in range(n) for d in range(3))
                assert dot_link[r, p, t] == link


def test_invalid_options_fail_before_mutation():
    from peapods_amd import Ising

    model = Ising((4, 4), temperatures=np.array([2.0]), seed=13)
    before = model._sim.get_spins().copy()
    with pytest.raises(ValueError, match="requires cluster_mode='sw'"):
        model.sample(1, cluster_update_interval=1, cluster_mode="wolff", cluster_action="observe",
                     warmup_ratio=0)
    with pytest.raises(ValueError, match="must be 'ring' or 'fft'"):
        model.sample(4, autocorrelation_backend="other", warmup_ratio=0)
    with pytest.raises(ValueError, match="requires autocorrelation_max_lag"):
        model.sample(4, autocorrelation_backend="fft", warmup_ratio=0)
    with pytest.raises(ValueError, match="sweep_mode"):
        model.sample(4, sweep_mode="heatbath")
    np.testing.assert_array_equal(model._sim.get_spins(), before)


def test_ising_surface_seed_determinism_and_keys():
    # reference tests/test_sampling_interfaces.py:12-48,75-118
    from peapods_amd import Ising

    temps = np.array([1.0, 2.0, 4.0], dtype=np.float32)
    first = Ising((4, 4), couplings="bimodal", temperatures=temps, n_replicas=2, seed=41)
    second = Ising((4, 4), couplings="bimodal", temperatures=temps, n_replicas=2, seed=41)
    np.testing.assert_array_equal(first.couplings, second.couplings)
    np.testing.assert_array_equal(first._sim.get_spins(), second._sim.get_spins())
    one = Ising((4, 4), couplings="bimodal", n_disorder=1, seed=7)
    many = Ising((4, 4), couplings="bimodal", n_disorder=3, seed=7)
    np.testing.assert_array_equal(one.couplings, many.couplings[0])

    result = first.sample(2, pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0)
    pt = result["per_disorder"]["parallel_tempering"]
    assert pt["edge_attempts"].shape == (1, 2) and np.all(pt["edge_attempts"] == 4)
    assert pt["edge_attempts"].dtype == np.uint64
    assert pt["round_trips"].shape == (1, 2, 3)
    for key in ("mags", "mags2", "mags4", "energies", "energies2", "overlap", "overlap2",
                "overlap4", "link_overlap", "link_overlap2", "link_overlap4"):
        assert result[key].shape == (3,) and result[key].dtype == np.float64
    assert len(result["overlap_histogram"]) == 3 and result["overlap_histogram"][0].shape == (17,)
    assert result["ql_at_q_sum"].shape == (3, 17)
    assert first.sg_binder.shape == (3,) and first.heat_capacity.shape == (3,)
    continued = first.sample(1, pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0)
    assert np.all(continued["per_disorder"]["parallel_tempering"]["edge_attempts"] == 6)
    first.reset()
    again = first.sample(1, pt_interval=1, pt_schedule="full_ladder", warmup_ratio=0)
    assert np.all(again["per_disorder"]["parallel_tempering"]["edge_attempts"] == 2)
    assert many.sample(4)["mags"].shape == (32,)


@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis,interval", [((4, 4), 3, 2, 1, 1), ((6, 6, 6), 5, 4, 2, 1),
                                                                ((12, 12, 12), 6, 4, 2, 2),
                                                                ((8, 8), 4, 3, 2, 1)])
def test_houdayer_overlap_cluster_move_matches_oracle(oracle, shape, n_temps, n_rep, n_dis, interval):
    # BASELINE configs[4]: +-J glass with the Houdayer move (default 'wolff' inner mode)
    couplings = bimodal(shape, len(shape), n_dis, 21)
    temps = np.linspace(0.7, 1.6, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 7)
    kwargs = dict(pt_interval=1, overlap_cluster_update_interval=interval,
                  overlap_cluster_build_mode="houdayer", overlap_cluster_mode="wolff",
                  warmup_ratio=0.25)
    a = gpu.sample(12, "metropolis", **kwargs)
    b = cpu.sample(12, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # without tempering, and continuing the same chains
    kwargs.pop("pt_interval")
    a = gpu.sample(5, "metropolis", **kwargs)
    b = cpu.sample(5, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("mode", ["jorg", "houdayer+jorg"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis", [((4, 4), 3, 2, 1), ((6, 6, 6), 5, 4, 2),
                                                       ((12, 12, 12), 6, 4, 2), ((2, 4, 6), 3, 2, 1)])
def test_jorg_overlap_cluster_move_matches_oracle(oracle, mode, shape, n_temps, n_rep, n_dis):
    # Joerg's stochastic bonds consume the pair generator in depth-first order; the GPU replays it
    couplings = bimodal(shape, len(shape), n_dis, 33)
    temps = np.linspace(0.7, 1.6, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 9)
    kwargs = dict(pt_interval=2, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                  overlap_cluster_mode="wolff", warmup_ratio=0.25)
    a = gpu.sample(10, "metropolis", **kwargs)
    b = cpu.sample(10, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("mode", ["houdayer", "jorg", "houdayer+jorg"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis", [((4, 4), 3, 2, 1), ((6, 6, 6), 5, 4, 2),
                                                       ((12, 12, 12), 6, 4, 2), ((2, 4, 6), 3, 2, 1),
                                                       ((5, 7), 4, 6, 1)])
def test_sw_variant_of_the_overlap_moves_matches_oracle(oracle, mode, shape, n_temps, n_rep, n_dis):
    """cluster_mode='sw' (overlap.rs:228-318,417-520): union-find over every forward bond in
    (site, dim) order, one draw per eligible Joerg bond, one coin per multi-site cluster in the
    order of its smallest site.  The GPU builds the same partition with parallel components."""
    couplings = bimodal(shape, len(shape), n_dis, 35)
    temps = np.linspace(0.7, 1.9, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 19)
    kwargs = dict(pt_interval=2, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                  overlap_cluster_mode="sw", warmup_ratio=0.25)
    a = gpu.sample(10, "metropolis", **kwargs)
    b = cpu.sample(10, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)
    # wolff and sw calls share the pair generators
    kwargs["overlap_cluster_mode"] = "wolff"
    a = gpu.sample(4, "metropolis", **kwargs)
    b = cpu.sample(4, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("shape,n_temps,n_rep", [((16, 16, 16), 3, 2), ((7, 9, 11), 3, 4), ((3, 5, 40), 3, 2),
                                                 ((70,), 3, 2), ((33, 3), 4, 2), ((1, 6, 2), 3, 2)])
def test_word_parallel_cluster_phases_on_awkward_extents(oracle, shape, n_temps, n_rep):
    """The union-find kernel works on 32-site words of its bitmaps wherever the lattice is hypercubic (k_cluster_sw:
    neighbour = shifted bitmap, unions from packed bond words, coins deposited per word).  Extents that are not
    multiples of 32 put rows across word boundaries; 16^3 has more (direction, word) cells than threads; a row of 40
    has runs longer than a word; extents of 1 and 2 make self-bonds and double bonds.  All three moves, same state and
    results as the reference's site-by-site loops (overlap.rs:228-318,417-520,555-971)."""
    couplings = bimodal(shape, len(shape), 2, 77)
    temps = np.linspace(0.6, 2.2, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 23)
    for mode in ("houdayer", "jorg", "cmr"):
        kwargs = dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                      overlap_cluster_mode="sw", warmup_ratio=0.0)
        a = gpu.sample(5, "metropolis", **kwargs)
        b = cpu.sample(5, "metropolis", **kwargs)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("shape,offsets", [((40,), [[1], [2]]), ((6, 10), [[1, 0], [0, 1], [0, 2]]),
                                           ((9, 7), [[0, 1], [2, 0]]), ((5, 6, 7), [[0, 0, 1], [0, 1, 0], [0, 0, 3]])])
def test_word_parallel_cluster_phases_with_longer_axis_offsets(oracle, shape, offsets):
    """Offsets of more than one site along an axis still give two offsets per direction (implicit neighbours), but the
    wrap flag of such a direction changes along a run of next-site bonds: the square rule that drops redundant unions
    (k_cluster_sw, union_word) and the shifted-bitmap neighbours must hold there too.  Pair moves and Swendsen-Wang."""
    nn = len(offsets)
    rng = np.random.default_rng(11)
    couplings = (2 * rng.integers(0, 2, size=(2,) + tuple(shape) + (nn,)) - 1).astype(np.float32)
    temps = np.linspace(0.7, 2.1, 3)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, 4, 29, offsets)
    for mode in ("houdayer", "jorg", "cmr"):
        kwargs = dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                      overlap_cluster_mode="sw", warmup_ratio=0.0)
        a = gpu.sample(5, "metropolis", **kwargs)
        b = cpu.sample(5, "metropolis", **kwargs)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu, check_energy=False)
    gpu, cpu = make_pair(oracle, shape, np.abs(couplings), temps * 2, 1, 3, offsets)
    kwargs = dict(cluster_update_interval=1, cluster_mode="sw", pt_interval=1, warmup_ratio=0.0)
    a = gpu.sample(6, "metropolis", **kwargs)
    b = cpu.sample(6, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("shape,temp", [((8, 8, 8), 1.2), ((12, 12, 12), 2.5), ((16, 16, 16), 4.4), ((30, 30), 1.5)])
def test_swendsen_wang_with_more_bonds_than_the_union_list_holds(oracle, shape, temp):
    """A cold ferromagnet accepts nearly every bond: more bonds than the N entries of the packed union list, so the
    kernel falls back to one pass per direction (k_cluster_sw, union_pass).  fk.rs:113-192."""
    couplings = np.ones(tuple(shape) + (len(shape),), np.float32)
    temps = np.array([temp, temp * 1.5])
    gpu, cpu = make_pair(oracle, shape, couplings, temps, 1, 5)
    kwargs = dict(cluster_update_interval=1, cluster_mode="sw", pt_interval=1, warmup_ratio=0.0)
    a = gpu.sample(6, "metropolis", **kwargs)
    b = cpu.sample(6, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("inner", ["sw", "wolff"])
@pytest.mark.parametrize("mode", ["cmr", "houdayer+cmr+jorg"])
@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis", [((4, 4), 3, 2, 1), ((6, 6, 6), 5, 4, 2),
                                                       ((12, 12, 12), 4, 2, 1), ((2, 4, 6), 3, 2, 1)])
def test_cmr_overlap_cluster_move_matches_oracle(oracle, inner, mode, shape, n_temps, n_rep, n_dis):
    """CMR (overlap.rs:555-971): blue clusters over doubly satisfied bonds (p = 1 - r^2), joint
    flips, then grey clusters = blue + singly satisfied red bonds (p = 1 - r) with a 2-bit draw
    deciding which system flips; sw = union-find path, wolff = two-phase depth-first search."""
    couplings = bimodal(shape, len(shape), n_dis, 41)
    temps = np.linspace(0.7, 1.9, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 29)
    kwargs = dict(pt_interval=2, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                  overlap_cluster_mode=inner, warmup_ratio=0.25)
    a = gpu.sample(10, "metropolis", **kwargs)
    b = cpu.sample(10, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)


@pytest.mark.parametrize("inner", ["sw", "wolff"])
@pytest.mark.parametrize("mode,n_rep", [("houd4", 4), ("houd4", 8), ("houd6", 6), ("cmr+houd4", 4),
                                        ("houdayer+houd4+jorg", 5)])
def test_houdayer_groups_of_n_replicas_match_oracle(oracle, inner, mode, n_rep):
    """"houdN" (overlap.rs:153-339 with group_size N; experimental upstream but part of its test
    protocols, tests/spin_glass_crossings.py:139-176): N replicas whose spins sum to zero on the
    active sites flip together."""
    shape = (6, 6, 6)
    couplings = bimodal(shape, 3, 2, 51)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.8, 1.8, 4), n_rep, 61)
    kwargs = dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
                  overlap_cluster_mode=inner, warmup_ratio=0.25)
    a = gpu.sample(12, "metropolis", **kwargs)
    b = cpu.sample(12, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


def test_cmr_with_diluted_couplings(oracle):
    shape = (8, 8, 8)
    rng = np.random.default_rng(4)
    couplings = rng.integers(-1, 2, size=shape + (3,)).astype(np.float32)
    for inner in ("sw", "wolff"):
        gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.6, 2.2, 6), 4, 37)
        kwargs = dict(pt_interval=1, overlap_cluster_update_interval=2, overlap_cluster_build_mode="cmr",
                      overlap_cluster_mode=inner, warmup_ratio=0.0)
        a = gpu.sample(12, "metropolis", **kwargs)
        b = cpu.sample(12, "metropolis", **kwargs)
        assert_results_equal(a, b)
        assert_state_equal(gpu, cpu)


def test_sw_variant_with_diluted_couplings(oracle):
    shape = (8, 8, 8)
    rng = np.random.default_rng(3)
    couplings = rng.integers(-1, 2, size=shape + (3,)).astype(np.float32)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.6, 2.2, 6), 4, 23)
    kwargs = dict(pt_interval=1, overlap_cluster_update_interval=2, overlap_cluster_build_mode="jorg",
                  overlap_cluster_mode="sw", warmup_ratio=0.0)
    a = gpu.sample(12, "metropolis", **kwargs)
    b = cpu.sample(12, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("cluster_mode", ["sw", "wolff"])
@pytest.mark.parametrize("shape,kind,n_temps,n_rep,n_dis,offsets", [
    ((8, 8), "ferro", 4, 1, 1, None),
    ((6, 6, 6), "ferro", 5, 2, 1, None),
    ((6, 6, 6), "bimodal", 4, 2, 2, None),
    ((5, 7), "diluted", 3, 2, 1, None),
    ((6, 6), "ferro", 3, 2, 1, [[1, 0], [0, 1], [1, -1]]),
    ((2, 4, 6), "bimodal", 3, 2, 1, None),
])
def test_fk_cluster_update_matches_oracle(oracle, cluster_mode, shape, kind, n_temps, n_rep, n_dis, offsets):
    """cluster_update_interval (clusters/fk.rs): Swendsen-Wang = union-find over every forward
    bond in (site, dim) order, one draw per satisfied bond, one coin per cluster in the order of
    its smallest site; Wolff = seed + depth-first search with bonds drawn as they are met.  Both use
    the system's own generator between the sweep draws and the tempering draws."""
    nn = len(offsets) if offsets else len(shape)
    full = ((n_dis,) if n_dis > 1 else ()) + tuple(shape) + (nn,)
    rng = np.random.default_rng(2)
    if kind == "ferro":
        couplings = np.ones(full, np.float32)
        temps = np.linspace(1.8, 3.2, n_temps) if len(shape) == 2 else np.linspace(3.8, 5.2, n_temps)
    elif kind == "bimodal":
        couplings = (2 * rng.integers(0, 2, size=full) - 1).astype(np.float32)
        temps = np.linspace(0.8, 2.0, n_temps)
    else:
        couplings = rng.integers(-1, 2, size=full).astype(np.float32)
        temps = np.linspace(0.8, 2.0, n_temps)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, 31, offsets)
    kwargs = dict(cluster_update_interval=1, cluster_mode=cluster_mode, pt_interval=1, warmup_ratio=0.25)
    a = gpu.sample(12, "metropolis", **kwargs)
    b = cpu.sample(12, "metropolis", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    kwargs = dict(cluster_update_interval=3, cluster_mode=cluster_mode, warmup_ratio=0.0)
    a = gpu.sample(7, "gibbs", **kwargs)
    b = cpu.sample(7, "gibbs", **kwargs)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


@pytest.mark.parametrize("kind", ["bimodal", "gaussian"])
def test_equilibration_diagnostic_matches_oracle(oracle, kind):
    """equilibration_diagnostic=True (equilibration.rs, mod.rs:511-541): energies and the link
    overlap of every sweep, warm-up included, averaged up to the checkpoints 128, 256, ..., n."""
    shape = (4, 4, 4)
    couplings = bimodal(shape, 3, 2, 5) if kind == "bimodal" else gaussian(shape, 3, 2, 5)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.9, 2.1, 5), 4, 3)
    kwargs = dict(pt_interval=1, overlap_cluster_update_interval=2, warmup_ratio=0.25,
                  equilibration_diagnostic=True)
    a = gpu.sample(300, "metropolis", **kwargs)
    b = cpu.sample(300, "metropolis", **kwargs)
    np.testing.assert_array_equal(a["equil_sweeps"], [128, 256, 300])
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)
    # without replicas the link overlap column is zero (mod.rs:537-540)
    gpu1, cpu1 = make_pair(oracle, shape, couplings, [1.5, 2.5], 1, 3)
    a = gpu1.sample(130, "gibbs", equilibration_diagnostic=True)
    b = cpu1.sample(130, "gibbs", equilibration_diagnostic=True)
    assert_results_equal(a, b)
    assert not a["equil_link_overlap_avg"].any()


@pytest.mark.parametrize("n_rep", [1, 4])
def test_autocorrelation_times_match_oracle(oracle, n_rep):
    """autocorrelation_max_lag (autocorrelation.rs ring backend + Sokal window, mod.rs:342-371,
    580-595): per-sweep <m^2> and <q^2> series in f32, products in f64, tau_int per temperature."""
    shape = (8, 8)
    couplings = np.ones((2,) + shape + (2,), np.float32)
    gpu, cpu = make_pair(oracle, shape, couplings, [2.0, 2.269, 2.6], n_rep, 6)
    kwargs = dict(cluster_update_interval=2, cluster_mode="sw", pt_interval=1, warmup_ratio=0.25,
                  autocorrelation_max_lag=150)
    a = gpu.sample(1200, "metropolis", **kwargs)
    b = cpu.sample(1200, "metropolis", **kwargs)
    assert ("overlap2_tau" in a) == (n_rep >= 2)
    assert_results_equal(a, b)
    assert np.all(a["mags2_tau"] > 0.4)
    # the lag is capped at a quarter of the measurement sweeps (mod.rs:343-345)
    a = gpu.sample(40, "gibbs", warmup_ratio=0.5, autocorrelation_max_lag=1000)
    b = cpu.sample(40, "gibbs", warmup_ratio=0.5, autocorrelation_max_lag=1000)
    assert_results_equal(a, b)


@pytest.mark.parametrize("mode", ["metropolis", "gibbs"])
def test_autocorrelation_series_on_the_relayed_protocol(oracle, mode):
    """With the overlap measured inside the next sweep's launch, a launch files the m^2 sample of one recorded sweep
    and the q^2 sample of the sweep before it (mod.rs:580-595 pushes both for the same sweep): the series, and so
    the integrated times, must not notice -- tempering every sweep, warm-up sweeps in front, a second call after."""
    shape = (6, 6, 6)
    couplings = bimodal(shape, 3, 2, 31)
    gpu, cpu = make_pair(oracle, shape, couplings, np.linspace(0.9, 1.8, 8), 4, 8)
    kwargs = dict(pt_interval=1, warmup_ratio=0.25, autocorrelation_max_lag=20)
    a = gpu.sample(160, mode, **kwargs)
    b = cpu.sample(160, mode, **kwargs)
    assert "overlap2_tau" in a and "mags2_tau" in a
    assert_results_equal(a, b)
    a = gpu.sample(90, mode, pt_interval=3, warmup_ratio=0.0, autocorrelation_max_lag=10)
    b = cpu.sample(90, mode, pt_interval=3, warmup_ratio=0.0, autocorrelation_max_lag=10)
    assert_results_equal(a, b)
    assert_state_equal(gpu, cpu)


def test_fft_autocorrelation_backend_agrees_with_the_ring(oracle):
    # reference tests/test_sampling_interfaces.py:159-194 and autocorrelation.rs:345-372: the two backends agree
    # to 1e-9; the ring result is the exact one (and is compared bit for bit with the oracle elsewhere)
    from peapods_amd import Ising

    kw = dict(lattice_shape=(4, 4), couplings="bimodal", temperatures=np.array([1.0, 2.0], dtype=np.float32),
              n_replicas=2, seed=37)
    default = Ising(**kw).sample(64, autocorrelation_max_lag=8, warmup_ratio=0)
    ring = Ising(**kw).sample(64, autocorrelation_max_lag=8, autocorrelation_backend="ring", warmup_ratio=0)
    fft = Ising(**kw).sample(64, autocorrelation_max_lag=8, autocorrelation_backend="fft", warmup_ratio=0)
    np.testing.assert_array_equal(default["mags2_tau"], ring["mags2_tau"])
    np.testing.assert_allclose(fft["mags2_tau"], default["mags2_tau"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fft["overlap2_tau"], default["overlap2_tau"], rtol=0, atol=1e-9)
    for key in default:  # everything else is untouched by the choice of backend
        if not key.endswith("_tau") and key != "per_disorder" and key != "overlap_histogram":
            np.testing.assert_array_equal(fft[key], default[key], err_msg=key)
    # a longer series on a larger lattice, several realisations
    big = dict(lattice_shape=(6, 6, 6), couplings="bimodal", temperatures=np.linspace(0.9, 2.0, 5).astype(np.float32),
               n_replicas=2, n_disorder=3, seed=5)
    a = Ising(**big).sample(700, autocorrelation_max_lag=60, warmup_ratio=0.25, pt_interval=1)
    b = Ising(**big).sample(700, autocorrelation_max_lag=60, autocorrelation_backend="fft", warmup_ratio=0.25, pt_interval=1)
    np.testing.assert_allclose(b["mags2_tau"], a["mags2_tau"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(b["overlap2_tau"], a["overlap2_tau"], rtol=0, atol=1e-9)


def test_unbuilt_overlap_modes_are_refused_not_emulated():
    from peapods_amd import IsingSimulation

    couplings = bimodal((4, 4), 2, 1, 2)
    sim = IsingSimulation([4, 4], couplings, np.array([1.0, 2.0], np.float32), 2, None, 1)
    before = sim.get_spins().copy()
    with pytest.raises(ValueError, match="requires autocorrelation_max_lag"):
        sim.sample(8, "metropolis", autocorrelation_backend="fft")
    with pytest.raises(ValueError, match="n_replicas >= max group_size"):
        sim.sample(2, "metropolis", overlap_cluster_update_interval=1, overlap_cluster_build_mode="houd4")
    np.testing.assert_array_equal(sim.get_spins(), before)


# ---- state setters: the inner boundary in the other direction (sweep.rs:220-229 borrows the caller's slices) ----
@pytest.mark.parametrize("offsets", [None, [[1, 0], [0, 1]]])
def test_unit_lookup_with_permuted_systems_on_the_device(oracle, offsets):
    """The reference's own permuted-systems test (mcmc/sweep.rs:382-442) on the HIP path: 8 x 8, couplings cycling
    -1, 0, +1, temperatures 0.7 / 2.0 / 5.0, system_ids = [2, 0, 1], spins -1 at every fifth index, generators seeded
    71, 72, 73, twenty Metropolis sweeps -- state handed over with pa_sim_set_spins / set_rng_state / set_system_ids,
    compared with the oracle's slice-level sweep (table path and logarithm path agree there, test_oracle_kat.py)."""
    import ctypes as C

    from peapods_amd import IsingSimulation
    from peapods_amd import _native as nat

    lib = oracle.load()
    lat = oracle.Lattice([8, 8], offsets=offsets)
    n = lat.n_spins
    couplings = np.array([[-1.0, 0.0, 1.0][i % 3] for i in range(n * lat.n_neighbors)], np.float32)
    temps = np.array([0.7, 2.0, 5.0], dtype=np.float32)
    system_ids = np.array([2, 0, 1], dtype=np.uint64)
    initial = np.array([-1 if i % 5 == 0 else 1 for i in range(3 * n)], dtype=np.int8)

    gpu = IsingSimulation([8, 8], couplings.reshape(8, 8, 2), temps, 1, offsets, 5)
    gpu.set_spins(initial)
    for system, seed in enumerate((71, 72, 73)):
        state = np.zeros(4, dtype=np.uint64)
        nat.lib().pa_xoshiro_state(seed, 0, nat.ptr(state))
        gpu.set_rng_state(0, system, state)
    gpu.set_system_ids(system_ids)
    np.testing.assert_array_equal(gpu.get_spins(0), initial)            # spins stay with their systems
    np.testing.assert_array_equal(gpu.get_system_ids(0), system_ids)

    want_spins = initial.copy()
    rngs = oracle.make_rngs([71, 72, 73])
    lookup = lib.orc_lookup_new(couplings.ctypes.data_as(C.c_void_p), len(couplings), temps.ctypes.data_as(C.c_void_p), 3,
                                lat.n_neighbors)
    energies = np.zeros(3, dtype=np.float32)
    lib.orc_compute_energies(lat.handle, want_spins.ctypes.data, couplings.ctypes.data, 3, energies.ctypes.data, None, None)
    np.testing.assert_array_equal(gpu.get_energies(0), energies)         # recomputed by set_spins
    for _ in range(20):
        lib.orc_metropolis_sweep(lat.handle, want_spins.ctypes.data_as(C.c_void_p), couplings.ctypes.data_as(C.c_void_p),
                                 temps.ctypes.data_as(C.c_void_p), system_ids.ctypes.data_as(C.c_void_p), 3,
                                 C.byref(rngs), lookup)
    lib.orc_lookup_free(lookup)
    gpu.metropolis_sweep(20)
    np.testing.assert_array_equal(gpu.get_spins(0), want_spins)
    assert not np.array_equal(want_spins, initial)
    for system in range(3):
        got = gpu.get_rng_state(0, system)
        assert [int(x) for x in got] == [int(x) for x in rngs[system].s]


def test_setters_validate_before_touching_anything():
    from peapods_amd import IsingSimulation

    rng = np.random.default_rng(3)
    couplings = (2 * rng.integers(0, 2, size=(4, 4, 4, 3)) - 1).astype(np.float32)
    gpu = IsingSimulation([4, 4, 4], couplings, np.linspace(1.0, 2.0, 3).astype(np.float32), 2, None, 9)
    before = gpu.get_spins(0).copy()
    ids_before = gpu.get_system_ids(0).copy()
    bad = before.copy()
    bad[7] = 0
    with pytest.raises(ValueError, match="spins must be"):
        gpu.set_spins(bad)
    with pytest.raises(ValueError, match="permutation"):
        gpu.set_system_ids([0, 0, 2, 3, 4, 5])
    with pytest.raises(ValueError, match="each replica"):
        gpu.set_system_ids([3, 1, 2, 0, 4, 5])   # systems of different replica ladders exchanged
    with pytest.raises(ValueError):
        gpu.set_rng_state(0, 6, [1, 2, 3, 4])
    np.testing.assert_array_equal(gpu.get_spins(0), before)
    np.testing.assert_array_equal(gpu.get_system_ids(0), ids_before)


def test_set_system_ids_then_sample_matches_an_oracle_that_reached_the_same_state(oracle):
    """A permutation the tempering dynamics themselves produce: run both engines, read the GPU state back, rebuild it
    on a FRESH handle through the setters, continue both -- the continued trajectories must agree."""
    shape, temps = (4, 4, 4), np.linspace(0.9, 2.5, 5).astype(np.float32)
    couplings = bimodal(shape, 3, 1, 21)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, 2, 77)
    kw = dict(pt_interval=1, warmup_ratio=0.0)
    gpu.sample(30, "metropolis", **kw)
    cpu.sample(30, "metropolis", **kw)
    ids = gpu.get_system_ids(0)
    assert not np.array_equal(ids, np.arange(10))   # the ladder has moved
    from peapods_amd import IsingSimulation

    fresh = IsingSimulation(list(shape), couplings, temps, 2, None, 123456)  # another seed: everything is overwritten
    fresh.set_spins(gpu.get_spins(0))
    for s in range(10):
        fresh.set_rng_state(0, s, gpu.get_rng_state(0, s))
    fresh.set_system_ids(ids)
    assert_state_equal(fresh, cpu)
    fresh.metropolis_sweep(7)
    gpu.metropolis_sweep(7)
    np.testing.assert_array_equal(fresh.get_spins(0), gpu.get_spins(0))
    for s in range(10):
        np.testing.assert_array_equal(fresh.get_rng_state(0, s), gpu.get_rng_state(0, s))


def test_cached_sums_follow_the_systems_after_a_relayed_event(oracle):
    """pt_interval > 1 with the run ending inside the warm-up: the sweep after a relayed tempering event does not
    compute energies by the protocol, and the cached sums would stay in the slot order from before the swap (advisor
    finding, round 2).  The getters must return what the reference's caches hold: every system's own sums as of the
    last sweep that computed them (the one the event ran on) -- not another system's, and not recomputed ones."""
    shape, temps = (6, 6, 6), np.linspace(0.9, 2.5, 6).astype(np.float32)
    couplings = bimodal(shape, 3, 1, 5)
    gpu, cpu = make_pair(oracle, shape, couplings, temps, 2, 3)
    kw = dict(pt_interval=3, warmup_ratio=1.0)
    gpu.sample(23, "metropolis", **kw)   # events at sweeps 0, 3, ..., 21; sweep 22 follows the last one
    cpu.sample(23, "metropolis", **kw)
    np.testing.assert_array_equal(gpu.get_system_ids(0), cpu.get_system_ids(0))
    np.testing.assert_array_equal(gpu.get_spins(0), cpu.get_spins(0))
    np.testing.assert_array_equal(gpu.get_energies(0), cpu.get_energies(0))
    # ... which are the sums of the state sweep 21 left, system by system (the event moves temperatures, not spins)
    _, before = make_pair(oracle, shape, couplings, temps, 2, 3)
    before.sample(22, "metropolis", **kw)
    lib = oracle.load()
    lat = oracle.Lattice(list(shape))
    n_sys = 12
    energies = np.zeros(n_sys, dtype=np.float32)
    mags = np.zeros(n_sys, dtype=np.int64)
    J = np.ascontiguousarray(couplings.reshape(-1))
    spins = before.get_spins(0)
    lib.orc_compute_energies(lat.handle, spins.ctypes.data, J.ctypes.data, n_sys, energies.ctypes.data, mags.ctypes.data, None)
    np.testing.assert_array_equal(gpu.get_magnetization_sums(0), mags)
    np.testing.assert_array_equal(gpu.get_energies(0), energies)
    _, earlier = make_pair(oracle, shape, couplings, temps, 2, 3)
    earlier.sample(21, "metropolis", **kw)  # (sweeps 0..20: without the event of sweep 21)
    assert not np.array_equal(gpu.get_system_ids(0), earlier.get_system_ids(0))  # that event did exchange systems


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
def test_register_resident_sweep_matches_oracle(oracle, monkeypatch, schedule):
    """k_sweep_reg (kernels_reg.hip, opt-in): a 16^3 tile word's 4096 sites in the registers of ONE wave -- level mod 16 as
    the register index, y along a DPP row, z across registers and rows -- with the acceptance planes filed in that order
    by the random-stream kernel (SimDims::planes_perm).  Same trajectory as k_sweep_hc: 40 slots in two words (a
    tempering pair straddles them), both schedules, energies and overlaps from the swept registers."""
    from peapods_amd import IsingSimulation, _native as nat

    monkeypatch.setenv("PEAPODS_DEVELOPER", "1")
    monkeypatch.setenv("PEAPODS_REG", "1")
    shape = (16, 16, 16)
    rng = np.random.default_rng(16)
    couplings = (2 * rng.integers(0, 2, size=(3,) + shape + (3,)) - 1).astype(np.float32)
    temps = np.linspace(0.8, 1.6, 10).astype(np.float32)
    gpu = IsingSimulation(list(shape), couplings, temps, 4, None, 77)
    assert "PEAPODS_REG=1" in nat.lib().pa_active_switches().decode()
    cpu = oracle.OracleSimulation(list(shape), couplings, temps, 4, None, 77)
    kw = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.25)
    a = gpu.sample(12, "metropolis", **kw)
    b = cpu.sample(12, "metropolis", n_threads=3, **kw)
    for key in ("mags", "mags2", "mags4", "energies", "energies2", "overlap", "overlap2", "overlap4", "link_overlap",
                "link_overlap2", "ql_at_q_sum", "per_sample_overlap_histogram"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    for r in range(3):
        np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))
        np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r))
        np.testing.assert_array_equal(gpu.get_energies(r), cpu.get_energies(r))
