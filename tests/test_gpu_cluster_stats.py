"""Cluster statistics, graph observations and snapshots on the GPU against the oracle (reference
spin-sim/src/simulation/mod.rs:17-140,217-335,442-483,604-742; clusters/utils.rs:295-455).

Every histogram, top-four mean, observation summary, snapshot and the final state (spins, generator
states) is compared bit for bit.  Calls that collect anything take the union-find form of the
single-cluster moves, exactly when the reference does (recorded sweeps only)."""

import numpy as np
import pytest

from test_gpu_parity import assert_state_equal, bimodal, make_pair

pytestmark = pytest.mark.gpu


def assert_same(a, b, path=""):
    if isinstance(a, dict):
        assert set(a) == set(b), f"{path}: {sorted(a)} != {sorted(b)}"
        for key in a:
            assert_same(a[key], b[key], f"{path}/{key}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for k, (x, y) in enumerate(zip(a, b)):
            assert_same(x, y, f"{path}[{k}]")
    elif isinstance(a, np.ndarray):
        np.testing.assert_array_equal(a, b, err_msg=path)
    else:
        assert a == b, path


def run_both(oracle, shape, couplings, temps, n_rep, n_sweeps, seed=5, offsets=None, **kw):
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, seed, offsets)
    a = gpu.sample(n_sweeps, "metropolis", **kw)
    b = cpu.sample(n_sweeps, "metropolis", **kw)
    assert_same(a, b)
    assert_state_equal(gpu, cpu, check_energy=False)
    return a


def ferro(shape, nn):
    return np.ones(tuple(shape) + (nn,), dtype=np.float32)


@pytest.mark.parametrize("mode", ["sw", "wolff"])
@pytest.mark.parametrize("shape,kind,n_rep,n_dis", [((8, 8), "ferro", 1, 1), ((6, 6, 6), "bimodal", 2, 3),
                                                    ((5, 7), "ferro", 2, 1)])
def test_fk_cluster_size_histogram(oracle, mode, shape, kind, n_rep, n_dis):
    nn = len(shape)
    J = ferro(shape, nn) if kind == "ferro" else bimodal(shape, nn, n_dis, 3)
    temps = np.linspace(1.6, 3.2, 4)
    out = run_both(oracle, shape, J, temps, n_rep, 12, cluster_update_interval=1, cluster_mode=mode,
                   collect_cluster_stats=True, warmup_ratio=0.5, pt_interval=2)
    assert "fk_csd" in out and len(out["fk_csd"]) == 4
    n = int(np.prod(shape))
    n_real = J.shape[0] if J.ndim == nn + 2 else 1
    for hist in out["fk_csd"]:  # every recorded call partitions every system's lattice
        assert int((np.arange(n + 1) * hist).sum()) == n * 6 * n_rep * n_real


@pytest.mark.parametrize("shape,n_rep,n_dis", [((8, 8), 1, 2), ((4, 4, 4), 2, 1)])
def test_fk_observation_leaves_the_spins_alone(oracle, shape, n_rep, n_dis):
    nn = len(shape)
    J = np.stack([ferro(shape, nn)] * n_dis) if n_dis > 1 else ferro(shape, nn)
    temps = np.array([1.8, 2.269, 2.8], dtype=np.float32)
    out = run_both(oracle, shape, J, temps, n_rep, 10, cluster_update_interval=2, cluster_mode="sw",
                   cluster_action="observe", warmup_ratio=0.2)
    obs = out["per_disorder"]["cluster_observations"]["fk"]
    assert obs["observation_count"].shape == (n_dis, 3) and np.all(obs["observation_count"] == 4 * n_rep)
    assert ("winding_x" in obs) == (nn == 2)
    # the observation draws the bonds but moves nothing: same spins as a run without it would differ only
    # through the generator, so compare with the oracle (done in run_both) and check the densities' range
    assert np.all(obs["active_bond_density"] >= 0) and np.all(obs["active_bond_density"] <= 1)


OVERLAP_CASES = [
    # build mode, inner mode, shape, n_temps, n_replicas, n_disorder
    ("houdayer", "sw", (6, 6, 6), 3, 2, 2),
    ("houdayer", "wolff", (6, 6, 6), 3, 4, 1),
    ("jorg", "sw", (4, 4, 4), 4, 2, 2),
    ("jorg", "wolff", (6, 6, 6), 3, 4, 2),
    ("cmr", "sw", (4, 4, 4), 3, 2, 1),
    ("cmr", "wolff", (6, 6, 6), 4, 4, 2),
    ("cmr+houdayer", "sw", (4, 4, 4), 3, 4, 1),
    ("jorg+cmr+houdayer", "wolff", (4, 4, 4), 3, 2, 2),
    ("houd4", "sw", (4, 4, 4), 3, 4, 1),
    ("houd4", "wolff", (4, 4, 4), 3, 8, 1),
    ("houdayer", "wolff", (8, 8), 3, 2, 1),
    ("jorg", "sw", (6, 5), 3, 2, 1),
]


@pytest.mark.parametrize("build,inner,shape,n_temps,n_rep,n_dis", OVERLAP_CASES)
def test_overlap_cluster_statistics(oracle, build, inner, shape, n_temps, n_rep, n_dis):
    J = bimodal(shape, len(shape), n_dis, 11)
    temps = np.linspace(0.9, 2.1, n_temps)
    out = run_both(oracle, shape, J, temps, n_rep, 14, overlap_cluster_update_interval=1,
                   overlap_cluster_build_mode=build, overlap_cluster_mode=inner, collect_cluster_stats=True,
                   warmup_ratio=0.3, pt_interval=1)
    n_modes = build.count("+") + 1
    assert len(out["top_cluster_sizes"]) == n_modes and out["top_cluster_sizes"][0].shape == (n_temps, 4)
    assert len(out["overlap_csd"]) == n_modes


def test_statistics_of_both_cluster_kinds_with_diluted_couplings(oracle):
    shape = (5, 5, 4)
    J = bimodal(shape, 3, 2, 2) * (np.random.default_rng(1).random((2,) + shape + (3,)) < 0.8)
    out = run_both(oracle, shape, J.astype(np.float32), np.linspace(1.0, 2.5, 3), 2, 12,
                   cluster_update_interval=2, cluster_mode="sw", overlap_cluster_update_interval=3,
                   overlap_cluster_build_mode="jorg", overlap_cluster_mode="sw", collect_cluster_stats=True,
                   warmup_ratio=0.25)
    assert "fk_csd" in out and "overlap_csd" in out


def test_statistics_with_gaussian_couplings(oracle):
    shape = (4, 4, 4)
    J = np.random.default_rng(8).normal(size=shape + (3,)).astype(np.float32)
    run_both(oracle, shape, J, np.linspace(0.8, 2.0, 3), 2, 10, cluster_update_interval=1, cluster_mode="wolff",
             overlap_cluster_update_interval=2, overlap_cluster_build_mode="cmr", overlap_cluster_mode="wolff",
             collect_cluster_stats=True, warmup_ratio=0.2, pt_interval=1)


@pytest.mark.parametrize("build", ["houdayer", "jorg", "cmr", "houdayer+cmr"])
@pytest.mark.parametrize("shape", [(8, 8), (4, 4, 4)])
def test_overlap_graph_observation(oracle, build, shape):
    nn = len(shape)
    J = bimodal(shape, nn, 2, 4)
    temps = np.array([0.8, 1.4, 2.4], dtype=np.float32)
    out = run_both(oracle, shape, J, temps, 4, 9, overlap_cluster_update_interval=1,
                   overlap_cluster_build_mode=build, overlap_cluster_mode="sw",
                   overlap_cluster_action="observe", warmup_ratio=1 / 3, pt_interval=1)
    observed = out["per_disorder"]["cluster_observations"]
    expect = {"houdayer": "houdayer", "jorg": "jorg", "cmr": "cmr_blue"}
    assert set(observed) == {expect[m] for m in build.split("+")}
    for summary in observed.values():
        assert ("winding_both" in summary) == (nn == 2)
        assert summary["cluster_size_counts"].shape == (2, 3, int(np.prod(shape)) + 1)


def test_winding_is_seen_on_an_ordered_ferromagnet(oracle):
    # far below T_c nearly every bond of the (by then ordered or striped) 2D ferromagnet is active, so the
    # FK graph wraps the torus; far above, the bond density 1 - exp(-2/T) is below the percolation threshold
    shape = (8, 8)
    out = run_both(oracle, shape, ferro(shape, 2), np.array([0.5, 8.0], dtype=np.float32), 1, 120,
                   cluster_update_interval=1, cluster_mode="sw", cluster_action="observe", warmup_ratio=0.5)
    obs = out["per_disorder"]["cluster_observations"]["fk"]
    assert obs["winding_either"][0, 0] > 0.9 and obs["winding_either"][0, 1] < 0.1
    assert obs["top_four_component_fractions"][0, 0, :2].sum() > 0.9  # one domain, or two stripes
    assert obs["active_bond_density"][0, 0] > 0.7 > 0.3 > obs["active_bond_density"][0, 1]


SNAPSHOT_CASES = [
    ("houdayer", "wolff", (6, 6, 6), 2, 3, False),
    ("houdayer", "sw", (4, 4, 4), 4, 1, True),
    ("jorg", "wolff", (6, 6, 6), 2, 3, False),
    ("jorg", "sw", (4, 4, 4), 2, 2, False),
    ("cmr", "wolff", (4, 4, 4), 2, 3, False),
    ("cmr", "sw", (6, 6, 6), 4, 2, True),
    ("cmr+jorg", "wolff", (4, 4, 4), 2, 2, True),
    ("houdayer", "wolff", (8, 8), 2, 1, False),
]


@pytest.mark.parametrize("build,inner,shape,n_rep,n_dis,stats", SNAPSHOT_CASES)
def test_cluster_snapshots(oracle, build, inner, shape, n_rep, n_dis, stats):
    J = bimodal(shape, len(shape), n_dis, 21)
    temps = np.linspace(0.9, 2.0, 3)
    out = run_both(oracle, shape, J, temps, n_rep, 13, overlap_cluster_update_interval=2,
                   overlap_cluster_build_mode=build, overlap_cluster_mode=inner, snapshot_interval=4,
                   collect_cluster_stats=stats, warmup_ratio=0.25, pt_interval=1)
    snaps = out["cluster_snapshots"]
    assert [s["sweep_id"] for s in snaps] == [4, 8, 12]
    n = int(np.prod(shape))
    for snap in snaps:
        assert snap["cluster_ids"].shape == (3, n) and snap["spins"].shape == (3, 2, n)
        assert set(np.unique(snap["spins"])) <= {-1, 1}
        # ids are roots: every id names a site that carries its own id
        ids = snap["cluster_ids"]
        for t in range(3):
            assert np.all(ids[t][ids[t]] == ids[t])


def test_options_are_checked_like_the_reference():
    from peapods_amd import Ising

    model = Ising((4, 4), couplings="bimodal", temperatures=np.array([1.0, 2.0]), n_replicas=4, seed=1)
    with pytest.raises(ValueError, match="requires cluster_mode='sw'"):
        model.sample(4, cluster_update_interval=1, cluster_mode="wolff", cluster_action="observe")
    with pytest.raises(ValueError, match="requires overlap_cluster_mode='sw'"):
        model.sample(4, overlap_cluster_update_interval=1, overlap_cluster_action="observe")
    with pytest.raises(ValueError, match="positive multiple"):
        model.sample(4, overlap_cluster_update_interval=2, snapshot_interval=3)
    with pytest.raises(ValueError, match="not supported with overlap_cluster_action='observe'"):
        model.sample(4, overlap_cluster_update_interval=1, overlap_cluster_mode="sw",
                     overlap_cluster_action="observe", snapshot_interval=1)
    with pytest.raises(ValueError, match="houdN with N > 2"):
        model.sample(4, overlap_cluster_update_interval=1, overlap_cluster_mode="sw",
                     overlap_cluster_build_mode="houd4", overlap_cluster_action="observe")
    model.sample(8, cluster_update_interval=1, collect_cluster_stats=True, overlap_cluster_update_interval=1,
                 overlap_cluster_mode="sw", warmup_ratio=0)
    assert model.mean_cluster_size.shape == (2,) and len(model.top_cluster_sizes) == 1


N_RANDOM = int(__import__("os").environ.get("PEAPODS_RANDOM_STATS_CASES", "48"))  # raise for a longer hunt


@pytest.mark.parametrize("case_seed", range(N_RANDOM))
def test_random_configuration_with_cluster_reporting(oracle, case_seed):
    """The random configurations of test_gpu_random_configs.py with the reporting options layered on top:
    size statistics, observation instead of update, snapshots -- whatever the reference allows together."""
    from test_gpu_random_configs import random_case

    rng = np.random.default_rng(9000 + case_seed)
    shape, offsets, couplings, temps, n_rep, sweep_mode, opts = random_case(rng)
    # make sure at least one kind of cluster call is present
    if "cluster_update_interval" not in opts and ("overlap_cluster_update_interval" not in opts):
        if n_rep >= 2 and rng.random() < 0.5:
            opts["overlap_cluster_update_interval"] = 1
            opts["overlap_cluster_build_mode"] = str(rng.choice(["houdayer", "jorg", "cmr"]))
            opts["overlap_cluster_mode"] = str(rng.choice(["wolff", "sw"]))
        else:
            opts["cluster_update_interval"] = 1
            opts["cluster_mode"] = str(rng.choice(["wolff", "sw"]))
    opts["warmup_ratio"] = float(rng.choice([0.0, 0.25, 0.5]))
    roll = rng.random()
    if roll < 0.45:
        opts["collect_cluster_stats"] = True
    elif roll < 0.75:
        if "cluster_update_interval" in opts:
            opts["cluster_mode"] = "sw"
            opts["cluster_action"] = "observe"
        if "overlap_cluster_update_interval" in opts:
            opts["overlap_cluster_mode"] = "sw"
            opts["overlap_cluster_action"] = "observe"
            opts["overlap_cluster_build_mode"] = opts["overlap_cluster_build_mode"].replace("houd4", "houdayer")
    if "overlap_cluster_update_interval" in opts and opts.get("overlap_cluster_action") != "observe" and rng.random() < 0.5:
        opts["snapshot_interval"] = opts["overlap_cluster_update_interval"] * int(rng.choice([1, 2]))
    seed = int(rng.integers(0, 2**40))
    gpu, cpu = make_pair(oracle, shape, couplings, temps, n_rep, seed, offsets)
    for n_sweeps in (int(rng.integers(2, 10)), int(rng.integers(1, 8))):  # the second call continues the chains
        a = gpu.sample(n_sweeps, sweep_mode, **opts)
        b = cpu.sample(n_sweeps, sweep_mode, **opts)
        assert_same(a, b, f"shape={shape} offsets={offsets} n_temps={len(temps)} n_rep={n_rep} {sweep_mode} {opts}")
        assert_state_equal(gpu, cpu, check_energy=False)
