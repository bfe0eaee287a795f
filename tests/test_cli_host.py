"""Host-side result surface: seed tree, npz key layout, TOML/flag merging (no GPU needed).

Mirrors what the reference's scripts rely on: python/peapods/sweep.py:13-43,100-163 and
python/peapods/cli.py:318-566.
"""
import types

import numpy as np
import pytest

from peapods_amd import cli, sweep


def test_seed_tree_is_the_numpy_seedsequence_tree():
    words = sweep.run_seed_words(1234)
    assert words == [int(w) for w in np.random.SeedSequence(1234).generate_state(4, dtype=np.uint32)]
    a = sweep.run_child_seed(words, "bimodal", (8, 8, 8))
    ref = np.random.SeedSequence(words, spawn_key=(1, 3, 8, 8, 8)).generate_state(1, dtype=np.uint64)[0]
    assert a == int(ref)
    # distinct runs get distinct seeds; the same run the same seed
    assert a == sweep.run_child_seed(words, "bimodal", (8, 8, 8))
    assert len({a, sweep.run_child_seed(words, "gaussian", (8, 8, 8)),
                sweep.run_child_seed(words, "bimodal", (8, 8)),
                sweep.run_child_seed(words, "bimodal", (8, 8, 16))}) == 4
    with pytest.raises(ValueError):
        sweep.run_seed_words(-1)
    with pytest.raises(ValueError):
        sweep.run_seed_words(1.5)


def test_labels_and_flatten():
    assert sweep.config_label("bimodal", "houdayer", "wolff") == "bimodal"
    assert sweep.config_label("bimodal", "cmr", "sw") == "bimodal_cmr_sw"
    assert sweep.size_label((8, 8, 16)) == "8x8x16"
    pd = {"parallel_tempering": {"attempts": np.arange(3)},
          "cluster_observations": {"fk": {"n_clusters": np.ones(2)}}}
    flat = sweep.flatten_per_disorder(pd, prefix="8x8")
    assert set(flat) == {"8x8_per_disorder_pt_attempts",
                         "8x8_per_disorder_cluster_observations_fk_n_clusters"}
    assert set(sweep.flatten_per_disorder(pd)) == {"per_disorder_pt_attempts",
                                                   "per_disorder_cluster_observations_fk_n_clusters"}
    assert sweep.flatten_per_disorder({}) == {}


def test_cumulative_overlap_ratio():
    # two samples, one temperature, 5 bins: q grid -1,-0.5,0,0.5,1
    hist = np.array([[[0, 0, 4, 0, 0]], [[2, 0, 0, 0, 2]]], dtype=np.float64)
    q, ratio, mean, median = sweep.cumulative_overlap_ratio(hist)
    np.testing.assert_allclose(q, [0.0, 0.5, 1.0])
    np.testing.assert_allclose(mean[0], [0.5, 0.5, 1.0])
    np.testing.assert_allclose(median[0], [0.5, 0.5, 1.0])
    np.testing.assert_allclose(ratio[0], [1.0, 1.0, 1.0])
    # an empty sample does not divide by zero
    hist[1] = 0
    _, ratio, mean, _ = sweep.cumulative_overlap_ratio(hist)
    assert np.isfinite(ratio).all() and mean[0, 0] == 0.5


def _fake_model():
    m = types.SimpleNamespace()
    m.lattice_shape = (4, 4)
    m.temperatures = np.array([1.0, 2.0], dtype=np.float32)
    m.binder_cumulant = np.array([0.6, 0.1])
    m.heat_capacity = np.array([1.0, 0.5])
    m.energies_avg = np.array([1.9, 0.7])
    m.sg_binder = np.array([0.3, 0.0])
    m.overlap_histogram = np.ones((2, 17), dtype=np.uint64)
    m.per_sample_overlap_histogram = np.ones((3, 2, 17), dtype=np.uint64)
    m.mags2_tau = np.array([2.0, 1.0])
    m._equil_sweeps = np.array([128, 256])
    m._equil_energy_avg = np.zeros((2, 2))
    m._equil_link_overlap_avg = np.zeros((2, 2))
    m.per_disorder = {"parallel_tempering": {"attempts": np.zeros((3, 1))}}
    return m


def test_sweep_file_keys(tmp_path):
    path = sweep.write_sweep_file({"4x4": _fake_model()}, "bimodal", np.array([1.0, 2.0]), tmp_path)
    assert path.name == "sweep_bimodal.npz"
    with np.load(path) as z:
        keys = set(z.files)
        assert keys == {
            "temperatures", "4x4_lattice_shape", "4x4_binder_cumulant", "4x4_heat_capacity",
            "4x4_energies", "4x4_sg_binder", "4x4_overlap_histogram",
            "4x4_per_sample_overlap_histogram", "4x4_cumulative_overlap_q",
            "4x4_cumulative_overlap_ratio", "4x4_mags2_tau", "4x4_equil_sweeps",
            "4x4_equil_energy_avg", "4x4_equil_link_overlap_avg", "4x4_per_disorder_pt_attempts"}
        assert z["4x4_cumulative_overlap_ratio"].shape == (2, 9)
        np.testing.assert_array_equal(z["4x4_lattice_shape"], [4, 4])


def test_simulate_arrays_keys():
    m = _fake_model()
    result = {"mags": 1, "mags2": 2, "mags4": 3, "energies": 4, "energies2": 5, "overlap2": 6}
    d = sweep.simulate_arrays(m, result)
    assert {"temperatures", "binder_cumulant", "heat_capacity", "mags", "mags2", "mags4", "energies",
            "energies2", "overlap2", "sg_binder", "per_sample_overlap_histogram",
            "per_disorder_pt_attempts"} == set(d)


def test_print_table(capsys):
    m = _fake_model()
    m.top_cluster_sizes = np.array([[0.5, 0.25, 0.125, 0.0625]] * 2)
    cli.print_table(m)
    lines = capsys.readouterr().out.splitlines()
    assert lines[0].split()[:4] == ["T", "E", "Binder", "C_v"] and "Top-4 Clusters" in lines[0]
    assert set(lines[1]) == {"-"} and len(lines) == 4
    assert "(0.500, 0.250, 0.125, 0.062)" in lines[2]


def test_simulate_flags_and_defaults():
    args = cli.build_parser().parse_args(
        ["simulate", "--shape", "8", "8", "--temp-min", "1", "--temp-max", "3", "--n-sweeps", "10"])
    assert args.shape == [8, 8] and args.couplings == "ferro" and args.n_temps == 32
    assert args.temp_scale == "log" and args.warmup_ratio == 0.25 and args.output is None
    kw = cli.sample_kwargs(args)
    assert kw["sweep_mode"] == "metropolis" and kw["cluster_mode"] == "sw"
    assert kw["pt_schedule"] == "single_random_edge" and kw["overlap_cluster_mode"] == "wolff"
    assert kw["collect_cluster_stats"] is False and kw["autocorrelation_backend"] == "ring"
    np.testing.assert_allclose(cli.temperature_grid(1, 4, 3, "log"), [1, 2, 4])
    np.testing.assert_allclose(cli.temperature_grid(1, 4, 4, "linear"), [1, 2, 3, 4])
    with pytest.raises(SystemExit):  # required flags
        cli.build_parser().parse_args(["bench", "--shape", "8"])


TOML = """
[lattice]
sizes = [[4, 4], [6, 6]]
couplings = ["bimodal"]
[temperatures]
min = 0.5
max = 2.0
count = 6
scale = "linear"
[replicas]
n_replicas = 2
n_disorder = 3
[sampling]
n_sweeps = 100
warmup_ratio = 0.5
seed = 9
[parallel_tempering]
interval = 1
schedule = "full_ladder"
[overlap_cluster]
interval = 2
build_modes = ["houdayer", "jorg"]
cluster_mode = "sw"
[diagnostics]
equilibration_diagnostic = true
[diagnostics.autocorrelation]
max_lag = 16
[output]
save_data = true
dir = "out"
"""


def test_sweep_config_merge(tmp_path):
    cfg = tmp_path / "run.toml"
    cfg.write_text(TOML)
    parser = cli.build_parser()
    kw = cli.sweep_settings(parser.parse_args(["sweep", "--config", str(cfg)]))
    assert kw["sizes"] == [(4, 4), (6, 6)] and kw["couplings"] == ("bimodal",)
    assert (kw["temp_min"], kw["temp_max"], kw["n_temps"], kw["temp_scale"]) == (0.5, 2.0, 6, "linear")
    assert kw["n_replicas"] == 2 and kw["n_disorder"] == 3 and kw["seed"] == 9
    assert kw["pt_interval"] == 1 and kw["pt_schedule"] == "full_ladder"
    assert kw["overlap_cluster_build_mode"] == ("houdayer", "jorg")
    assert kw["overlap_cluster_mode"] == ("sw",) and kw["overlap_cluster_update_interval"] == 2
    assert kw["equilibration_diagnostic"] is True and kw["autocorrelation_max_lag"] == 16
    assert kw["save_data"] is True and kw["output_dir"] == "out" and kw["warmup_ratio"] == 0.5
    assert kw["cluster_mode"] == "sw" and kw["save_plots"] is False  # defaults
    # flags override the file
    kw = cli.sweep_settings(parser.parse_args(
        ["sweep", "--config", str(cfg), "--sizes", "8,8,8", "--n-sweeps", "7", "--seed", "1",
         "--overlap-cluster-mode", "wolff", "sw"]))
    assert kw["sizes"] == [(8, 8, 8)] and kw["n_sweeps"] == 7 and kw["seed"] == 1
    assert kw["overlap_cluster_mode"] == ["wolff", "sw"]
    # missing essentials
    with pytest.raises(SystemExit):
        cli.sweep_settings(parser.parse_args(["sweep", "--sizes", "4,4"]))
