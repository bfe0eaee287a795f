"""`simulate` / `bench` / `sweep` on the GPU engine: the files and tables they produce hold exactly the
numbers of an oracle run fed the same couplings, temperatures and seeds (reference
python/peapods/cli.py:595-653, python/peapods/sweep.py:100-163,351-512)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_result(oracle, model, n_sweeps, **kw):
    cpu = oracle.OracleSimulation(list(model.lattice_shape), model.couplings, model.temperatures,
                                  model.n_replicas, None, model._constructor_dynamics_seed)
    return cpu.sample(n_sweeps, "metropolis", **kw)


def test_simulate_writes_the_reference_npz(oracle, tmp_path, capsys):
    from peapods_amd import cli

    out = tmp_path / "run.npz"
    argv = ["simulate", "--shape", "6", "6", "6", "--couplings", "bimodal", "--temp-min", "0.8",
            "--temp-max", "2.4", "--n-temps", "5", "--n-replicas", "2", "--n-disorder", "2",
            "--seed", "11", "--n-sweeps", "40", "--pt-interval", "1", "--warmup-ratio", "0.25",
            "--overlap-cluster-update-interval", "2", "-o", str(out)]
    args = cli.build_parser().parse_args(argv)
    model = cli.run_simulate(args)
    text = capsys.readouterr().out
    assert "Overlap Binder" in text.splitlines()[0] and f"Results saved to {out}" in text

    ref = oracle_result(oracle, model, 40, pt_interval=1, warmup_ratio=0.25,
                        overlap_cluster_update_interval=2, overlap_cluster_build_mode="houdayer",
                        overlap_cluster_mode="wolff")
    with np.load(out) as z:
        for key in ("mags", "mags2", "mags4", "energies", "energies2", "overlap", "overlap2",
                    "overlap4"):
            np.testing.assert_array_equal(z[key], ref[key], err_msg=key)
        np.testing.assert_array_equal(z["temperatures"], np.geomspace(0.8, 2.4, 5).astype(np.float32))
        np.testing.assert_array_equal(z["sg_binder"], 1 - ref["overlap4"] / (3 * ref["overlap2"] ** 2))
        np.testing.assert_array_equal(
            z["heat_capacity"],
            216 * (ref["energies2"] - ref["energies"] ** 2) / z["temperatures"] ** 2)
        np.testing.assert_array_equal(z["per_sample_overlap_histogram"],
                                      ref["per_sample_overlap_histogram"])
        for name, values in ref["per_disorder"]["parallel_tempering"].items():
            np.testing.assert_array_equal(z[f"per_disorder_pt_{name}"], values, err_msg=name)


def test_bench_reports_time_per_sweep(capsys):
    from peapods_amd import cli

    cli.main(["bench", "--shape", "8", "8", "--temp-min", "1.5", "--temp-max", "3", "--n-temps", "4",
              "--n-sweeps", "20", "--cluster-interval", "1"])
    lines = capsys.readouterr().out.splitlines()
    assert lines[0] == "Lattice: 8x8  |  Temps: 4  |  Sweeps: 20"
    assert lines[1].startswith("Total: ") and lines[1].endswith(" ms/sweep")


def test_sweep_runs_every_configuration_with_its_own_seed(oracle, tmp_path):
    from peapods_amd import cli, sweep

    cfg = tmp_path / "glass.toml"
    cfg.write_text(f"""
[lattice]
sizes = [[4, 4, 4], [6, 6, 6]]
couplings = ["bimodal"]
[temperatures]
min = 0.9
max = 2.0
count = 4
[replicas]
n_replicas = 2
n_disorder = 2
[sampling]
n_sweeps = 30
warmup_ratio = 0.0
seed = 5
[parallel_tempering]
interval = 1
[overlap_cluster]
interval = 1
build_modes = ["houdayer", "jorg"]
[diagnostics]
equilibration_diagnostic = true
[diagnostics.autocorrelation]
max_lag = 8
[output]
save_data = true
dir = "{tmp_path}"
""")
    results = cli.run_sweep_cli(cli.build_parser().parse_args(["sweep", "--config", str(cfg)]))
    assert list(results) == ["bimodal", "bimodal_jorg"]
    assert list(results["bimodal"]) == ["4x4x4", "6x6x6"]
    root = sweep.run_seed_words(5)
    temps = np.geomspace(0.9, 2.0, 4)
    for label, build_mode in (("bimodal", "houdayer"), ("bimodal_jorg", "jorg")):
        with np.load(tmp_path / f"sweep_{label}.npz") as z:
            np.testing.assert_array_equal(z["temperatures"], temps)
            for shape in ((4, 4, 4), (6, 6, 6)):
                slabel = sweep.size_label(shape)
                model = results[label][slabel]
                assert model.seed == sweep.run_child_seed(root, "bimodal", shape)
                ref = oracle_result(oracle, model, 30, pt_interval=1, warmup_ratio=0.0,
                                    overlap_cluster_update_interval=1,
                                    overlap_cluster_build_mode=build_mode,
                                    overlap_cluster_mode="wolff", autocorrelation_max_lag=8,
                                    equilibration_diagnostic=True)
                np.testing.assert_array_equal(z[f"{slabel}_lattice_shape"], shape)
                np.testing.assert_array_equal(z[f"{slabel}_energies"], ref["energies"])
                np.testing.assert_array_equal(
                    z[f"{slabel}_sg_binder"], 1 - ref["overlap4"] / (3 * ref["overlap2"] ** 2))
                np.testing.assert_array_equal(z[f"{slabel}_overlap_histogram"],
                                              np.array(list(ref["overlap_histogram"])))
                np.testing.assert_array_equal(z[f"{slabel}_mags2_tau"], ref["mags2_tau"])
                np.testing.assert_array_equal(z[f"{slabel}_overlap2_tau"], ref["overlap2_tau"])
                np.testing.assert_array_equal(z[f"{slabel}_equil_sweeps"], ref["equil_sweeps"])
                np.testing.assert_array_equal(z[f"{slabel}_equil_energy_avg"],
                                              ref["equil_energy_avg"])
                np.testing.assert_array_equal(z[f"{slabel}_equil_link_overlap_avg"],
                                              ref["equil_link_overlap_avg"])
                assert z[f"{slabel}_cumulative_overlap_ratio"].shape[0] == 4
                assert f"{slabel}_per_disorder_pt_edge_attempts" in z.files
    # both configurations of one size share couplings and seed: only the move differs
    a, b = results["bimodal"]["4x4x4"], results["bimodal_jorg"]["4x4x4"]
    np.testing.assert_array_equal(a.couplings, b.couplings)
