"""Randomised differential test: many small random configurations (lattice shape and offsets,
couplings, temperatures, replicas, disorder, every sampling option) run on the HIP engine and on
the CPU oracle; all outputs and the final state must agree bit for bit."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("PEAPODS_RANDOM_CASES", "60"))  # raise for a longer hunt

OFFSETS = {
    1: [None],
    2: [None, [[1, 0], [0, 1], [1, -1]], [[1, 1], [1, -1]], [[1, 0]]],
    3: [None, [[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]],
        [[1, 1, 0], [1, 0, 1], [0, 1, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]],
        [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2, 0, 1]], [[0, 0, 1], [1, 2, 0]]],
    4: [None, [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]]],
}


def random_case(rng):
    n_dims = int(rng.choice([1, 2, 2, 3, 3, 3, 4]))
    max_extent = {1: 40, 2: 9, 3: 6, 4: 4}[n_dims]
    shape = tuple(int(x) for x in rng.integers(1, max_extent + 1, size=n_dims))
    offsets = OFFSETS[n_dims][int(rng.integers(len(OFFSETS[n_dims])))]
    nn = len(offsets) if offsets else n_dims
    n_dis = int(rng.choice([1, 1, 2, 3]))
    kind = str(rng.choice(["bimodal", "ferro", "diluted", "gaussian", "antiferro"]))
    full = ((n_dis,) if n_dis > 1 else ()) + shape + (nn,)
    if kind == "bimodal":
        couplings = 2 * rng.integers(0, 2, size=full) - 1
    elif kind == "ferro":
        couplings = np.ones(full)
    elif kind == "antiferro":
        couplings = -np.ones(full)
    elif kind == "diluted":
        couplings = rng.integers(-1, 2, size=full)
    else:
        couplings = rng.standard_normal(size=full)
    n_temps = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
    n_rep = int(rng.choice([1, 2, 2, 3, 4, 5, 8]))
    temps = np.sort(rng.uniform(0.3, 5.0, size=n_temps))
    if rng.random() < 0.15:
        rng.shuffle(temps)  # unsorted ladders are legal (hot / cold slots by value)
    opts = dict(warmup_ratio=float(rng.choice([0.0, 0.25, 0.5, 1.0])))
    if rng.random() < 0.7:
        opts["pt_interval"] = int(rng.choice([1, 1, 2, 3]))
        opts["pt_schedule"] = str(rng.choice(["single_random_edge", "full_ladder"]))
    if n_rep >= 2 and rng.random() < 0.6:
        modes = ["houdayer", "jorg", "cmr"]
        if n_rep >= 4:
            modes.append("houd4")
        picked = [str(m) for m in rng.choice(modes, size=int(rng.integers(1, 4)))]
        opts["overlap_cluster_update_interval"] = int(rng.choice([1, 2]))
        opts["overlap_cluster_build_mode"] = "+".join(picked)
        opts["overlap_cluster_mode"] = str(rng.choice(["wolff", "sw"]))
    if rng.random() < 0.4:
        opts["cluster_update_interval"] = int(rng.choice([1, 3]))
        opts["cluster_mode"] = str(rng.choice(["wolff", "sw"]))
    if rng.random() < 0.25:
        opts["equilibration_diagnostic"] = True
    if rng.random() < 0.25:
        opts["autocorrelation_max_lag"] = int(rng.choice([1, 3, 50]))
    sweep_mode = str(rng.choice(["metropolis", "gibbs"]))
    return shape, offsets, couplings.astype(np.float32), temps.astype(np.float32), n_rep, sweep_mode, opts


@pytest.mark.parametrize("case_seed", range(N_CASES))
def test_random_configuration_matches_oracle(oracle, case_seed):
    from peapods_amd import IsingSimulation

    rng = np.random.default_rng(1000 + case_seed)
    shape, offsets, couplings, temps, n_rep, sweep_mode, opts = random_case(rng)
    seed = int(rng.integers(0, 2**40))
    gpu = IsingSimulation(list(shape), couplings, temps, n_rep, offsets, seed)
    cpu = oracle.OracleSimulation(list(shape), couplings, temps, n_rep, offsets, seed)
    for n_sweeps in (int(rng.integers(1, 9)), int(rng.integers(1, 14))):  # second call continues the chains
        a = gpu.sample(n_sweeps, sweep_mode, **opts)
        b = cpu.sample(n_sweeps, sweep_mode, **opts)
        context = f"shape={shape} offsets={offsets} n_temps={len(temps)} n_rep={n_rep} {sweep_mode} {opts}"
        assert set(a.keys()) == set(b.keys()), context
        for key in a:
            if key == "per_disorder":
                for k2 in a[key]["parallel_tempering"]:
                    np.testing.assert_array_equal(a[key]["parallel_tempering"][k2],
                                                  b[key]["parallel_tempering"][k2], err_msg=f"{k2} {context}")
            elif key == "overlap_histogram":
                for x, y in zip(a[key], b[key]):
                    np.testing.assert_array_equal(x, y, err_msg=context)
            else:
                np.testing.assert_array_equal(a[key], b[key], err_msg=f"{key} {context}")
        for r in range(gpu.n_realizations):
            np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r), err_msg=context)
            np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r), err_msg=context)
            np.testing.assert_array_equal(gpu.get_energies(r), cpu.get_energies(r), err_msg=context)
            for s in range(gpu.n_systems):
                np.testing.assert_array_equal(gpu.get_rng_state(r, s), cpu.get_rng_state(r, s), err_msg=context)


@pytest.mark.parametrize("case_seed", range(32))
def test_random_large_lattice_configuration_matches_oracle(oracle, case_seed):
    """The same on lattices that do not fit the LDS kernels (pipelined sweep / global-memory general
    path): no cluster moves there."""
    from peapods_amd import IsingSimulation

    rng = np.random.default_rng(5000 + case_seed)
    shape = [(26, 26, 26), (150, 120), (20, 33, 31), (13000,), (40, 20, 20), (128, 130), (24, 24, 30),
             (27, 25, 23)][case_seed % 8]
    n_dims = len(shape)
    offsets = [[1, 0], [0, 1], [1, -1]] if (n_dims == 2 and case_seed % 2) else None
    nn = len(offsets) if offsets else n_dims
    kind = ["bimodal", "gaussian", "ferro", "diluted"][int(rng.integers(4))]
    full = tuple(shape) + (nn,)
    couplings = {"bimodal": lambda: 2 * rng.integers(0, 2, size=full) - 1, "ferro": lambda: np.ones(full),
                 "diluted": lambda: rng.integers(-1, 2, size=full),
                 "gaussian": lambda: rng.standard_normal(size=full)}[kind]().astype(np.float32)
    n_temps, n_rep = int(rng.choice([3, 9, 20])), int(rng.choice([1, 2, 4]))
    temps = np.sort(rng.uniform(0.5, 5.0, size=n_temps)).astype(np.float32)
    opts = dict(warmup_ratio=0.25, pt_interval=int(rng.choice([1, 2])),
                pt_schedule=str(rng.choice(["single_random_edge", "full_ladder"])))
    if rng.random() < 0.5:
        opts["equilibration_diagnostic"] = True
    sweep_mode = str(rng.choice(["metropolis", "gibbs"]))
    gpu = IsingSimulation(list(shape), couplings, temps, n_rep, offsets, 99 + case_seed)
    cpu = oracle.OracleSimulation(list(shape), couplings, temps, n_rep, offsets, 99 + case_seed)
    for n_sweeps in (4, 3):
        a = gpu.sample(n_sweeps, sweep_mode, **opts)
        b = cpu.sample(n_sweeps, sweep_mode, n_threads=8, **opts)
        for key in b:
            if key == "per_disorder":
                for k2 in a[key]["parallel_tempering"]:
                    np.testing.assert_array_equal(a[key]["parallel_tempering"][k2], b[key]["parallel_tempering"][k2])
            elif key == "overlap_histogram":
                for x, y in zip(a[key], b[key]):
                    np.testing.assert_array_equal(x, y)
            else:
                np.testing.assert_array_equal(a[key], b[key], err_msg=key)
        np.testing.assert_array_equal(gpu.get_spins(0), cpu.get_spins(0))
        np.testing.assert_array_equal(gpu.get_system_ids(0), cpu.get_system_ids(0))
        np.testing.assert_array_equal(gpu.get_energies(0), cpu.get_energies(0))
