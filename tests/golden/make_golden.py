#!/usr/bin/env python3
"""Writes tests/golden/*.npz: inputs and expected outputs of small runs.

The vectors are produced by the CPU oracle (oracle/peapods_oracle.c), which is pinned by the
reference's own known-answer and physics tests (tests/test_oracle_kat.py) -- the reference itself
cannot be built in this image (no Rust toolchain), so these are NOT outputs of upstream binaries.
They freeze the oracle's behaviour (tests/test_golden.py, CPU) and give the GPU parity tests a
target that needs no oracle at run time (tests/test_gpu_golden.py).

    python tests/golden/make_golden.py        # regenerate (only after a deliberate oracle change)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

CASES = {
    # BASELINE.json configs[0]: 2D 32x32 ferromagnet, 8 temperatures, 1 replica, Metropolis
    "c1_ferro_32x32": dict(shape=(32, 32), kind="ferro", n_dis=1, temps=np.linspace(1.5, 3.2, 8), n_rep=1,
                           seed=42, n_sweeps=16, sweep_mode="metropolis", opts=dict(warmup_ratio=0.25)),
    "glass_6cubed_pt_houdayer": dict(shape=(6, 6, 6), kind="bimodal", n_dis=2, temps=np.linspace(0.8, 1.6, 5), n_rep=4,
                                     seed=7, n_sweeps=24, sweep_mode="metropolis",
                                     opts=dict(warmup_ratio=0.25, pt_interval=1, pt_schedule="full_ladder",
                                               overlap_cluster_update_interval=2,
                                               overlap_cluster_build_mode="houdayer",
                                               overlap_cluster_mode="wolff")),
    "gaussian_4cubed_gibbs_jorg": dict(shape=(4, 4, 4), kind="gaussian", n_dis=1, temps=np.array([0.7, 1.1, 1.9]), n_rep=2,
                                       seed=3, n_sweeps=20, sweep_mode="gibbs",
                                       opts=dict(warmup_ratio=0.0, pt_interval=2, overlap_cluster_update_interval=1,
                                                 overlap_cluster_build_mode="jorg", overlap_cluster_mode="sw")),
    "ferro_8x8_sw_statistics": dict(shape=(8, 8), kind="ferro", n_dis=1, temps=np.array([1.9, 2.269, 2.7]), n_rep=2,
                                    seed=11, n_sweeps=18, sweep_mode="metropolis",
                                    opts=dict(warmup_ratio=0.5, cluster_update_interval=1, cluster_mode="sw",
                                              collect_cluster_stats=True)),
    "ferro_8x8_observed": dict(shape=(8, 8), kind="ferro", n_dis=2, temps=np.array([1.5, 2.269, 3.5]), n_rep=1,
                               seed=19, n_sweeps=12, sweep_mode="metropolis",
                               opts=dict(warmup_ratio=0.25, cluster_update_interval=2, cluster_mode="sw",
                                         cluster_action="observe")),
    "glass_4cubed_cmr_snapshots": dict(shape=(4, 4, 4), kind="bimodal", n_dis=2, temps=np.array([0.9, 1.3, 2.0]), n_rep=2,
                                       seed=23, n_sweeps=13, sweep_mode="metropolis",
                                       opts=dict(warmup_ratio=0.25, pt_interval=1, overlap_cluster_update_interval=2,
                                                 overlap_cluster_build_mode="cmr", overlap_cluster_mode="sw",
                                                 snapshot_interval=4, collect_cluster_stats=True)),
}


def couplings_of(case):
    rng = np.random.default_rng(1000 + case["seed"])
    shape, nn = tuple(case["shape"]), len(case["shape"])
    full = ((case["n_dis"],) if case["n_dis"] > 1 else ()) + shape + (nn,)
    if case["kind"] == "ferro":
        return np.ones(full, dtype=np.float32)
    if case["kind"] == "bimodal":
        return (2 * rng.integers(0, 2, size=full) - 1).astype(np.float32)
    return rng.standard_normal(size=full).astype(np.float32)


def flatten(value, prefix, out):
    if isinstance(value, dict):
        for key, item in value.items():
            flatten(item, f"{prefix}/{key}", out)
    elif isinstance(value, (list, tuple)):
        out[f"{prefix}/#len"] = np.array(len(value))
        for k, item in enumerate(value):
            flatten(item, f"{prefix}/{k}", out)
    else:
        out[prefix] = np.asarray(value)


def run(sim, case):
    """Result dictionary + final state of a finished simulation object (oracle or engine), flattened."""
    result = sim.sample(case["n_sweeps"], case["sweep_mode"], **case["opts"])
    out = {}
    flatten(result, "result", out)
    for r in range(sim.n_realizations):
        out[f"state/spins/{r}"] = sim.get_spins(r)
        out[f"state/system_ids/{r}"] = sim.get_system_ids(r)
        out[f"state/rng0/{r}"] = sim.get_rng_state(r, 0)
    return out


def main():
    from oracle import bindings

    bindings.build()
    for name, case in CASES.items():
        J = couplings_of(case)
        temps = np.asarray(case["temps"], dtype=np.float32)
        sim = bindings.OracleSimulation(list(case["shape"]), J, temps, case["n_rep"], None, case["seed"])
        expected = run(sim, case)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), couplings=J, temperatures=temps, **expected)
        print(name, sum(v.nbytes for v in expected.values()), "bytes of expected output")


if __name__ == "__main__":
    main()
