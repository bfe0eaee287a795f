#!/usr/bin/env python3
"""Times the sweep pipeline on the C3 ensemble shape for every acceptance rule:
unit couplings (integer table) x {metropolis, gibbs} and Gaussian couplings (logarithmic rule)
x {metropolis, gibbs}.  Prints ms per step and attempts/s; diagnostic only."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from peapods_amd import IsingSimulation  # noqa: E402

SHAPE = (16, 16, 16)
N_T, N_R = 24, 4
n_dis = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(0)
temps = np.linspace(0.8, 1.4, N_T).astype(np.float32)
out = {}
for kind in ("bimodal", "gaussian"):
    if kind == "bimodal":
        J = (2 * rng.integers(0, 2, size=(n_dis,) + SHAPE + (3,)) - 1).astype(np.float32)
    else:
        J = rng.standard_normal(size=(n_dis,) + SHAPE + (3,)).astype(np.float32)
    sim = IsingSimulation(list(SHAPE), J, temps, N_R, None, 42)
    for mode in ("metropolis", "gibbs"):
        cfg = sim.make_config(steps, mode, pt_interval=1, warmup_ratio=0.0)
        sim.run(sim.make_config(20, mode, pt_interval=1, warmup_ratio=0.0))
        t0 = time.perf_counter()
        sim.run(cfg)
        dt = time.perf_counter() - t0
        attempts = 4096 * N_T * N_R * n_dis * steps
        k = sim.profile_kernels(sim.make_config(32, mode, pt_interval=1, warmup_ratio=0.0), 32)
        out[f"{kind}/{mode}"] = {"ms_per_step": dt / steps * 1e3, "gflips": attempts / dt / 1e9,
                                 "kernels_ms": k}
    del sim
print(json.dumps(out, indent=1))
