#!/usr/bin/env python3
"""Step time of the bench ensemble with Gaussian couplings (under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peapods_amd import IsingSimulation  # noqa: E402
from peapods_amd.spin_models import generate_couplings, seed_material  # noqa: E402

work = bench.WORKLOADS["c2"]
seq, dyn = seed_material(bench.ROOT_SEED)
temps = np.linspace(*work["t_range"], work["n_temps"]).astype(np.float32)
mode = sys.argv[1] if len(sys.argv) > 1 else "metropolis"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
J = np.stack(generate_couplings("gaussian", work["shape"], 3, 128, seq, realizations=range(128)))
sim = IsingSimulation(list(work["shape"]), J, temps, 4, None, dyn)
cfg = lambda n: sim.make_config(n, mode, warmup_ratio=0.0, pt_interval=1)  # noqa: E731
sim.run(cfg(30))
sim.synchronize()
t0 = time.perf_counter()
sim.run(cfg(steps), wait=False)
sim.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"gaussian {mode}: {dt * 1e6:.2f} us per step, {4096 * 24 * 4 * 128 / dt / 1e9:.1f} G attempts/s")
