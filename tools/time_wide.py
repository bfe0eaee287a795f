#!/usr/bin/env python3
"""Step time of the bench ensemble on the bcc / fcc lattices (unit couplings, 4 to 6 forward directions):
python tools/time_wide.py [bcc|fcc] [steps]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peapods_amd import IsingSimulation  # noqa: E402
from peapods_amd.spin_models import generate_couplings, seed_material  # noqa: E402

OFFSETS = {"bcc": [[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]],
           "fcc": [[1, 1, 0], [1, 0, 1], [0, 1, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]]}
which = sys.argv[1] if len(sys.argv) > 1 else "fcc"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
work = bench.WORKLOADS["c2"]
seq, dyn = seed_material(bench.ROOT_SEED)
temps = np.linspace(*work["t_range"], work["n_temps"]).astype(np.float32)
offsets = OFFSETS[which]
J = np.stack(generate_couplings("bimodal", work["shape"], len(offsets), 128, seq, realizations=range(128)))
sim = IsingSimulation(list(work["shape"]), J, temps, 4, offsets, dyn)
cfg = lambda n: sim.make_config(n, "metropolis", warmup_ratio=0.0, pt_interval=1)  # noqa: E731
sim.run(cfg(30))
sim.synchronize()
t0 = time.perf_counter()
sim.run(cfg(steps), wait=False)
sim.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{which}: {dt * 1e6:.2f} us per step, {4096 * 24 * 4 * 128 / dt / 1e9:.1f} G attempts/s")
