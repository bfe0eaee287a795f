#!/usr/bin/env python3
"""Step time of protocol variants on the bench lattice (16^3 +-J, 24 T, 128 disorder): which paths are fast."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peapods_amd import IsingSimulation  # noqa: E402
from peapods_amd.spin_models import generate_couplings, seed_material  # noqa: E402

work = bench.WORKLOADS["c2"]
coupling_seq, dynamics_seed = seed_material(bench.ROOT_SEED)
temps = np.linspace(*work["t_range"], work["n_temps"]).astype(np.float32)


def timed(label, n_rep, mode, couplings="bimodal", dilution=0.0, offsets=None, **kw):
    nn = len(offsets) if offsets else 3
    J = np.stack(generate_couplings(couplings, work["shape"], nn, 128, coupling_seq, realizations=range(128)))
    if dilution:
        J = J * (np.random.default_rng(5).random(J.shape) >= dilution)
    sim = IsingSimulation(list(work["shape"]), J, temps, n_rep, offsets, dynamics_seed)
    cfg = lambda n: sim.make_config(n, mode, **kw)  # noqa: E731
    sim.run(cfg(60))
    sim.synchronize()
    t0 = time.perf_counter()
    sim.run(cfg(600), wait=False)
    sim.synchronize()
    dt = (time.perf_counter() - t0) / 600
    print(f"{label:58s} {dt * 1e6:8.2f} us per step  {4096 * 24 * n_rep * 128 / dt / 1e9:7.1f} G attempts/s", flush=True)


timed("metropolis, 4 replicas, PT every sweep", 4, "metropolis", warmup_ratio=0.0, pt_interval=1)
timed("metropolis, 4 replicas, PT every 10th sweep", 4, "metropolis", warmup_ratio=0.0, pt_interval=10)
timed("metropolis, 4 replicas, no PT", 4, "metropolis", warmup_ratio=0.0)
timed("metropolis, 4 replicas, warm-up only (nothing recorded)", 4, "metropolis", warmup_ratio=1.0, pt_interval=1)
timed("metropolis, 1 replica (no overlap), PT every sweep", 1, "metropolis", warmup_ratio=0.0, pt_interval=1)
timed("metropolis, 2 replicas, PT every sweep", 2, "metropolis", warmup_ratio=0.0, pt_interval=1)
timed("gibbs, 4 replicas, PT every sweep", 4, "gibbs", warmup_ratio=0.0, pt_interval=1)
timed("gibbs, 1 replica, PT every sweep", 1, "gibbs", warmup_ratio=0.0, pt_interval=1)
timed("metropolis, 4 replicas, autocorrelation", 4, "metropolis", warmup_ratio=0.0, pt_interval=1, autocorrelation_max_lag=100)
timed("metropolis, 4 replicas, Gaussian couplings", 4, "metropolis", couplings="gaussian", warmup_ratio=0.0, pt_interval=1)
timed("metropolis, 4 replicas, +-J with 20 % of the bonds removed", 4, "metropolis", dilution=0.2, warmup_ratio=0.0, pt_interval=1)
timed("gibbs, 4 replicas, +-J with 20 % of the bonds removed", 4, "gibbs", dilution=0.2, warmup_ratio=0.0, pt_interval=1)
BCC = [[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]]
FCC = [[1, 1, 0], [1, 0, 1], [0, 1, 1], [1, -1, 0], [1, 0, -1], [0, 1, -1]]
timed("metropolis, 4 replicas, +-J on the bcc lattice (8 neighbours)", 4, "metropolis", offsets=BCC, warmup_ratio=0.0, pt_interval=1)
timed("metropolis, 4 replicas, +-J on the fcc lattice (12 neighbours)", 4, "metropolis", offsets=FCC, warmup_ratio=0.0, pt_interval=1)
