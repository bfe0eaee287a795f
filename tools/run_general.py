#!/usr/bin/env python3
"""Runs the general-coupling (Gaussian) pipeline on the C3 ensemble shape; for rocprofv3."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from peapods_amd import IsingSimulation  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "metropolis"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_dis = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rng = np.random.default_rng(0)
J = rng.standard_normal(size=(n_dis, 16, 16, 16, 3)).astype(np.float32)
sim = IsingSimulation([16, 16, 16], J, np.linspace(0.8, 1.4, 24).astype(np.float32), 4, None, 42)
sim.run(sim.make_config(steps, mode, pt_interval=1, warmup_ratio=0.0))
