#!/usr/bin/env python3
"""Per-call fixed cost of sample(): wall time of K-step calls for several K on the bench ensemble
(a straight line through them: slope = steady-state step, intercept = what every call pays once)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

work = bench.WORKLOADS["c2"]
sim, *_ = bench.build_ensemble(work, 0, 128, 1, 0)
cfg = lambda n: sim.make_config(n, "metropolis", warmup_ratio=0.0, pt_interval=1)  # noqa: E731
sim.run(cfg(200))
rows = []
for k in (1, 5, 20, 20, 20, 100, 500, 2000, 20, 20):
    sim.synchronize()
    t0 = time.perf_counter()
    sim.run(cfg(k), wait=False)
    t1 = time.perf_counter()
    sim.synchronize()
    t2 = time.perf_counter()
    rows.append({"steps": k, "enqueue_ms": (t1 - t0) * 1e3, "total_ms": (t2 - t0) * 1e3, "us_per_step": (t2 - t0) * 1e6 / k})
ks = np.array([r["steps"] for r in rows], dtype=float)
ts = np.array([r["total_ms"] for r in rows])
slope, intercept = np.polyfit(ks, ts, 1)
print(json.dumps({"rows": rows, "fit_us_per_step": slope * 1e3, "fit_fixed_ms": intercept}, indent=1))
