#!/usr/bin/env python3
"""Steady-state step time of the bench ensemble (c2) under both single-site rules."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

work = bench.WORKLOADS["c2"]
sim, *_ = bench.build_ensemble(work, 0, 128, 1, 0)
for mode in ("metropolis", "gibbs"):
    cfg = lambda n: sim.make_config(n, mode, warmup_ratio=0.0, pt_interval=1)  # noqa: E731
    sim.run(cfg(100))
    sim.synchronize()
    t0 = time.perf_counter()
    sim.run(cfg(1000), wait=False)
    sim.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {dt / 1000 * 1e6:.2f} us per step, {4096 * 96 * 128 / (dt / 1000) / 1e9:.1f} G attempts/s", flush=True)
