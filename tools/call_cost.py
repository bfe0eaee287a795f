#!/usr/bin/env python3
"""Where a short sample() call spends its time: wall time (min of several) of K-step calls for small K, the host's
enqueue part, and the cost of make_config / synchronize alone."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

work = bench.WORKLOADS["c2"]
sim, *_ = bench.build_ensemble(work, 0, 128, 1, 0)
cfg = lambda n: sim.make_config(n, "metropolis", warmup_ratio=0.0, pt_interval=1)  # noqa: E731
sim.run(cfg(200))
out = {}
t = []
for _ in range(20):
    t0 = time.perf_counter(); c = cfg(20); t.append((time.perf_counter() - t0) * 1e6)
out["make_config_us"] = min(t)
t = []
for _ in range(20):
    sim.synchronize(); t0 = time.perf_counter(); sim.synchronize(); t.append((time.perf_counter() - t0) * 1e6)
out["idle_synchronize_us"] = min(t)
rows = []
for k in (1, 2, 3, 5, 10, 20, 40):
    c = cfg(k)
    best = None
    every = []
    for _ in range(7):
        sim.synchronize()
        t0 = time.perf_counter()
        sim.run(c, wait=False)
        t1 = time.perf_counter()
        sim.synchronize()
        t2 = time.perf_counter()
        cand = ((t2 - t0) * 1e6, (t1 - t0) * 1e6)
        every.append(round(cand[0], 1))
        best = cand if best is None or cand[0] < best[0] else best
    rows.append({"steps": k, "total_us": best[0], "enqueue_us": best[1], "us_per_step": best[0] / k, "all_us": every})
out["rows"] = rows
print(json.dumps(out, indent=1))
