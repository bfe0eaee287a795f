#!/usr/bin/env python3
"""Time series of consecutive K-step sample() calls right after start-up (the driver's format: K = 20): how long the
per-call time takes to settle, and what the host's part of each call is.  argv: [K] [calls] [sleep_us between calls]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pause = float(sys.argv[3]) * 1e-6 if len(sys.argv) > 3 else 0.0
work = bench.WORKLOADS["c2"]
sim, *_ = bench.build_ensemble(work, 0, 128, 1, 0)
cfg = sim.make_config(k, "metropolis", warmup_ratio=0.0, pt_interval=1)
sim.run(sim.make_config(5, "metropolis", warmup_ratio=0.0, pt_interval=1))
total, enq = [], []
for _ in range(calls):
    sim.synchronize()
    if pause:
        time.sleep(pause)
    t0 = time.perf_counter()
    sim.run(cfg, wait=False)
    t1 = time.perf_counter()
    sim.synchronize()
    t2 = time.perf_counter()
    total.append(round((t2 - t0) * 1e6 / k, 1))
    enq.append(round((t1 - t0) * 1e6 / k, 1))
print(json.dumps({"steps": k, "pause_us": pause * 1e6, "us_per_step": total, "enqueue_us_per_step": enq}))
