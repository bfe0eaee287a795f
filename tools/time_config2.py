"""Timing of BASELINE configs[1]: 64^3 ferromagnet, 32 temperatures x 2 replicas."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peapods_amd import IsingSimulation
shape = (64, 64, 64)
t0 = time.perf_counter()
sim = IsingSimulation(list(shape), np.ones(shape + (3,), np.float32), np.linspace(4.111, 4.911, 32).astype(np.float32), 2, None, 42)
print("create %.2f s" % (time.perf_counter() - t0))
for label, kw in (("sweep only", dict(warmup_ratio=1.0)), ("sweep + measurement", dict(warmup_ratio=0.0)),
                  ("sweep + measurement + PT", dict(warmup_ratio=0.0, pt_interval=1))):
    cfg = sim.make_config(50, "metropolis", **kw)
    sim.run(cfg)
    times = []
    for _ in range(4):
        t0 = time.perf_counter(); sim.run(cfg); times.append((time.perf_counter() - t0) / 50 * 1e6)
    flips = 64**3 * 64
    print("64^3 x 64 systems, %s: %s us/sweep, best %.2f Gflips/s" % (label, [round(t) for t in times], flips / min(times) / 1e3))
one = IsingSimulation(list(shape), np.ones(shape + (3,), np.float32), np.linspace(4.111, 4.911, 32).astype(np.float32), 2, None, 42, stream_groups=1)
print(one.profile_step(one.make_config(10, "metropolis", warmup_ratio=0.0, pt_interval=1), 10))
