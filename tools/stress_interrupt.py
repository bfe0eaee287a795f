#!/usr/bin/env python3
"""Repeats the cancel test with random delays and prints what happened each time (diagnosis)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402
from peapods_amd import IsingSimulation  # noqa: E402

rng = np.random.default_rng(0)
shape, n_rep = (6, 6, 6), 2
temps = np.linspace(1.0, 2.0, 4).astype(np.float32)
J = (2 * rng.integers(0, 2, size=(2,) + shape + (3,)) - 1).astype(np.float32)
kw = dict(pt_interval=1, warmup_ratio=0.0)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    gpu = IsingSimulation(list(shape), J, temps, n_rep, None, 9 + trial)
    cpu = ob.OracleSimulation(list(shape), J, temps, n_rep, None, 9 + trial)
    delay = float(rng.uniform(0.02, 0.5))
    if len(sys.argv) > 2 and sys.argv[2] == "sigint":
        import signal

        timer = threading.Timer(delay, lambda: signal.raise_signal(signal.SIGINT))
    else:
        timer = threading.Timer(delay, lambda: gpu._lib.pa_sim_request_interrupt(gpu._handle))
    t0 = time.perf_counter()
    timer.start()
    try:
        gpu.run(gpu.make_config(4_000_000, "metropolis", **kw))
        print("not interrupted?!")
    except KeyboardInterrupt:
        pass
    took = time.perf_counter() - t0
    ran = int(gpu._lib.pa_sim_sweeps_completed(gpu._handle))
    ref = IsingSimulation(list(shape), J, temps, n_rep, None, 9 + trial)
    ref.run(ref.make_config(ran, "metropolis", **kw))
    same_gpu = all(np.array_equal(gpu.get_spins(r), ref.get_spins(r)) for r in range(2))
    cpu2 = ob.OracleSimulation(list(shape), J, temps, n_rep, None, 9 + trial)
    cpu2.run(cpu2.make_config(ran, "metropolis", **kw))
    same_cpu = all(np.array_equal(gpu.get_spins(r), cpu2.get_spins(r)) for r in range(2))
    print(f"   library says {ran} sweeps; equal to an uninterrupted GPU run of that length: {same_gpu}; to the oracle: {same_cpu}")
    found = None
    for chunk in range(20000):
        if all(np.array_equal(gpu.get_spins(r), cpu.get_spins(r)) for r in range(2)):
            found = 32 * chunk
            break
        cpu.run(cpu.make_config(32, "metropolis", **kw))
    print(f"trial {trial}: delay {delay:.3f} s, returned after {took:.3f} s, stopped after {found} sweeps", flush=True)
