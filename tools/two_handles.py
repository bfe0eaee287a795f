"""Diagnostic: two independent half-size handles driven by two host threads vs one full handle."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PEAPODS_DEVELOPER", "1")
os.environ.setdefault("PEAPODS_STREAMS", "1")
import bench
from peapods_amd import IsingSimulation
from peapods_amd.spin_models import generate_couplings, seed_material

def make(first, n):
    seq, dyn = seed_material(42)
    c = np.stack(generate_couplings("bimodal", bench.SHAPE, 3, 128, seq, realizations=range(first, first + n)))
    t = np.linspace(0.8, 1.4, 24).astype(np.float32)
    return IsingSimulation(list(bench.SHAPE), c, t, 4, None, dyn, first_realization=first)

def run(sim, n):
    cfg = sim.make_config(n, "metropolis", pt_interval=1, warmup_ratio=0.0)
    sim.run(cfg)

n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 300
full = make(0, 128); run(full, 20)
t0 = time.perf_counter(); run(full, n); t_full = time.perf_counter() - t0
parts = [make(k * 128 // n_parts, 128 // n_parts) for k in range(n_parts)]
for s in parts: run(s, 20)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(s, n)) for s in parts]
[t.start() for t in th]; [t.join() for t in th]
t_parts = time.perf_counter() - t0
print("one handle (128 realisations): %.1f us/sweep; %d handles on %d host threads: %.1f us/sweep" % (t_full / n * 1e6, n_parts, n_parts, t_parts / n * 1e6))
