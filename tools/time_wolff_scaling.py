"""Single-cluster (depth-first) Joerg / CMR moves on 12^3 x 24 T x 4 replicas for 8 ... 128 disorder samples:
the cost per call barely depends on the ensemble size (830 us at 8 samples, 1.16 ms at 128): it is the longest
search of the call, one wave walking a cluster of up to N sites at ~0.45 us per visited site."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from peapods_amd import IsingSimulation
SHAPE=(12,12,12)
temps = np.linspace(0.8, 1.4, 24).astype(np.float32)
out={}
for n_dis in (8, 32, 64, 84, 128):
    rng = np.random.default_rng(0)
    J = (2 * rng.integers(0, 2, size=(n_dis,) + SHAPE + (3,)) - 1).astype(np.float32)
    sim = IsingSimulation(list(SHAPE), J, temps, 4, None, 42)
    def run(**kw):
        cfg = sim.make_config(100, "metropolis", pt_interval=1, warmup_ratio=0.0, **kw)
        sim.run(sim.make_config(200, "metropolis", pt_interval=1, warmup_ratio=0.0, **kw))
        t0=time.perf_counter(); sim.run(cfg); return (time.perf_counter()-t0)/100*1e6
    base = run()
    out[n_dis] = {"none": round(base,1)}
    for mode in ("jorg","cmr"):
        out[n_dis][mode] = round(run(overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode, overlap_cluster_mode="wolff"),1)
    del sim
print(json.dumps(out, indent=1))
