"""Experiment (verdict r3, item 6): one launch per Metropolis sweep with the sweep workgroups released by per-word counters
(k_stream_sweep_flagged, PEAPODS_DEVELOPER=1 PEAPODS_FLAGGED=1) against the two launches it replaces, on the bench ensemble
(16^3 x 96 systems x 128 realisations), sweep-only protocol (pa_sim_metropolis_sweep), one stream.  Also checks that both
leave the same spins."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PEAPODS_DEVELOPER"] = "1"
import bench  # noqa: E402

work = bench.WORKLOADS["c2"]


def run(flagged, n=400):
    if flagged:
        os.environ["PEAPODS_FLAGGED"] = "1"
    else:
        os.environ.pop("PEAPODS_FLAGGED", None)
    sim, *_ = bench.build_ensemble(work, 0, work["n_disorder"], 1, 0, stream_groups=1)
    sim.metropolis_sweep(50)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sim.metropolis_sweep(n)
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6, sim.get_spins(0).copy(), sim.get_spins(127).copy()


a, sa0, sa1 = run(False)
b, sb0, sb1 = run(True)
print(f"two launches (k_rng + k_sweep_hc): {a:.1f} us per sweep;  one flagged launch: {b:.1f} us per sweep")
print("same spins after 50 + 5 x 400 sweeps:", bool((sa0 == sb0).all() and (sa1 == sb1).all()))
