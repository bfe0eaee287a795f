"""Diagnostic: wall time per call of the phase-level entry points on the bench ensemble."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sim, *_ = bench.build_ensemble(0, 128, 1)
def t(fn, n=200):
    fn(); sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    sim.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
import ctypes as C
from peapods_amd import _native as nat
lib = nat.lib()
print("metropolis_sweep x1 (k_rng + k_sweep, no energy): %.1f us" % (t(lambda: sim.metropolis_sweep(1), 100)))
print("metropolis_sweep x20 per sweep: %.1f us" % (t(lambda: sim.metropolis_sweep(20), 10) / 20))
print("compute_energies (k_sweep energy only): %.1f us" % t(sim.compute_energies, 100))
print("overlap_dots (k_overlap + copies): %.1f us" % t(sim.overlap_dots, 50))
