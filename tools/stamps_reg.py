"""Diagnostic: phase times of every k_sweep_reg wave of one launch on the bench ensemble (s_memrealtime)."""
import ctypes as C, os, sys
os.environ.setdefault("PEAPODS_DEVELOPER", "1")  # the register-resident sweep is opt-in
os.environ.setdefault("PEAPODS_REG", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peapods_amd import _native as nat
sim, *_ = bench.build_ensemble(bench.WORKLOADS["c2"], 0, 128, 1, 0)
lib = nat.lib()
lib.pa_debug_sweep_reg_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
out = np.zeros((384, 8), dtype=np.uint64)
sim.run(sim.make_config(50, "metropolis", pt_interval=1, warmup_ratio=0.0))
names = ["spins", "planes wait", "levels 0-15", "levels 16-31", "levels 32-47", "store", "energy"]
for rep in range(3):
    n = lib.pa_debug_sweep_reg_stamps(sim._handle, nat.ptr(out), 384)
    t = out[:n].astype(np.int64) * 10  # ns
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)
    print(f"rep {rep}: {n} waves; kernel span {(t[:, 7].max() - t0) / 1e3:.1f} us; start spread {(t[:, 0].max() - t0) / 1e3:.1f} us")
    for k, name in enumerate(names):
        print(f"   {name:14s} min {d[:, k].min() / 1e3:6.2f}  median {np.median(d[:, k]) / 1e3:6.2f}  max {d[:, k].max() / 1e3:6.2f} us")
    total = t[:, 7] - t[:, 0]
    print(f"   per-wave total: min {total.min() / 1e3:.2f} median {np.median(total) / 1e3:.2f} max {total.max() / 1e3:.2f} us")
