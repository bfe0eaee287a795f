"""Diagnostic: phase times of every k_sweep_hc workgroup of one launch on the bench ensemble (s_memrealtime)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peapods_amd import _native as nat
sim, *_ = bench.build_ensemble(bench.WORKLOADS["c2"], 0, 128, 1, 0)
lib = nat.lib()
lib.pa_debug_sweep_hc_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
out = np.zeros((384, 8), dtype=np.uint64)
sim.run(sim.make_config(50, "metropolis", pt_interval=1, warmup_ratio=0.0))
for rep in range(3):
    n = lib.pa_debug_sweep_hc_stamps(sim._handle, nat.ptr(out), 384)
    t = out[:n, :6].astype(np.int64) * 10  # ns
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)
    names = ["load", "passes", "writeback", "(hand-over)", "energy"]
    print(f"rep {rep}: {n} workgroups; kernel span {(t[:, 5].max() - t0) / 1e3:.1f} us; start spread {(t[:, 0].max() - t0) / 1e3:.1f} us")
    for k, name in enumerate(names):
        print(f"   {name:12s} min {d[:, k].min() / 1e3:6.2f}  median {np.median(d[:, k]) / 1e3:6.2f}  max {d[:, k].max() / 1e3:6.2f} us")
    total = t[:, 5] - t[:, 0]
    print(f"   per-workgroup total: min {total.min() / 1e3:.2f} median {np.median(total) / 1e3:.2f} max {total.max() / 1e3:.2f} us")
