"""Diagnostic: phase stamps of one k_sweep workgroup on the bench ensemble."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peapods_amd import _native as nat
sim, *_ = bench.build_ensemble(0, 128, 1)
lib = nat.lib()
lib.pa_debug_sweep_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros(6, dtype=np.uint64)
for _ in range(3):
    lib.pa_debug_sweep_stamps(sim._handle, nat.ptr(out))
    d = np.diff(out.astype(np.int64))
    print("cycles: load", d[0], "passes", d[1], "writeback", d[2], "energy-loop", d[3], "reduce", d[4], "total", int(out[5] - out[0]))

lib.pa_debug_measure_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros(7, dtype=np.uint64)
for _ in range(3):
    lib.pa_debug_measure_stamps(sim._handle, nat.ptr(out))
    d = np.diff(out.astype(np.int64))
    print("measure cycles: load", d[0], "overlap", d[1], "stats", d[2], "pt", d[3], "permute", d[4], "swap+wb", d[5], "total", int(out[6] - out[0]))
