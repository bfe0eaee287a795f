"""Diagnostic: when every slab of one k_sweep_slab launch (bench config c1) passes each level barrier (s_memrealtime)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from peapods_amd import _native as nat
sim, *_ = bench.build_ensemble(bench.WORKLOADS["c1"], 0, 1, 1, 0)
lib = nat.lib()
lib.pa_debug_slab_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
n_levels = 3 * 63 + 1
cap_blocks = 8 * 64
out = np.zeros((cap_blocks, n_levels + 4), dtype=np.uint64)
sim.run(sim.make_config(20, "metropolis", pt_interval=1, warmup_ratio=0.0))
for rep in range(2):
    out[:] = 0
    n = lib.pa_debug_slab_stamps(sim._handle, nat.ptr(out), out.size, 3)
    if n <= 0:
        raise SystemExit(f"pa_debug_slab_stamps returned {n}")
    K = n // 8
    print(f"rep {rep}: {n} workgroup slots, {K} slabs per word")
    live = [b for b in range(n) if out[b, 1] >= out[b, 0] and out[b, 2 + int(out[b, 0])] != 0]
    t0 = min(int(out[b, 2 + int(out[b, 0])] & ((1 << 63) - 1)) for b in live)
    for b in live:
        if b % 8 != 0:  # word 0 only
            continue
        first, last = int(out[b, 0]), int(out[b, 1])
        raw = out[b, 2 + first: 3 + last]
        redo = (raw >> np.uint64(63)).astype(np.int64)
        t = (raw & np.uint64((1 << 63) - 1)).astype(np.int64) * 10  # ns
        dt = np.diff(t)
        print(f"  slab {b // 8:2d}: levels {first:3d}..{last:3d}  start {(t[0] - t0 * 10) / 1e3:7.2f} us  end {(t[-1] - t0 * 10) / 1e3:7.2f} us  "
              f"re-reads {int(redo.sum()):3d}  per level: median {np.median(dt):5.0f} ns  p90 {np.percentile(dt, 90):5.0f}  max {dt.max():6.0f}  "
              f"first 8: {' '.join(str(int(x)) for x in dt[:8])}")
