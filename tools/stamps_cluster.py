"""Diagnostic: where a union-find cluster move (k_cluster_sw) spends its time, phase by phase and task by task, on the
bench ensemble of BASELINE configs[4] (12^3, 24 temperatures x 4 replicas, 128 realisations, one stream group).
   python tools/stamps_cluster.py [houdayer|jorg|cmr] [sw|wolff]
Every workgroup (= one pair of one realisation) stamps s_memrealtime (100 MHz) at its phase boundaries
(pa_debug_cluster_stamps, include/peapods_amd_debug.h)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peapods_amd import _native as nat  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "jorg"
variant = sys.argv[2] if len(sys.argv) > 2 else "sw"
work = bench.WORKLOADS["c4"]
sim, *_ = bench.build_ensemble(work, 0, work["n_disorder"], 1, 0, stream_groups=1)
kw = dict(work["sample"], pt_interval=1, overlap_cluster_build_mode=mode, overlap_cluster_mode=variant)
sim.run(sim.make_config(100, "metropolis", warmup_ratio=0.0, **kw))
lib = nat.lib()
lib.pa_debug_cluster_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
n_blocks = lib.pa_debug_cluster_stamps(sim._handle, 1, None, 0)
sim.run(sim.make_config(8, "metropolis", warmup_ratio=0.0, **kw))
out = np.zeros((n_blocks, 16), dtype=np.uint64)
got = lib.pa_debug_cluster_stamps(sim._handle, 0, nat.ptr(out), n_blocks)
a = out[:got].astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print(f"{mode}/{variant}: {len(a)} workgroups, launch span {(a[:, 11].max() - t0) / 100:.1f} us, "
      f"mean per workgroup {(a[:, 11] - a[:, 0]).mean() / 100:.2f} us")
order = [0, 1, 2, 3, 12, 4, 5, 6, 7, 8, 9, 11]
names = ["bitmaps of the pair", "eligible bonds", "bond draws", "labels by runs", "unions (packed list)", "flatten",
         "report", "roots that draw", "cluster draws", "flips", "generator state, write-back"]
for k, name in enumerate(names):
    d = (a[:, order[k + 1]] - a[:, order[k]]) / 100.0
    print(f"  {name:28s} mean {d.mean():6.2f} us   p90 {np.percentile(d, 90):6.2f}")
print("  start of the workgroups after the first (quartiles, us):", np.percentile(a[:, 0] - t0, [25, 50, 75, 100]) / 100)
