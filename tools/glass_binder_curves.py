"""SG Binder curves of the 3D +-J glass for the overlap moves (the reference's crossing protocol)."""
import sys
import numpy as np
from peapods_amd import Ising

temps = np.linspace(0.8, 1.4, 12).astype(np.float32)
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for label, n_rep, extra in [
    ("none/2", 2, None),
    ("houdayer/2", 2, {}),
    ("houdayer/4", 4, {}),
    ("houd4 sw/4", 4, {"overlap_cluster_build_mode": "houd4", "overlap_cluster_mode": "sw"}),
    ("cmr sw/4", 4, {"overlap_cluster_build_mode": "cmr", "overlap_cluster_mode": "sw"}),
    ("cmr+houd4 sw/4", 4, {"overlap_cluster_build_mode": "cmr+houd4", "overlap_cluster_mode": "sw"}),
]:
    for L in (8,):
        m = Ising((L, L, L), couplings="bimodal", temperatures=temps, n_replicas=n_rep, n_disorder=25, seed=42)
        kw = {} if extra is None else dict(overlap_cluster_update_interval=1, **extra)
        m.sample(sweeps, sweep_mode="metropolis", pt_interval=1, warmup_ratio=0.25, **kw)
        print(f"{label:16s} L={L} sg_binder", np.round(m.sg_binder, 3), "q2", np.round(m.overlap2[[0, -1]], 4),
              "E", np.round(m.energies_avg[[0, -1]], 4))
