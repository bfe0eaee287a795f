#!/usr/bin/env python3
"""Per-sweep wall time of small, launch-bound configurations (the sizes of the reference's own
physics scripts): where the step time is kernel-launch overhead rather than kernel time."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from peapods_amd import Ising  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
out = {}


def timed(label, model, **kw):
    model.sample(200, warmup_ratio=0.25, **kw)
    t0 = time.perf_counter()
    model.sample(sweeps, warmup_ratio=0.25, **kw)
    out[label] = round((time.perf_counter() - t0) / sweeps * 1e6, 2)


tc = 2.0 / np.log(1 + np.sqrt(2))
timed("C1 32^2 8T metropolis", Ising((32, 32), temperatures=np.linspace(tc - 0.3, tc + 0.3, 8).astype(np.float32), seed=1))
timed("32^2 32T x2 metropolis+pt", Ising((32, 32), temperatures=np.linspace(tc - 0.3, tc + 0.3, 32).astype(np.float32),
                                        n_replicas=2, seed=1), pt_interval=1)
timed("32^2 32T x2 metropolis+pt+fk sw", Ising((32, 32), temperatures=np.linspace(tc - 0.3, tc + 0.3, 32).astype(np.float32),
                                              n_replicas=2, seed=1), pt_interval=1, cluster_update_interval=1, cluster_mode="sw")
timed("fcc 10^3 24T x2 metropolis+pt+fk sw", Ising((10, 10, 10), temperatures=np.linspace(9.2, 10.4, 24).astype(np.float32),
                                                  n_replicas=2, geometry="fcc", seed=1),
      pt_interval=1, cluster_update_interval=1, cluster_mode="sw")
t2 = np.linspace(0.8, 1.4, 12).astype(np.float32)
timed("8^3 +-J 12T x2 x25 metropolis+pt+houdayer", Ising((8, 8, 8), couplings="bimodal", temperatures=t2, n_replicas=2,
                                                        n_disorder=25, seed=1), pt_interval=1, overlap_cluster_update_interval=1)
timed("8^3 +-J 12T x2 x25 metropolis+pt+jorg sw", Ising((8, 8, 8), couplings="bimodal", temperatures=t2, n_replicas=2,
                                                       n_disorder=25, seed=1), pt_interval=1, overlap_cluster_update_interval=1,
      overlap_cluster_build_mode="jorg", overlap_cluster_mode="sw")
timed("8^3 gaussian 1T x2 x64 metropolis+pt+houdayer", Ising((8, 8, 8), couplings="gaussian", temperatures=np.array([1.4], np.float32),
                                                            n_replicas=2, n_disorder=64, seed=1),
      pt_interval=1, overlap_cluster_update_interval=1)
print(json.dumps({"us_per_sweep": out}, indent=1))
