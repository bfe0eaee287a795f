"""Equilibrium energies with and without the cluster moves (max |difference| over the ladder):
ferromagnet 16^2 around T_c with FK moves, 8^3 +-J glass with the overlap moves in their
statistics-collecting (union-find) forms.  Run with PYTHONPATH=. on a GPU box."""
import numpy as np
from peapods_amd import Ising
tc = 2.0 / np.log(1 + np.sqrt(2))
temps = np.linspace(tc - 0.3, tc + 0.3, 16).astype(np.float32)
def fm(**kw):
    m = Ising((16, 16), temperatures=temps, n_replicas=2, seed=42)
    m.sample(10000, sweep_mode="metropolis", pt_interval=1, warmup_ratio=0.25, **kw)
    return m.energies_avg
plain = fm()
for label, kw in [("sw", dict(cluster_update_interval=1, cluster_mode="sw")),
                  ("wolff", dict(cluster_update_interval=1, cluster_mode="wolff")),
                  ("wolff+stats", dict(cluster_update_interval=1, cluster_mode="wolff", collect_cluster_stats=True)),
                  ("sw+stats", dict(cluster_update_interval=1, cluster_mode="sw", collect_cluster_stats=True))]:
    e = fm(**kw)
    print("fm", label, np.max(np.abs(e - plain)).round(5))
t2 = np.linspace(0.8, 1.4, 12).astype(np.float32)
def glass(**kw):
    m = Ising((8, 8, 8), couplings="bimodal", temperatures=t2, n_replicas=4, n_disorder=25, seed=42)
    m.sample(10000, sweep_mode="metropolis", pt_interval=1, warmup_ratio=0.25, **kw)
    return m.energies_avg
p2 = glass()
for build in ("houdayer", "jorg", "cmr"):
    for inner in ("wolff", "sw"):
        e = glass(overlap_cluster_update_interval=1, overlap_cluster_build_mode=build, overlap_cluster_mode=inner, collect_cluster_stats=True)
        print("glass+stats", build, inner, np.max(np.abs(e - p2)).round(5))
