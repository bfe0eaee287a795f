#!/usr/bin/env python3
"""Times the overlap-cluster moves on BASELINE configs[4]: 12^3 +-J, 24 T x 4 replicas x 128
disorder, move every sweep, with and without the move (difference = cost of the move)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from peapods_amd import IsingSimulation  # noqa: E402

SHAPE = (12, 12, 12)
n_dis = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(0)
J = (2 * rng.integers(0, 2, size=(n_dis,) + SHAPE + (3,)) - 1).astype(np.float32)
temps = np.linspace(0.8, 1.4, 24).astype(np.float32)
sim = IsingSimulation(list(SHAPE), J, temps, 4, None, 42)
out = {}


def run(label, **kw):
    cfg = sim.make_config(steps, "metropolis", pt_interval=1, warmup_ratio=0.0, **kw)
    sim.run(sim.make_config(10, "metropolis", pt_interval=1, warmup_ratio=0.0, **kw))
    t0 = time.perf_counter()
    sim.run(cfg)
    out[label] = (time.perf_counter() - t0) / steps * 1e6


run("no move")
for mode in ("houdayer", "jorg", "cmr"):
    for inner in ("wolff", "sw"):
        run(f"{mode}/{inner}", overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
            overlap_cluster_mode=inner)
for mode in ("houdayer", "jorg", "cmr"):  # the same moves while collecting sizes + top-four (recorded sweeps)
    for inner in ("wolff", "sw"):
        run(f"{mode}/{inner} + statistics", overlap_cluster_update_interval=1, overlap_cluster_build_mode=mode,
            overlap_cluster_mode=inner, collect_cluster_stats=True)
run("houdayer/sw observed", overlap_cluster_update_interval=1, overlap_cluster_build_mode="houdayer",
    overlap_cluster_mode="sw", overlap_cluster_action="observe")
print(json.dumps({"us_per_step": out}, indent=1))

# ---- FK cluster updates on a ferromagnet (16^3, 32 temperatures around T_c, 2 replicas, 16 chains)
shape = (16, 16, 16)
Jf = np.ones((16,) + shape + (3,), np.float32)
sim = IsingSimulation(list(shape), Jf, np.linspace(4.111, 4.911, 32).astype(np.float32), 2, None, 42)
out = {}
run("fm no cluster update")
run("fm fk/sw", cluster_update_interval=1, cluster_mode="sw")
run("fm fk/wolff", cluster_update_interval=1, cluster_mode="wolff")
run("fm fk/sw + statistics", cluster_update_interval=1, cluster_mode="sw", collect_cluster_stats=True)
run("fm fk/wolff + statistics", cluster_update_interval=1, cluster_mode="wolff", collect_cluster_stats=True)
print(json.dumps({"fk_us_per_step": out}, indent=1))
