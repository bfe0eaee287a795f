#!/usr/bin/env python3
"""Race hunt at full size: the same bench ensemble sampled twice from the same seed must agree bit for bit
(result blocks, tempering counters, final spins) -- for c2, c4 (each overlap move) and c1."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def same(x, y, where):
    if isinstance(x, dict):
        assert x.keys() == y.keys(), where
        for k in x:
            same(x[k], y[k], f"{where}.{k}")
    elif isinstance(x, (list, tuple)):
        assert len(x) == len(y), where
        for k, (u, v) in enumerate(zip(x, y)):
            same(u, v, f"{where}[{k}]")
    else:
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), f"{where} differs"


def one(config, n_sweeps, **sample):
    work = bench.WORKLOADS[config]
    outs = []
    for _ in range(2):
        sim, *_ = bench.build_ensemble(work, 0, work["n_disorder"], 1, 0)
        kw = dict(warmup_ratio=0.0, pt_interval=1)
        kw.update(work.get("sample", {}))
        kw.update(sample)
        res = sim.sample(n_sweeps, kw.pop("sweep_mode", "metropolis"), **kw)
        spins = np.concatenate([sim.get_spins(r) for r in range(0, work["n_disorder"], max(1, work["n_disorder"] // 8))])
        outs.append((res, spins))
        del sim
    (a, sa), (b, sb) = outs
    assert np.array_equal(sa, sb), f"{config} {sample}: final spins differ"
    same(a, b, f"{config} {sample}")
    print(f"{config} {sample or ''}: {n_sweeps} sweeps twice, identical", flush=True)


if __name__ == "__main__":
    one("c2", 3000)
    one("c2", 1500, sweep_mode="gibbs")
    one("c2", 1500, pt_schedule="full_ladder") if "full_ladder" in sys.argv else None
    for mode in ("houdayer", "jorg", "cmr"):
        one("c4", 400, overlap_cluster_build_mode=mode, overlap_cluster_mode="wolff")
    one("c1", 600)
