"""Multi-GPU `sample()` (peapods_amd.distributed): a disorder ensemble sharded over several handles gives the
reference's full result dictionary, bit-identical to one handle holding every realisation.

A GPU box has one device, so the ranks are threads of this process with their handles on the same GPU and the
exchange is `LoopbackGroup` -- everything above the transport (run, collect_parts, gather_parts in realisation
order, assemble_parts) is the code the RCCL path runs; the RCCL transport itself (pa_comm_gatherv) is exercised
with the one rank a single device allows.  More than one GPU: unmeasured on hardware (no such box here)."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ensemble(shape, n_dis, seed=42):
    from peapods_amd.spin_models import generate_couplings, seed_material

    seq, dyn = seed_material(seed)
    couplings = np.stack(generate_couplings("bimodal", shape, len(shape), n_dis, seq))
    return couplings, dyn


def _assert_same(a, b, path=""):
    assert type(a) is type(b) or (np.isscalar(a) and np.isscalar(b)), path
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()), path
        for k in a:
            _assert_same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _assert_same(x, y, f"{path}[{i}]")
    else:
        x, y = np.asarray(a), np.asarray(b)
        assert x.dtype == y.dtype and x.shape == y.shape, path
        np.testing.assert_array_equal(x, y, err_msg=path)


def _run_sharded(shape, couplings, temps, n_rep, dyn, world, n_sweeps, kwargs, root=None):
    from peapods_amd import IsingSimulation
    from peapods_amd import distributed as dist

    n_dis = couplings.shape[0]
    group = dist.LoopbackGroup(world)
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            first, stop = dist.shard_range(n_dis, rank, world)
            sim = IsingSimulation(list(shape), couplings[first:stop], temps, n_rep, None, dyn, first_realization=first)
            results[rank] = dist.sample(sim, group.endpoint(rank), n_sweeps, "metropolis", root=root, **kwargs)
        except BaseException as exc:  # a failed rank must not leave the others waiting at the barrier
            errors.append(exc)
            group._barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    if errors:
        raise errors[0]
    return results


CONFIGS = [
    # shape, n_temps, n_replicas, n_disorder, world, sample kwargs
    ((6, 6, 6), 5, 4, 5, 2, dict(pt_interval=1, warmup_ratio=0.25)),                                # uneven: 2 + 3
    ((6, 6, 6), 4, 2, 5, 3, dict(pt_interval=2, pt_schedule="full_ladder", warmup_ratio=0.0,        # 1 + 2 + 2
                                 autocorrelation_max_lag=4, equilibration_diagnostic=True)),
    ((4, 4, 4), 3, 4, 4, 2, dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode="cmr",
                                 overlap_cluster_mode="sw", collect_cluster_stats=True, snapshot_interval=2,
                                 cluster_update_interval=3, warmup_ratio=0.25)),
    ((8, 8), 4, 2, 3, 2, dict(overlap_cluster_update_interval=1, overlap_cluster_build_mode="houdayer",
                              overlap_cluster_mode="sw", overlap_cluster_action="observe", warmup_ratio=0.0)),
    ((4, 4, 4), 3, 1, 3, 3, dict(warmup_ratio=0.5)),                                                # no replica pairs
]


@pytest.mark.parametrize("shape,n_temps,n_rep,n_dis,world,kwargs", CONFIGS)
def test_sharded_sample_equals_one_handle_on_every_key(shape, n_temps, n_rep, n_dis, world, kwargs):
    from peapods_amd import IsingSimulation

    couplings, dyn = _ensemble(shape, n_dis)
    temps = np.linspace(0.9, 2.2, n_temps).astype(np.float32)
    n_sweeps = 24 if "equilibration_diagnostic" not in kwargs else 300
    whole = IsingSimulation(list(shape), couplings, temps, n_rep, None, dyn).sample(n_sweeps, "metropolis", **kwargs)
    for got in _run_sharded(shape, couplings, temps, n_rep, dyn, world, n_sweeps, kwargs):
        _assert_same(got, whole)  # every rank holds the full dictionary: same keys, order, dtypes, shapes, bits


def test_sharded_sample_matches_the_oracle_and_honours_root(oracle):
    from peapods_amd import IsingSimulation  # noqa: F401

    shape, n_temps, n_rep, n_dis = (6, 6, 6), 4, 2, 5
    couplings, dyn = _ensemble(shape, n_dis, seed=7)
    temps = np.linspace(0.8, 1.6, n_temps).astype(np.float32)
    kwargs = dict(pt_interval=1, warmup_ratio=0.25)
    results = _run_sharded(shape, couplings, temps, n_rep, dyn, 2, 20, kwargs, root=1)
    assert results[0] is None                     # only the root assembles (and receives) the result
    want = oracle.OracleSimulation(list(shape), couplings, temps, n_rep, None, dyn).sample(20, "metropolis", **kwargs)
    got = results[1]
    assert set(got) == set(want)
    for key in want:
        if key == "per_disorder":
            for k2, v in want[key]["parallel_tempering"].items():
                np.testing.assert_array_equal(got[key]["parallel_tempering"][k2], v, err_msg=k2)
        elif key == "overlap_histogram":
            np.testing.assert_array_equal(np.stack(got[key]), np.stack(want[key]))
        else:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_ranks_with_different_arguments_fail_together():
    """gather_parts compares a digest of what every rank is about to contribute, so a mismatch raises on every
    rank instead of hanging in a collective of different sizes."""
    from peapods_amd import distributed as dist

    group = dist.LoopbackGroup(2)
    errors = [None, None]

    def rank_main(rank):
        parts = {"blocks": np.zeros((2, 4 + rank))}  # different row shapes
        try:
            dist.gather_parts(parts, group.endpoint(rank))
        except RuntimeError as exc:
            errors[rank] = exc

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    assert all(isinstance(e, RuntimeError) and "different result arrays" in str(e) for e in errors)


def test_rccl_gatherv_and_distributed_sample_with_one_rank(oracle):
    """The RCCL transport of the same path with the one rank a single device allows: pa_comm_gatherv to every rank
    and to a root, empty contributions, and distributed.sample over a `Comm`."""
    from peapods_amd import IsingSimulation
    from peapods_amd import distributed as dist

    comm = dist.Comm(0, 1, 0)
    try:
        x = (np.arange(30, dtype=np.uint64) * 7).reshape(5, 3, 2)
        np.testing.assert_array_equal(comm.gather_rows(x), x)
        np.testing.assert_array_equal(comm.gather_rows(x, root=0), x)
        assert comm.gather_rows(x[:0]).shape == (0, 3, 2)
        shape, n_rep = (4, 4, 4), 2
        temps = np.linspace(1.0, 2.0, 3).astype(np.float32)
        couplings, dyn = _ensemble(shape, 3, seed=5)
        kwargs = dict(pt_interval=1, warmup_ratio=0.25)
        got = dist.sample(IsingSimulation(list(shape), couplings, temps, n_rep, None, dyn), comm, 12, "metropolis", **kwargs)
        want = IsingSimulation(list(shape), couplings, temps, n_rep, None, dyn).sample(12, "metropolis", **kwargs)
        _assert_same(got, want)
    finally:
        comm.close()


def test_ising_with_comm_shards_the_disorder_and_returns_the_whole_result():
    """`Ising(..., comm=...)`: the user-facing class on several ranks -- each builds only its shard of the couplings
    (same SeedSequence children as the unsharded run) and every rank ends with the observables of all realisations."""
    from peapods_amd import Ising
    from peapods_amd import distributed as dist

    args = dict(couplings="bimodal", temperatures=np.linspace(0.9, 1.8, 4), n_replicas=2, n_disorder=5, seed=11)
    kwargs = dict(pt_interval=1, warmup_ratio=0.25)
    whole = Ising((6, 6, 6), **args)
    want = whole.sample(16, **kwargs)
    group = dist.LoopbackGroup(2)
    models, errors = [None, None], []

    def rank_main(rank):
        try:
            models[rank] = Ising((6, 6, 6), comm=group.endpoint(rank), **args)
            models[rank].result = models[rank].sample(16, **kwargs)
        except BaseException as exc:
            errors.append(exc)
            group._barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    if errors:
        raise errors[0]
    assert models[0].couplings.shape[0] == 2 and models[1].couplings.shape[0] == 3
    for model in models:
        _assert_same(model.result, want)
        np.testing.assert_array_equal(model.sg_binder, whole.sg_binder)
        np.testing.assert_array_equal(model.heat_capacity, whole.heat_capacity)
