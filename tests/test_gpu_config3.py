"""BASELINE configs[3] at its own size, rehearsed on one GPU: 16^3 +-J, 24 temperatures x 4 replicas, 1024 disorder
realisations in 8 shards of 128 (global indices 0..1023), the full-result gather to the root at its real payload
(3 x 1024 x 24 x 4097 x 8 B = 2.4 GB of histograms and conditional sums).

A GPU box has one device, so the 8 ranks are threads with their handles on the same GPU and the exchange is
`LoopbackGroup`: everything above the transport -- shard ranges, seeds by global realisation index, run,
collect_parts, gather_parts in realisation order, the reference's aggregation (simulation/mod.rs:887-921,
statistics/results.rs:165-282) -- is the code the 8-GPU run executes.  The RCCL transport takes the same payload
with the one rank a single device allows (pa_comm_gatherv's device staging of an 805 MB array)."""

import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPE = (16, 16, 16)
N = 4096
N_TEMPS, N_REP = 24, 4
N_DIS, WORLD = 1024, 8
N_SWEEPS = 12
KW = dict(pt_interval=1, pt_schedule="single_random_edge", warmup_ratio=0.25)


def _shard_couplings(first, stop):
    from peapods_amd.spin_models import generate_couplings, seed_material

    seq, dyn = seed_material(42)
    return np.stack(generate_couplings("bimodal", SHAPE, 3, N_DIS, seq, realizations=range(first, stop))), dyn


def _temps():
    return np.linspace(0.8, 1.4, N_TEMPS).astype(np.float32)


def _assert_same(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), path
        for k in a:
            _assert_same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _assert_same(x, y, f"{path}[{i}]")
    else:
        x, y = np.asarray(a), np.asarray(b)
        assert x.dtype == y.dtype and x.shape == y.shape, path
        assert np.array_equal(x, y), path  # (no assert_array_equal: its mismatch report copies the 805 MB operands)


def test_config3_eight_shards_of_128_realisations(oracle):
    psutil = pytest.importorskip("psutil")
    from peapods_amd import IsingSimulation, _core
    from peapods_amd import _native as nat
    from peapods_amd import distributed as dist

    need = 14 << 30  # two assembled results (2.4 GB each) + the parts they are concatenated from
    if psutil.virtual_memory().available < need:
        pytest.fail(f"configs[3] rehearsal needs {need >> 30} GiB of free host memory")

    temps = _temps()
    group = dist.LoopbackGroup(WORLD)
    sims, results, errors, timing = [None] * WORLD, [None] * WORLD, [], {}

    def rank_main(rank):
        try:
            first, stop = dist.shard_range(N_DIS, rank, WORLD)
            assert (first, stop) == (128 * rank, 128 * (rank + 1))
            couplings, dyn = _shard_couplings(first, stop)
            sim = IsingSimulation(list(SHAPE), couplings, temps, N_REP, None, dyn, first_realization=first)
            sims[rank] = (sim, couplings, dyn)
            comm = group.endpoint(rank)
            cfg = sim.make_config(N_SWEEPS, "metropolis", **KW)
            sim.run(cfg)
            parts, meta = sim.collect_parts(cfg, True)
            comm.barrier()
            t0 = time.perf_counter()
            gathered = dist.gather_parts(parts, comm, root=0)
            comm.barrier()
            if rank == 0:
                timing["gather_ms"] = (time.perf_counter() - t0) * 1e3
                timing["gather_bytes"] = int(sum(a.nbytes for a in gathered.values()))
                results[0] = _core.assemble_parts(sim._lib, gathered, meta)
            else:
                assert gathered is None
        except BaseException as exc:  # a failed rank must not leave the others waiting at the barrier
            errors.append(exc)
            group._barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(900)
    if errors:
        raise errors[0]
    got = results[0]

    # (1) the payload is the one DESIGN section 6 states: three [1024][24][4097] 8-byte arrays + blocks + counters
    hist_bytes = 3 * N_DIS * N_TEMPS * (N + 1) * 8
    assert hist_bytes <= timing["gather_bytes"] < hist_bytes + (8 << 20)
    print(f"\nconfigs[3] rehearsal: root gathered {timing['gather_bytes']} bytes from {WORLD} shards in "
          f"{timing['gather_ms']:.0f} ms (threads of one process)")
    assert got["per_sample_overlap_histogram"].shape == (N_DIS, N_TEMPS, N + 1)
    pt = got["per_disorder"]["parallel_tempering"]
    assert pt["edge_attempts"].shape == (N_DIS, N_TEMPS - 1) and pt["round_trips"].shape == (N_DIS, N_REP, N_TEMPS)
    assert np.all(pt["edge_attempts"].sum(axis=1) == N_SWEEPS * N_REP)
    n_rec = N_SWEEPS - int(np.floor(N_SWEEPS * 0.25 + 0.5))
    assert np.all(got["per_sample_overlap_histogram"].sum(axis=2) == n_rec * (N_REP // 2))

    # (2) realisations beyond the first shard, each against an oracle run of that realisation alone (seeds by global
    # index: src/lib.rs:30-32): first of shard 1, a middle one, first of the last shard, the very last
    for g in (128, 511, 896, 1023):
        rank, local = divmod(g, 128)
        sim, couplings, dyn = sims[rank]
        cpu = oracle.OracleSimulation(list(SHAPE), couplings[local:local + 1], temps, N_REP, None, dyn, first_realization=g)
        want = cpu.sample(N_SWEEPS, "metropolis", **KW)
        np.testing.assert_array_equal(sim.get_spins(local), cpu.get_spins(0))
        np.testing.assert_array_equal(sim.get_system_ids(local), cpu.get_system_ids(0))
        np.testing.assert_array_equal(sim.get_energies(local), cpu.get_energies(0))
        for s in range(N_TEMPS * N_REP):
            np.testing.assert_array_equal(sim.get_rng_state(local, s), cpu.get_rng_state(0, s))
        np.testing.assert_array_equal(sim.result_blocks()[local], cpu.result_blocks()[0])
        np.testing.assert_array_equal(got["per_sample_overlap_histogram"][g], np.stack(want["overlap_histogram"]))
        np.testing.assert_array_equal(got["per_sample_ql_at_q_sum"][g], want["ql_at_q_sum"])
        np.testing.assert_array_equal(got["per_sample_ql2_at_q_sum"][g], want["ql2_at_q_sum"])
        for key in pt:
            np.testing.assert_array_equal(pt[key][g], want["per_disorder"]["parallel_tempering"][key][0], err_msg=key)
    del sims[:], sim, cpu

    # (3) the same 1024 realisations as 8 handles run one after another, their parts concatenated by hand and aggregated
    # once: the dictionary the sharded run assembled on its root must be this one, bit for bit
    chunks, meta = [], None
    for rank in range(WORLD):
        couplings, dyn = _shard_couplings(128 * rank, 128 * (rank + 1))
        one = IsingSimulation(list(SHAPE), couplings, temps, N_REP, None, dyn, first_realization=128 * rank)
        cfg = one.make_config(N_SWEEPS, "metropolis", **KW)
        one.run(cfg)
        parts, meta = one.collect_parts(cfg, True)
        chunks.append(parts)
        del one
    whole = {k: np.concatenate([c[k] for c in chunks], axis=0) for k in chunks[0]}
    del chunks
    want = _core.assemble_parts(nat.lib(), whole, meta)
    _assert_same(got, want)
    del want, whole

    # (4) the RCCL transport with this payload: the 805 MB histogram array through pa_comm_gatherv's device staging
    # (to a root and to every rank), with the one rank a single device allows
    comm = dist.Comm(0, 1, 0)
    try:
        hist = got["per_sample_overlap_histogram"]
        t0 = time.perf_counter()
        back = comm.gather_rows(hist, root=0)
        ms = (time.perf_counter() - t0) * 1e3
        assert back.nbytes == N_DIS * N_TEMPS * (N + 1) * 8 and np.array_equal(back, hist)
        print(f"configs[3] rehearsal: pa_comm_gatherv of {back.nbytes} bytes (one RCCL rank, host -> device -> host) {ms:.0f} ms")
        del back
        assert np.array_equal(comm.gather_rows(got["per_sample_ql_at_q_sum"]), got["per_sample_ql_at_q_sum"])
    finally:
        comm.close()

