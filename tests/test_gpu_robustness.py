"""GPU robustness checks around the hot path: sharded handles with snapshots, cooperative cancel, histogram
accumulators across calls (they are never memset), and a seed soak of the kernels whose correctness rests on
inter-workgroup protocols (the slab pipeline, the small-block random stream, the fused overlap launch).
Everything is compared with the oracle bit for bit."""

import signal
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bimodal(shape, n_dis, seed):
    rng = np.random.default_rng(seed)
    return (2 * rng.integers(0, 2, size=(n_dis,) + tuple(shape) + (len(shape),)) - 1).astype(np.float32)


@pytest.mark.parametrize("build", ["jorg", "cmr", "houdayer"])
def test_sharded_handles_with_snapshots_match_one_handle(oracle, build):
    """Snapshots (and the union-find form of the single-cluster moves they force) belong to GLOBAL realisation
    0 only (mod.rs:302, overlap.rs:376-380): a handle that holds a later shard takes none and keeps its first
    realisation on the search path, so the per-realisation trajectories do not depend on the sharding."""
    from peapods_amd import IsingSimulation

    shape, n_dis, n_rep = (6, 6, 6), 4, 4
    temps = np.linspace(0.8, 1.6, 5).astype(np.float32)
    J = bimodal(shape, n_dis, 11)
    kw = dict(pt_interval=1, warmup_ratio=0.25, overlap_cluster_update_interval=2, overlap_cluster_build_mode=build,
              overlap_cluster_mode="wolff", snapshot_interval=4)
    whole = IsingSimulation(list(shape), J, temps, n_rep, None, 5)
    cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 5)
    out_whole = whole.sample(16, "metropolis", **kw)
    cpu.sample(16, "metropolis", **kw)
    lo = IsingSimulation(list(shape), J[:2], temps, n_rep, None, 5, first_realization=0)
    hi = IsingSimulation(list(shape), J[2:], temps, n_rep, None, 5, first_realization=2)
    out_lo = lo.sample(16, "metropolis", **kw)
    out_hi = hi.sample(16, "metropolis", **kw)
    assert "cluster_snapshots" in out_whole and "cluster_snapshots" in out_lo
    assert "cluster_snapshots" not in out_hi
    for r in range(n_dis):
        part, local = (lo, r) if r < 2 else (hi, r - 2)
        np.testing.assert_array_equal(part.get_spins(local), cpu.get_spins(r))
        np.testing.assert_array_equal(part.get_spins(local), whole.get_spins(r))
        np.testing.assert_array_equal(part.get_system_ids(local), whole.get_system_ids(r))
        for s in range(whole.n_systems):
            np.testing.assert_array_equal(part.get_rng_state(local, s), whole.get_rng_state(r, s))


@pytest.mark.parametrize("how", ["request", "sigint"])
def test_interrupted_sample_stops_early_and_leaves_a_replayable_state(oracle, how):
    """src/lib.rs:304-333: a cancel request ends the run with KeyboardInterrupt("interrupted"); the state is the
    one after the last sweep that ran, and sampling continues from it exactly as the reference would."""
    from peapods_amd import IsingSimulation

    shape, n_rep = (6, 6, 6), 2
    temps = np.linspace(1.0, 2.0, 4).astype(np.float32)
    J = bimodal(shape, 2, 3)
    gpu = IsingSimulation(list(shape), J, temps, n_rep, None, 9)
    cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 9)
    kw = dict(pt_interval=1, warmup_ratio=0.0)
    n_asked = 4_000_000  # minutes of work if it were not interrupted
    if how == "request":
        timer = threading.Timer(0.3, lambda: gpu._lib.pa_sim_request_interrupt(gpu._handle))
    else:
        timer = threading.Timer(0.3, lambda: signal.raise_signal(signal.SIGINT))
    t0 = time.perf_counter()
    timer.start()
    with pytest.raises(KeyboardInterrupt):
        gpu.run(gpu.make_config(n_asked, "metropolis", **kw))
    took = time.perf_counter() - t0
    timer.cancel()
    assert took < 30.0, "the request was not honoured while work was queued"
    n_ran = int(gpu._lib.pa_sim_sweeps_completed(gpu._handle))
    assert 0 < n_ran < n_asked and n_ran % 32 == 0
    cpu.run(cpu.make_config(n_ran, "metropolis", **kw))  # the reference's state after that many sweeps
    for r in range(2):
        np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))
    for r in range(2):
        np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r))
        for s in range(gpu.n_systems):
            np.testing.assert_array_equal(gpu.get_rng_state(r, s), cpu.get_rng_state(r, s))
    # ... and it continues like the reference
    a = gpu.sample(24, "metropolis", pt_interval=1, warmup_ratio=0.25)
    b = cpu.sample(24, "metropolis", pt_interval=1, warmup_ratio=0.25)
    for key in ("energies", "mags2", "overlap2", "link_overlap"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    for r in range(2):
        np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))


def test_histograms_restart_with_every_sample_call(oracle):
    """P(q) and the conditional link-overlap sums restart with every sample() (OverlapAccum::new per run,
    overlap.rs:206-249) although the device arrays are never cleared: bins the second run does not reach must
    read zero, bins it reaches must not carry the first run's counts."""
    from peapods_amd import IsingSimulation

    shape, n_rep = (4, 4, 4), 4
    temps = np.array([0.5, 1.0, 3.0, 8.0], dtype=np.float32)  # cold: few bins; hot: many
    J = bimodal(shape, 3, 21)
    gpu = IsingSimulation(list(shape), J, temps, n_rep, None, 2)
    cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 2)
    for n_sweeps, warm in ((40, 0.0), (6, 0.5), (25, 0.2)):
        a = gpu.sample(n_sweeps, "metropolis", pt_interval=1, warmup_ratio=warm)
        b = cpu.sample(n_sweeps, "metropolis", pt_interval=1, warmup_ratio=warm)
        np.testing.assert_array_equal(a["per_sample_overlap_histogram"], b["per_sample_overlap_histogram"])
        np.testing.assert_array_equal(a["per_sample_ql_at_q_sum"], b["per_sample_ql_at_q_sum"])
        np.testing.assert_array_equal(a["per_sample_ql2_at_q_sum"], b["per_sample_ql2_at_q_sum"])
        np.testing.assert_array_equal(a["ql_at_q_sum"], b["ql_at_q_sum"])
        assert int(np.sum(a["per_sample_overlap_histogram"])) == 3 * 4 * 2 * (n_sweeps - int(np.floor(n_sweeps * warm + 0.5)))


# ---- soak: many seeds through the kernels that hand data between workgroups ---------------------------------
def _soak(oracle, shape, n_temps, n_rep, n_dis, seeds, sweeps, kind="ferro", **kw):
    from peapods_amd import IsingSimulation

    temps = np.linspace(2.0, 5.0, n_temps).astype(np.float32)
    for seed in seeds:
        if kind == "ferro":
            J = np.ones((n_dis,) + tuple(shape) + (len(shape),), dtype=np.float32)
        else:
            J = bimodal(shape, n_dis, 1000 + seed)
        gpu = IsingSimulation(list(shape), J, temps, n_rep, None, seed)
        cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, seed)
        a = gpu.sample(sweeps, "metropolis", **kw)
        b = cpu.sample(sweeps, "metropolis", **kw)
        gpu.synchronize()  # reports slab_error / stream_error as PA_ERR_DEVICE
        for r in range(n_dis):
            np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r), err_msg=f"seed {seed}")
            np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r), err_msg=f"seed {seed}")
        for key in ("energies", "mags2"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{key}, seed {seed}")
        if n_rep >= 2:
            np.testing.assert_array_equal(a["overlap2"], b["overlap2"], err_msg=f"seed {seed}")


def test_soak_slab_pipeline_24_cubed(oracle):
    """k_sweep_slab (LDS-resident slabs, tagged exchange records, speculative re-reads): 50 seeds."""
    _soak(oracle, (24, 24, 24), 4, 2, 1, range(50), 2, pt_interval=1, warmup_ratio=0.0)


def test_soak_slab_pipeline_64_cubed(oracle):
    _soak(oracle, (64, 64, 64), 2, 1, 1, range(6), 2, pt_interval=1, warmup_ratio=0.0)


def test_soak_small_block_random_stream(oracle):
    """Small ensembles take short random-stream blocks (31-site rule): cursor in / cursor_out race class."""
    _soak(oracle, (8, 8), 3, 2, 1, range(50), 6, kind="glass", pt_interval=1, warmup_ratio=0.0)


def test_soak_fused_overlap_launch(oracle):
    """k_rng_overlap: the overlap of sweep s-1 measured inside the launch that draws sweep s."""
    _soak(oracle, (8, 8, 8), 6, 4, 3, range(50), 5, kind="glass", pt_interval=1, warmup_ratio=0.2)


def test_rccl_communicator_single_rank(oracle):
    """pa_comm_* (RCCL bound by dlopen, no torch): a one-rank communicator on this box's GPU runs the same code
    path the 8-GPU bench takes -- id, init, barrier, all-reduce, all-gather, gathered result blocks."""
    from peapods_amd import IsingSimulation
    from peapods_amd.distributed import Comm, aggregate_mean, all_gather_blocks

    comm = Comm(0, 1, 0)
    try:
        comm.barrier()
        assert comm.max_f64(2.5) == 2.5
        np.testing.assert_array_equal(comm.sum_u64([3, 4]), np.array([3, 4], dtype=np.uint64))
        x = np.arange(12, dtype=np.float64).reshape(3, 4)
        np.testing.assert_array_equal(comm.all_gather(x), x[None])
        shape, n_rep = (4, 4, 4), 2
        temps = np.linspace(1.0, 2.0, 3).astype(np.float32)
        J = bimodal(shape, 3, 8)
        sim = IsingSimulation(list(shape), J, temps, n_rep, None, 4)
        sim.sample(10, "metropolis", pt_interval=1)
        blocks = sim.result_blocks()
        np.testing.assert_array_equal(comm.gather_result_blocks(sim, 3), blocks)
        np.testing.assert_array_equal(all_gather_blocks(blocks, comm), blocks)
        cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 4)
        want = cpu.sample(10, "metropolis", pt_interval=1)
        np.testing.assert_array_equal(aggregate_mean(blocks)[:3], want["mags"])
    finally:
        comm.close()


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
def test_profile_step_advances_the_state_like_sample(oracle, schedule):
    """pa_sim_profile_step runs sample()'s own launch sequence (relayed measurement block included), once with an event in
    front of every launch and once without: the ensemble ends exactly where two sample() calls would leave it, and the
    results are those of the second."""
    from peapods_amd import IsingSimulation

    shape, n_rep = (8, 8, 8), 4
    temps = np.linspace(0.8, 1.4, 6).astype(np.float32)
    J = bimodal(shape, 3, 17)
    gpu = IsingSimulation(list(shape), J, temps, n_rep, None, 12, stream_groups=1)
    cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 12)
    kw = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.0)
    gpu.sample(7, "metropolis", **kw)
    cpu.sample(7, "metropolis", **kw)
    table, step_ms = gpu.profile_step(gpu.make_config(9, "metropolis", **kw), 9)
    assert step_ms > 0 and any(name.startswith("k_rng") for name in table) and any(name.startswith("k_sweep") for name in table)
    cpu.sample(9, "metropolis", **kw)
    want = cpu.sample(9, "metropolis", **kw)
    for r in range(3):
        np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))
        np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r))
        np.testing.assert_array_equal(gpu.get_energies(r), cpu.get_energies(r))
    blocks = gpu.result_blocks()  # and the statistics of those 9 sweeps are the reference's
    from peapods_amd.distributed import aggregate_mean

    np.testing.assert_array_equal(aggregate_mean(blocks)[: len(temps)], want["mags"])
    a = gpu.sample(5, "metropolis", **kw)
    b = cpu.sample(5, "metropolis", **kw)
    np.testing.assert_array_equal(a["overlap2"], b["overlap2"])


@pytest.mark.gpu
@pytest.mark.parametrize("config,n_sweeps,extra", [
    ("c2", 300, {}),
    ("c2", 150, {"pt_schedule": "full_ladder"}),
    ("c4", 60, {"overlap_cluster_build_mode": "jorg", "overlap_cluster_mode": "wolff"}),
    ("c1", 60, {}),
])
def test_full_size_runs_are_reproducible(config, n_sweeps, extra):
    """Race hunt at BASELINE sizes (tools/determinism_soak.py): two handles, one seed, the same protocol -- result
    blocks, tempering counters and final spins must agree bit for bit (the reference is deterministic under a seed,
    src/lib.rs:116-173)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "determinism_soak.py")
    spec = importlib.util.spec_from_file_location("determinism_soak", path)
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    soak.one(config, n_sweeps, **extra)


@pytest.mark.parametrize("schedule", ["single_random_edge", "full_ladder"])
def test_captured_graph_pairs_replay_the_same_trajectory(oracle, monkeypatch, schedule):
    """PEAPODS_GRAPH=1: pairs of steady-state sweeps go through one captured hipGraph per stream group (opt-in: it
    measured slower).  The replayed launches must leave exactly what launch-by-launch enqueueing leaves --
    across calls too (the second call replays the graphs of the first; the Gibbs call captures its own)."""
    from peapods_amd import IsingSimulation

    monkeypatch.setenv("PEAPODS_DEVELOPER", "1")
    monkeypatch.setenv("PEAPODS_GRAPH", "1")
    shape, n_rep = (8, 8, 8), 4
    temps = np.linspace(0.8, 1.4, 6).astype(np.float32)
    J = bimodal(shape, 3, 23)
    gpu = IsingSimulation(list(shape), J, temps, n_rep, None, 31)
    cpu = oracle.OracleSimulation(list(shape), J, temps, n_rep, None, 31)
    kw = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.25)
    for n_sweeps, mode in ((21, "metropolis"), (12, "metropolis"), (9, "gibbs"), (14, "metropolis")):
        a = gpu.sample(n_sweeps, mode, **kw)
        b = cpu.sample(n_sweeps, mode, **kw)
        for key in ("mags", "mags2", "energies", "energies2", "overlap", "overlap2", "overlap4", "link_overlap",
                    "ql_at_q_sum", "per_sample_overlap_histogram"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{mode} {n_sweeps} {key}")
        for r in range(3):
            np.testing.assert_array_equal(gpu.get_spins(r), cpu.get_spins(r))
            np.testing.assert_array_equal(gpu.get_system_ids(r), cpu.get_system_ids(r))
        pt_a, pt_b = a["per_disorder"]["parallel_tempering"], b["per_disorder"]["parallel_tempering"]
        for key in pt_a:
            np.testing.assert_array_equal(pt_a[key], pt_b[key], err_msg=key)


def test_step_profile_names_every_launch_of_a_step():
    """pa_sim_profile_step: sample()'s own launch sequence with an event in front of every launch (PA_LAUNCH), by kernel --
    cluster moves included, which pa_sim_profile_kernels declines.  The kernels' times add up to the step."""
    from peapods_amd import IsingSimulation

    shape = (8, 8, 8)
    rng = np.random.default_rng(2)
    couplings = (2 * rng.integers(0, 2, size=(4,) + shape + (3,)) - 1).astype(np.float32)
    temps = np.linspace(0.8, 1.6, 6).astype(np.float32)
    two = IsingSimulation(list(shape), couplings, temps, 4, None, 3)
    cfg = two.make_config(8, "metropolis", pt_interval=1, warmup_ratio=0.0)
    with pytest.raises(ValueError, match="one stream group"):
        two.profile_step(cfg, 8)
    one = IsingSimulation(list(shape), couplings, temps, 4, None, 3, stream_groups=1)
    table, step_ms = one.profile_step(cfg, 16)
    bases = {name.split("<")[0] for name in table}
    assert "k_rng_relay" in bases and "k_sweep_hc" in bases, table
    assert step_ms > 0 and abs(sum(ms for ms, _ in table.values()) - step_ms) < 0.02 * step_ms + 1e-4
    relay = [v for k, v in table.items() if k.startswith("k_rng_relay")]
    assert abs(sum(n for _, n in relay) - 1.0) < 0.15  # one random-stream launch per sweep
    # a step with the Joerg move: its launches are in the table
    cfg4 = one.make_config(8, "metropolis", pt_interval=1, warmup_ratio=0.0, overlap_cluster_update_interval=1,
                           overlap_cluster_build_mode="jorg")
    table4, step4 = one.profile_step(cfg4, 8)
    assert any(name.startswith("k_wolff_wave") for name in table4), table4
    assert step4 > step_ms
    # the trajectory does not depend on the stream groups: both handles continue alike
    two.sample(40, "metropolis", pt_interval=1)   # (profile_step ran 2 x 16 + 2 x 8 sweeps, in two protocols: replay them)
    fresh_one = IsingSimulation(list(shape), couplings, temps, 4, None, 3, stream_groups=1)
    fresh_two = IsingSimulation(list(shape), couplings, temps, 4, None, 3)
    a = fresh_one.sample(12, "metropolis", pt_interval=1)
    b = fresh_two.sample(12, "metropolis", pt_interval=1)
    np.testing.assert_array_equal(a["energies"], b["energies"])
    np.testing.assert_array_equal(fresh_one.get_spins(3), fresh_two.get_spins(3))


def test_cluster_phase_stamps_cover_every_pair_and_do_not_disturb_the_run():
    """pa_debug_cluster_stamps (include/peapods_amd_debug.h): every workgroup of the union-find cluster kernel stamps its
    phase boundaries; the stamps of a workgroup rise, and a run that stamps leaves the state of one that does not."""
    import ctypes as C

    from peapods_amd import IsingSimulation
    from peapods_amd import _native as nat

    shape = (6, 6, 6)
    rng = np.random.default_rng(5)
    couplings = (2 * rng.integers(0, 2, size=(3,) + shape + (3,)) - 1).astype(np.float32)
    temps = np.linspace(0.8, 1.8, 5).astype(np.float32)
    kw = dict(pt_interval=1, overlap_cluster_update_interval=1, overlap_cluster_build_mode="jorg", overlap_cluster_mode="sw")
    stamped = IsingSimulation(list(shape), couplings, temps, 4, None, 9, stream_groups=1)
    plain = IsingSimulation(list(shape), couplings, temps, 4, None, 9, stream_groups=1)
    lib = nat.lib()
    lib.pa_debug_cluster_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    n_blocks = lib.pa_debug_cluster_stamps(stamped._handle, 1, None, 0)
    assert n_blocks == 3 * 5 * 4  # >= realisations x temperatures x pairs; the buffer also covers one block per system
    a = stamped.sample(6, "metropolis", **kw)
    b = plain.sample(6, "metropolis", **kw)
    out = np.zeros((n_blocks, 16), dtype=np.uint64)
    assert lib.pa_debug_cluster_stamps(stamped._handle, 0, nat.ptr(out), n_blocks) == n_blocks
    rows = out[: 3 * 5 * 2].astype(np.int64)  # the pairs of the last launch
    order = [0, 1, 2, 3, 12, 4, 5, 6, 7, 8, 9, 11]
    assert (rows[:, order] > 0).all()
    assert (np.diff(rows[:, order], axis=1) >= 0).all()
    np.testing.assert_array_equal(a["energies"], b["energies"])
    np.testing.assert_array_equal(stamped.get_spins(2), plain.get_spins(2))
    assert lib.pa_debug_cluster_stamps(stamped._handle, 0, nat.ptr(out), n_blocks) == 0  # stopped: nothing to read
