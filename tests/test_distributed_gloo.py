"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): disorder shards are independent,
the observable blocks are all-gathered, and the aggregate equals the single-process result bit
for bit.  The per-shard work is done by the oracle here (there is no GPU in this test); what is
under test is the product's sharding, gather and aggregation code."""

import os
import socket
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ensemble():
    from peapods_amd.spin_models import generate_couplings, seed_material

    shape, n_dis = (4, 4, 4), 5
    coupling_seq, dyn = seed_material(42)
    couplings = np.stack(generate_couplings("bimodal", shape, 3, n_dis, coupling_seq))
    temps = np.linspace(0.8, 1.4, 4).astype(np.float32)
    return shape, n_dis, couplings, temps, dyn


def _blocks(first, stop):
    from oracle import bindings as ob

    shape, n_dis, couplings, temps, dyn = _ensemble()
    sim = ob.OracleSimulation(list(shape), couplings[first:stop], temps, 2, None, dyn,
                              first_realization=first)
    sim.run(sim.make_config(12, "metropolis", pt_interval=1, warmup_ratio=0.25))
    return sim.result_blocks()


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from peapods_amd.distributed import aggregate_mean, all_gather_blocks, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_dis = _ensemble()[1]
    first, stop = shard_range(n_dis, rank, world)
    gathered = all_gather_blocks(_blocks(first, stop), dist)
    if rank == 0:
        np.save(out_path, np.vstack([aggregate_mean(gathered)[None, :], gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    import torch.multiprocessing as mp

    from peapods_amd.distributed import aggregate_mean

    out_path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = np.load(out_path)
    n_dis = _ensemble()[1]
    single = _blocks(0, n_dis)
    np.testing.assert_array_equal(got[1:], single)  # shards 3 + 2 realisations, seeds by global index
    np.testing.assert_array_equal(got[0], aggregate_mean(single))


def _id_worker(rank, world, port, out_dir, channel):
    sys.path.insert(0, ROOT)
    from peapods_amd.distributed import exchange_id

    got = exchange_id(rank, world, lambda: bytes(range(128)), "127.0.0.1", port, timeout=30.0, channel=channel)
    with open(os.path.join(out_dir, f"id{rank}.bin"), "wb") as fh:
        fh.write(got)
    if rank == 0 and channel == "file":  # (Comm removes the file once the communicator exists; here: once all have it)
        from peapods_amd.distributed import forget_id_files

        deadline = time.monotonic() + 30
        while time.monotonic() < deadline and not all(os.path.exists(os.path.join(out_dir, f"id{r}.bin")) for r in range(world)):
            time.sleep(0.05)
        forget_id_files()


@pytest.mark.parametrize("channel", ["file", "tcp"])
def test_communicator_id_travels_from_rank_0(tmp_path, channel):
    """The torch-free rendezvous of the RCCL path: rank 0 hands the 128-byte id to the others (who may ask before
    rank 0 is ready) -- through a file named after the launch (the default on one node: under torch.distributed.run
    MASTER_PORT belongs to the agent's store and cannot be bound) or over TCP."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    port = _free_port()
    if channel == "file":  # hold the port, as torch.distributed.run's agent does
        holder = socket.socket()
        holder.bind(("127.0.0.1", port))
        holder.listen(1)
    procs = [ctx.Process(target=_id_worker, args=(r, 3, port, str(tmp_path), channel)) for r in (2, 1, 0)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(3):
        assert open(tmp_path / f"id{r}.bin", "rb").read() == bytes(range(128))


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts N workers itself and exits non-zero when fewer devices exist
    (it must never report n_gpus smaller than asked)."""
    import subprocess

    env = dict(os.environ)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert run.returncode != 0
    assert "GPU(s) are visible" in run.stderr
    # a worker environment whose world size contradicts --gpus is refused too
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=dict(env, RANK="0", WORLD_SIZE="1"), timeout=120)
    assert run.returncode == 2 and "WORLD_SIZE=1" in run.stderr
