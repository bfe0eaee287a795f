"""Disorder sharding over ranks (one process per GPU) and the final observable exchange.

Disorder realisations are independent (reference simulation/mod.rs:887-903), so rank g of G owns
a contiguous block of them and runs with no data-path collective.  Realisation seeds depend only
on the global realisation index (src/lib.rs:30-32), hence per-realisation results do not depend
on G.  The only exchange is one all-gather of the fixed-size per-realisation observable blocks,
after which the reference's disorder aggregation (statistics/results.rs:165-282) runs in
realisation order.

On GPUs the exchange is `Comm`: RCCL through the library's own C ABI (pa_comm_*, csrc/comm.hip) --
no torch.  The 128-byte RCCL id travels from rank 0 to the others through a file named after the launch (one
node) or a plain TCP socket on MASTER_ADDR:MASTER_PORT (`exchange_id`).  The CPU rehearsal in tests/ passes a gloo
`torch.distributed` module to `all_gather_blocks` instead; that branch is test plumbing only.
"""

import ctypes as C
import os
import socket
import sys
import time

import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous realisation range [start, stop) of `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    start = n_total * rank // world
    stop = n_total * (rank + 1) // world
    return start, stop


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=120.0, channel=None):
    """Rank 0 creates the communicator id (`make_id()` -> bytes) and hands it to the other world - 1 ranks.

    channel "file" (default when MASTER_ADDR is this host): rank 0 writes the id to a file named after
    MASTER_PORT and the launcher's pid -- every rank of one launch has the same parent, torch.distributed.run's
    agent or bench.py's own launcher -- and the others poll for it.  Nothing is bound: under torch.distributed.run
    MASTER_PORT is already held by the agent's rendezvous store.
    channel "tcp" (default otherwise): rank 0 serves the id on MASTER_ADDR:MASTER_PORT, the others connect
    (retrying until rank 0 listens)."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("MASTER_PORT", "29500"))
    if world == 1:
        return make_id()
    if channel is None:
        channel = "file" if addr in ("127.0.0.1", "localhost", "::1") else "tcp"
    if channel == "file":
        return _exchange_id_file(rank, make_id, port, timeout)
    if rank == 0:
        payload = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(payload)
        return payload
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                chunks = b""
                while True:
                    part = conn.recv(4096)
                    if not part:
                        break
                    chunks += part
            if chunks:
                return chunks
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id from {addr}:{port}")
        time.sleep(0.05)


_ID_FILES = []


def forget_id_files():
    """Rank 0: remove the id files it wrote (call after the communicator exists: every rank has read them)."""
    while _ID_FILES:
        try:
            os.remove(_ID_FILES.pop())
        except OSError:
            pass


def _exchange_id_file(rank, make_id, port, timeout):
    import tempfile

    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    path = os.path.join(tempfile.gettempdir(), f"peapods_comm_{os.getppid()}_{port}_{restart}.id")
    if rank == 0:
        payload = make_id()
        scratch = f"{path}.{os.getpid()}"
        with open(scratch, "wb") as fh:
            fh.write(payload)
        os.replace(scratch, path)  # readers see the whole id or nothing
        _ID_FILES.append(path)     # removed by forget_id_files() once every rank holds the id
        return payload
    deadline = time.monotonic() + timeout
    while True:
        try:
            with open(path, "rb") as fh:
                payload = fh.read()
            if payload:
                return payload
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id in {path}")
        time.sleep(0.02)


class Comm:
    """One RCCL communicator rank (pa_comm, include/peapods_amd.h): all-gather of host arrays staged
    through device memory, max / sum all-reduce, barrier."""

    def __init__(self, rank, world, device, addr=None, port=None):
        from . import _native as nat

        self._nat = nat
        self._lib = nat.lib()
        self._handle = None
        self.rank, self.world = int(rank), int(world)

        def make_id():
            buf = C.create_string_buffer(nat.PA_COMM_ID_BYTES)
            err = C.create_string_buffer(512)
            if self._lib.pa_comm_unique_id(buf, err, len(err)) != nat.PA_OK:
                raise RuntimeError(err.value.decode())
            return buf.raw

        # RCCL prints a version banner on stdout when it initialises; programs that print machine-readable
        # results there (bench.py) must not carry it, so file descriptor 1 points at stderr meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = exchange_id(self.rank, self.world, make_id, addr, port)
            if len(uid) != nat.PA_COMM_ID_BYTES:
                raise RuntimeError("communicator id of the wrong size")
            handle = C.c_void_p()
            err = C.create_string_buffer(512)
            code = self._lib.pa_comm_init_rank(C.create_string_buffer(uid, len(uid)), self.world, self.rank,
                                               int(device), C.byref(handle), err, len(err))
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        if code != nat.PA_OK:
            raise RuntimeError(err.value.decode())
        self._handle = handle
        forget_id_files()  # ncclCommInitRank returned: every rank has the id

    def close(self):
        if self._handle:
            self._lib.pa_comm_destroy(self._handle)
            self._handle = None

    __del__ = close

    def _check(self, code):
        if code != self._nat.PA_OK:
            raise RuntimeError(self._lib.pa_comm_last_error(self._handle).decode())

    def barrier(self):
        self._check(self._lib.pa_comm_barrier(self._handle))

    def max_f64(self, value):
        buf = np.array([value], dtype=np.float64)
        self._check(self._lib.pa_comm_allreduce_max_f64(self._handle, self._nat.ptr(buf), 1))
        return float(buf[0])

    def sum_u64(self, values):
        buf = np.ascontiguousarray(values, dtype=np.uint64).copy()
        self._check(self._lib.pa_comm_allreduce_sum_u64(self._handle, self._nat.ptr(buf), buf.size))
        return buf

    def all_gather(self, array):
        """[world, *array.shape]: every rank's array (same shape and dtype everywhere)."""
        array = np.ascontiguousarray(array)
        out = np.empty((self.world,) + array.shape, dtype=array.dtype)
        self._check(self._lib.pa_comm_allgather(self._handle, self._nat.ptr(array), self._nat.ptr(out), array.nbytes))
        return out

    def gather_result_blocks(self, sim, n_total):
        """Every rank's per-realisation result blocks in global realisation order: [n_total][block]."""
        length = int(self._lib.pa_sim_result_block_len(sim._handle))
        out = np.empty((int(n_total), length), dtype=np.float64)
        got = C.c_uint64()
        self._check(self._lib.pa_comm_gather_result_blocks(self._handle, sim._handle, self._nat.ptr(out), int(n_total),
                                                           C.byref(got)))
        if got.value != n_total:
            raise RuntimeError(f"gathered {got.value} result blocks, expected {n_total}")
        return out


def all_gather_blocks(blocks, comm=None):
    """All-gathers [n_local][block_len] f64 blocks into [n_total][block_len] in rank order.  Ranks
    may own different numbers of realisations.  `comm` is a `Comm` (RCCL), None (one process), or --
    in the CPU tests only -- an initialised `torch.distributed` module (gloo)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    if comm is None:
        return blocks
    if isinstance(comm, Comm):
        counts = comm.all_gather(np.array([blocks.shape[0]], dtype=np.uint64)).reshape(-1)
        n_max = int(counts.max())
        padded = np.zeros((n_max, blocks.shape[1]), dtype=np.float64)
        padded[: blocks.shape[0]] = blocks
        gathered = comm.all_gather(padded)
        return np.concatenate([gathered[r, : int(counts[r])] for r in range(comm.world)], axis=0)
    return _all_gather_blocks_gloo(blocks, comm)


def _all_gather_blocks_gloo(blocks, dist):
    """CPU rehearsal of the same exchange with torch.distributed (gloo) -- tests only."""
    if not dist.is_initialized():
        return blocks
    import torch

    world = dist.get_world_size()
    counts = torch.zeros(world, dtype=torch.int64)
    counts[dist.get_rank()] = blocks.shape[0]
    dist.all_reduce(counts)
    n_max = int(counts.max().item())
    padded = torch.zeros((n_max, blocks.shape[1]), dtype=torch.float64)
    padded[: blocks.shape[0]] = torch.from_numpy(blocks)
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    rows = [gathered[r][: int(counts[r].item())].numpy() for r in range(world)]
    return np.concatenate(rows, axis=0)


def aggregate_mean(blocks):
    """Mean over realisations, summed sequentially in realisation order then divided once
    (statistics/results.rs:165-196,262-282), through the C ABI helper."""
    from . import _native as nat

    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    out = np.empty(blocks.shape[1], dtype=np.float64)
    nat.lib().pa_aggregate_mean(nat.ptr(blocks), blocks.shape[0], blocks.shape[1], nat.ptr(out))
    return out
