"""Disorder sharding over ranks (one process per GPU), the final observable exchange, and the multi-GPU
`sample()`.

Disorder realisations are independent (reference simulation/mod.rs:887-903), so rank g of G owns
a contiguous block of them and runs with no data-path collective.  Realisation seeds depend only
on the global realisation index (src/lib.rs:30-32), hence per-realisation results do not depend
on G.  The only exchange comes after the run: every per-realisation result array -- moment blocks, P(q)
histograms and conditional sums, tempering counters, autocorrelation times, equilibration checkpoints,
cluster statistics, observations, snapshots -- is gathered in realisation order (`gather_parts`), after
which the reference's disorder aggregation (statistics/results.rs:165-282, overlap.rs:106-152,
src/lib.rs:344-487) runs exactly as it does for one handle (`_core.assemble_parts`): `sample()` below
returns the reference's full dictionary, bit-identical to a single handle holding every realisation.

On GPUs the exchange is `Comm`: RCCL through the library's own C ABI (pa_comm_*, csrc/comm.hip) --
no torch.  The 128-byte RCCL id travels from rank 0 to the others through a file named after the launch (one
node) or a plain TCP socket (`exchange_id`).  `LoopbackGroup` is the same exchange between the threads of
one process (several handles on one GPU); `GlooExchange` wraps an initialised `torch.distributed` (the CPU
rehearsal in tests/).
"""

import ctypes as C
import os
import socket
import sys
import threading
import time
import zlib

import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous realisation range [start, stop) of `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    start = n_total * rank // world
    stop = n_total * (rank + 1) // world
    return start, stop


def _single_node_launch(addr):
    """True when every rank of this launch runs on this host: torch.distributed.run exports
    LOCAL_WORLD_SIZE == WORLD_SIZE for one node, and bench.py's own launcher uses the loopback address;
    otherwise: does MASTER_ADDR name an address of this machine?"""
    local, world = os.environ.get("LOCAL_WORLD_SIZE"), os.environ.get("WORLD_SIZE")
    if local is not None and world is not None:
        return local == world
    if addr in ("127.0.0.1", "localhost", "::1"):
        return True
    try:
        target = {info[4][0] for info in socket.getaddrinfo(addr, None)}
        mine = {info[4][0] for info in socket.getaddrinfo(socket.gethostname(), None)} | {"127.0.0.1", "::1"}
        return bool(target & mine)
    except OSError:
        return False


def id_port(master_port):
    """Port rank 0 serves the communicator id on (multi-node launches): PEAPODS_ID_PORT, or MASTER_PORT + 1.
    Never MASTER_PORT itself -- under torch.distributed.run that one is held by the agent's TCPStore."""
    env = os.environ.get("PEAPODS_ID_PORT")
    return int(env) if env else int(master_port) + 1


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=120.0, channel=None, sequence=0):
    """Rank 0 creates the communicator id (`make_id()` -> bytes) and hands it to the other world - 1 ranks.

    channel "file" (default whenever the launch is on one node): rank 0 writes the id to a private file named
    after MASTER_PORT, the launcher's pid and `sequence` (the how-many-th communicator of this process) -- every
    rank of one launch has the same parent, torch.distributed.run's agent or bench.py's own launcher -- and the
    others poll for it.  Nothing is bound.
    channel "tcp" (several nodes): rank 0 serves the id on MASTER_ADDR:`id_port(MASTER_PORT)` (+ sequence), the
    others connect, retrying until rank 0 listens."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("MASTER_PORT", "29500"))
    if world == 1:
        return make_id()
    if channel is None:
        channel = "file" if _single_node_launch(addr) else "tcp"
    if channel == "file":
        return _exchange_id_file(rank, make_id, port, timeout, sequence)
    serve_port = id_port(port) + int(sequence)
    if rank == 0:
        payload = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            deadline = time.monotonic() + timeout
            while True:  # a previous communicator's socket may still linger on the port
                try:
                    srv.bind((addr, serve_port))
                    break
                except OSError:
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.1)
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
            with socket.create_connection((addr, serve_port), timeout=5.0) as conn:
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
            raise TimeoutError(f"rank {rank}: no communicator id from {addr}:{serve_port}")
        time.sleep(0.05)


_ID_FILES = []


def forget_id_files():
    """Rank 0: remove the id files it wrote (call after the communicator exists: every rank has read them)."""
    while _ID_FILES:
        try:
            os.remove(_ID_FILES.pop())
        except OSError:
            pass


def _id_directory():
    """A directory only this user can write (mode 0700), so nobody else can plant an id file."""
    import tempfile

    path = os.path.join(tempfile.gettempdir(), f"peapods_comm_{os.getuid()}")
    os.makedirs(path, mode=0o700, exist_ok=True)
    info = os.stat(path)
    if info.st_uid != os.getuid() or (info.st_mode & 0o077):
        raise RuntimeError(f"{path} is not a private directory of this user")
    return path


def _exchange_id_file(rank, make_id, port, timeout, sequence=0):
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    path = os.path.join(_id_directory(), f"{os.getppid()}_{port}_{restart}_{int(sequence)}.id")
    if rank == 0:
        payload = make_id()
        scratch = f"{path}.{os.getpid()}"
        fd = os.open(scratch, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        try:
            with os.fdopen(fd, "wb") as fh:
                fh.write(payload)
            os.replace(scratch, path)  # readers see the whole id or nothing
        except BaseException:
            try:
                os.remove(scratch)
            except OSError:
                pass
            raise
        _ID_FILES.append(path)     # removed by forget_id_files() once every rank holds the id (or on failure)
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

    _created = 0  # communicators this process has made: part of the id's file name / port, so a second Comm of
                  # one launch never reads the first one's id

    def __init__(self, rank, world, device, addr=None, port=None):
        from . import _native as nat

        self._nat = nat
        self._lib = nat.lib()
        self._handle = None
        self.rank, self.world = int(rank), int(world)
        sequence = Comm._created
        Comm._created += 1

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
            uid = exchange_id(self.rank, self.world, make_id, addr, port, sequence=sequence)
            if len(uid) != nat.PA_COMM_ID_BYTES:
                raise RuntimeError("communicator id of the wrong size")
            handle = C.c_void_p()
            err = C.create_string_buffer(512)
            code = self._lib.pa_comm_init_rank(C.create_string_buffer(uid, len(uid)), self.world, self.rank,
                                               int(device), C.byref(handle), err, len(err))
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            forget_id_files()  # ncclCommInitRank returned (every rank has the id) or failed: the file goes either way
        if code != nat.PA_OK:
            raise RuntimeError(err.value.decode())
        self._handle = handle

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

    def gather_rows(self, array, root=None):
        """Concatenation along axis 0 of every rank's `array` in rank order (ranks may hold different numbers of
        rows, also none; trailing shape and dtype agree) -- on every rank, or on `root` only (the others get None):
        pa_comm_gatherv = grouped ncclBroadcast / ncclSend + ncclRecv of exactly the bytes each rank has."""
        array = np.ascontiguousarray(array)
        row_bytes = int(np.prod(array.shape[1:], dtype=np.int64)) * array.dtype.itemsize
        counts = self.all_gather(np.array([array.shape[0]], dtype=np.uint64)).reshape(-1)
        total = int(counts.sum())
        keeps = root is None or int(root) == self.rank
        out = np.empty((total if keeps else 0,) + array.shape[1:], dtype=array.dtype)
        sizes = np.zeros(self.world, dtype=np.uint64)
        self._check(self._lib.pa_comm_gatherv(self._handle, self._nat.ptr(array), array.shape[0] * row_bytes,
                                              self._nat.ptr(out), out.nbytes, -1 if root is None else int(root),
                                              self._nat.ptr(sizes)))
        if row_bytes and not np.array_equal(sizes, counts * np.uint64(row_bytes)):
            raise RuntimeError("ranks disagree about the row size of a gathered array")
        return out if keeps else None

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


# ------------------------------------------------------------------------------------------------
# multi-GPU sample(): shards run independently, every per-realisation result is gathered, one aggregation
# ------------------------------------------------------------------------------------------------
class LoopbackGroup:
    """`world` exchange endpoints inside one process: rank r's thread calls the collective methods of
    `endpoint(r)` in the same order as the others (several handles on one GPU, or the tests' stand-in for a
    node of GPUs).  Same interface as `Comm`: rank, world, gather_rows, barrier."""

    def __init__(self, world):
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world

    def endpoint(self, rank):
        return _LoopbackEndpoint(self, int(rank))


class _LoopbackEndpoint:
    def __init__(self, group, rank):
        if not 0 <= rank < group.world:
            raise ValueError("rank out of range")
        self._group, self.rank, self.world = group, rank, group.world

    def barrier(self):
        self._group._barrier.wait()

    def gather_rows(self, array, root=None):
        group = self._group
        group._slots[self.rank] = np.ascontiguousarray(array)
        group._barrier.wait()
        out = np.concatenate(group._slots, axis=0) if root is None or int(root) == self.rank else None
        group._barrier.wait()  # nobody posts the next array before everyone has read this one
        return out


class GlooExchange:
    """The same interface over an initialised `torch.distributed` process group (gloo): the CPU rehearsal."""

    def __init__(self, dist):
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def gather_rows(self, array, root=None):
        import torch

        array = np.ascontiguousarray(array)
        counts = torch.zeros(self.world, dtype=torch.int64)
        counts[self.rank] = array.shape[0]
        self._dist.all_reduce(counts)
        row_bytes = int(np.prod(array.shape[1:], dtype=np.int64)) * array.dtype.itemsize
        n_max = int(counts.max().item())
        padded = torch.zeros(max(n_max * row_bytes, 1), dtype=torch.uint8)
        raw = np.frombuffer(array.tobytes(), dtype=np.uint8)
        padded[: raw.size] = torch.from_numpy(raw.copy())
        gathered = [torch.empty_like(padded) for _ in range(self.world)]
        self._dist.all_gather(gathered, padded)
        if root is not None and int(root) != self.rank:
            return None
        rows = [np.frombuffer(gathered[r].numpy().tobytes()[: int(counts[r]) * row_bytes], dtype=array.dtype)
                .reshape((int(counts[r]),) + array.shape[1:]) for r in range(self.world)]
        return np.concatenate(rows, axis=0)


def gather_parts(parts, comm, root=None):
    """Every array of `parts` (first axis = realisation, _core.IsingSimulation.collect_parts) concatenated over the
    ranks in rank order = global realisation order for contiguous shards.  Every rank must hold the same set of
    arrays (same sample configuration); a digest of names, dtypes and row shapes is compared first, so a mismatch
    raises on every rank instead of hanging in a collective.  With `root` only that rank receives data."""
    keys = list(parts)  # (collect_parts files them in one order on every rank; the result dictionary keeps it)
    described = "|".join(f"{k}:{np.asarray(parts[k]).dtype.str}:{np.asarray(parts[k]).shape[1:]}" for k in keys)
    digest = np.array([[zlib.crc32(described.encode())]], dtype=np.uint64)
    if not (comm.gather_rows(digest) == digest[0, 0]).all():
        raise RuntimeError("ranks hold different result arrays (different sample() arguments?): " + described)
    gathered = {k: comm.gather_rows(parts[k], root) for k in keys}
    return gathered if root is None or int(root) == comm.rank else None


def sample(sim, comm, n_sweeps, sweep_mode, root=None, **kwargs):
    """`IsingSimulation.sample` for a disorder ensemble sharded over the ranks of `comm` (a `Comm`, or any object
    with rank / world / gather_rows): `sim` is this rank's handle, created with its shard of the couplings and
    `first_realization = shard_range(n_disorder, rank, world)[0]`.  Every rank runs its realisations with no
    data-path collective; afterwards the per-realisation results are gathered in realisation order and aggregated
    once (statistics/results.rs:165-282, overlap.rs:106-152, src/lib.rs:344-487).  Returns the reference's result
    dictionary over ALL realisations -- bit-identical to one handle holding them all -- on every rank, or on
    `root` only (None elsewhere: the histogram payload is 2.4 MB per realisation at 16^3 x 24 temperatures)."""
    from . import _core

    # A rank that fails before the first collective (invalid arguments, PA_ERR_UNSUPPORTED, a device error, Ctrl-C) must
    # not leave the others waiting in it: every rank reports ok / failed first, and all of them raise when one failed.
    failure, parts, meta = None, None, None
    try:
        cfg = sim.make_config(n_sweeps, sweep_mode, **kwargs)
        sim.run(cfg)
        parts, meta = sim.collect_parts(cfg, kwargs.get("pt_interval") is not None)
    except BaseException as exc:  # (re-raised below, after the exchange)
        failure = exc
    status = comm.gather_rows(np.array([[0 if failure is None else 1]], dtype=np.uint64))
    if failure is not None:
        raise failure
    if status.any():
        failed = [int(r) for r in np.nonzero(status[:, 0])[0]]
        raise RuntimeError(f"sample() failed on rank(s) {failed}; nothing was gathered")
    gathered = gather_parts(parts, comm, root)
    if gathered is None:
        return None
    return _core.assemble_parts(sim._lib, gathered, meta)
