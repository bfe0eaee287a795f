"""Disorder sharding over ranks (one process per GPU) and the final observable exchange.

Disorder realisations are independent (reference simulation/mod.rs:887-903), so rank g of G owns
a contiguous block of them and runs with no data-path collective.  Realisation seeds depend only
on the global realisation index (src/lib.rs:30-32), hence per-realisation results do not depend
on G.  The only exchange is one all-gather of the fixed-size per-realisation observable blocks
(RCCL when the process group backend is "nccl"; gloo in the CPU tests), after which the
reference's disorder aggregation runs in realisation order.
"""

import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous realisation range [start, stop) of `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    start = n_total * rank // world
    stop = n_total * (rank + 1) // world
    return start, stop


def all_gather_blocks(blocks, dist=None, device=None):
    """All-gathers [n_local][block_len] f64 blocks into [n_total][block_len] in rank order.
    Ranks may own different numbers of realisations."""
    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return blocks
    import torch

    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[dist.get_rank()] = blocks.shape[0]
    dist.all_reduce(counts)
    n_max = int(counts.max().item())
    padded = torch.zeros((n_max, blocks.shape[1]), dtype=torch.float64, device=dev)
    padded[: blocks.shape[0]] = torch.from_numpy(blocks).to(dev)
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    rows = [gathered[r][: int(counts[r].item())].cpu().numpy() for r in range(world)]
    return np.concatenate(rows, axis=0)


def aggregate_mean(blocks):
    """Mean over realisations, summed sequentially in realisation order then divided once
    (statistics/results.rs:165-196,262-282), through the C ABI helper."""
    from . import _native as nat

    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    out = np.empty(blocks.shape[1], dtype=np.float64)
    nat.lib().pa_aggregate_mean(nat.ptr(blocks), blocks.shape[0], blocks.shape[1], nat.ptr(out))
    return out
