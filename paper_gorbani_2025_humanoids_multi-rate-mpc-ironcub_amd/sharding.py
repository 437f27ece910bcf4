"""Batch split of independent MPC instances across the GPUs of one node.

The path shards embarrassingly: instances never exchange data, so every rank owns a contiguous
slice of the global instance index range and rebuilds (or receives) only that slice.  The only
collectives are OFF the data path: a SUM/MAX all-reduce of a few counters for reporting and an
optional all-gather of the 24-double first-move blocks (164 B per instance).  With RCCL over xGMI
these are latency-bound messages; nothing here is bucketed or overlapped because nothing is on
the timed path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [first, first+count) slice of `total` instances owned by `rank`; slices differ by at
    most one instance and cover the range exactly (ragged totals allowed)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def reduce_counters(solved: int, iterations: int, max_err: float, elapsed: float, device=None):
    """All-reduce of the reporting counters: (sum solved, sum iterations, max error, max elapsed)."""
    import torch
    import torch.distributed as dist
    s = torch.tensor([float(solved), float(iterations)], dtype=torch.float64, device=device)
    m = torch.tensor([float(max_err), float(elapsed)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(s[0].item()), int(s[1].item()), float(m[0].item()), float(m[1].item())


def gather_first_moves(first_move_local, total: int):
    """All-gather of the per-rank first-move blocks into the global instance order (ragged slices)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return first_move_local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    width = first_move_local.shape[1]
    pad = max(counts)
    buf = torch.zeros((pad, width), dtype=first_move_local.dtype, device=first_move_local.device)
    buf[:counts[rank]] = first_move_local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def rank_inputs(cfg, synth, total: int, rank: int, world: int, workload: str = "hover", seed0: int = 1234) -> np.ndarray:
    first, count = shard_range(total, rank, world)
    return synth.make_batch(cfg, count, workload=workload, seed0=seed0, first_index=first)
