"""Multi-GPU plumbing: independent Systems shard across ranks with no data-path collective
(SURVEY.md §8e). The only communication is the end-of-run reduction of the throughput counters and
the max-over-ranks time, done with torch.distributed (backend "nccl" == RCCL over xGMI on the GPU
box, "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def rank_seed(base_seed: int, rank: int, systems_per_rank: int) -> int:
    """First LCG seed of a rank's shard: shards are disjoint slices of one global seed sequence."""
    return int(base_seed) + int(rank) * int(systems_per_rank)


def reduce_throughput(dist, elapsed_s: float, counters: Sequence[int], device=None) -> Tuple[float, list]:
    """MAX of the elapsed time and SUM of the integer counters over all ranks (identity without a
    process group)."""
    if dist is None or not dist.is_initialized():
        return float(elapsed_s), [int(c) for c in counters]
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(x) for x in counters], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(x) for x in c.tolist()]


def gather_times(dist, elapsed_s: float, device=None) -> List[float]:
    """Every rank's elapsed time, in rank order, on every rank (a one-element list without a process group): an N-GPU run is
    as slow as its slowest rank, and the bench line says which one that was."""
    if dist is None or not dist.is_initialized():
        return [float(elapsed_s)]
    import torch

    mine = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    everyone = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(everyone, mine)
    return [float(x.item()) for x in everyone]
