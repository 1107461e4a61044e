"""Multi-GPU plumbing: independent Systems shard across ranks with no data-path collective
(SURVEY.md §8e). The only communication is the end-of-run reduction of the throughput counters and
the max-over-ranks time, done with torch.distributed (backend "nccl" == RCCL over xGMI on the GPU
box, "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple


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


def throughput(total_counters: Dict[str, int], steps: int, elapsed_s: float) -> Dict[str, float]:
    """Whole-job rates from the summed per-step counters."""
    return {k: v * steps / elapsed_s for k, v in total_counters.items()}
