"""Multi-GPU = multi-stream: independent video streams are dealt round-robin to GPUs, one process
per GPU, and never exchange data (no collective on the data path).  The only
cross-rank traffic is control: a start/stop barrier and the max-over-ranks elapsed time.
These helpers are backend-agnostic (`gloo` on CPU in the tests, `gloo` next to the GPUs in
bench.py -- RCCL is deliberately unused)."""
from __future__ import annotations

from typing import List


def stream_seed(base_seed: int, stream_id: int) -> int:
    """Seed of the synthetic generator for video stream `stream_id`."""
    return int(base_seed) + int(stream_id)


def streams_of_rank(n_streams: int, rank: int, world: int) -> List[int]:
    """Stream s runs on rank s mod world."""
    return [s for s in range(int(n_streams)) if s % int(world) == int(rank)]


def barrier(dist=None):
    if dist is not None and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, dist=None) -> float:
    """max of a host float over all ranks (the timed region ends when the slowest stream does)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(value: float, dist=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def aggregate_throughput(steps_per_rank: int, elapsed_max_s: float, world: int) -> float:
    """Whole-job update steps per second: every rank ran `steps_per_rank` steps of its own stream."""
    return world * steps_per_rank / elapsed_max_s
