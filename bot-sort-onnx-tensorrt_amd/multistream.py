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


def run_concurrent(trackers, frame_lists, hip_streams=None, before_start=None) -> float:
    """Several independent video streams on ONE GPU: tracker k consumes frame_lists[k] on hip_streams[k]
    (torch.cuda.Stream), each from its own host thread (the C ABI releases the GIL while it enqueues; a handle
    is used by one thread only).  The launch-bound steps of different streams overlap on the device.
    Returns the wall time from the common start to the moment every stream has drained.

    hip_streams=None (recommended): fresh streams, first used by the worker threads.  The runtime binds a stream to
    one of its few hardware queues at first use; streams that were already used one after the other during set-up
    were observed to end up sharing queues (four streams then run like two: 24.5k instead of 35-44k steps/s at
    512 x 256), so set-up work belongs on the default stream."""
    import threading
    import time
    import torch
    n = len(trackers)
    if hip_streams is None:
        hip_streams = [torch.cuda.Stream() for _ in range(n)]
    assert n == len(frame_lists) == len(hip_streams) and n >= 1
    gate = threading.Barrier(n + 1)
    errors = []

    def worker(k):
        try:
            with torch.cuda.stream(hip_streams[k]):
                gate.wait()
                for fr in frame_lists[k]:
                    trackers[k].step(*fr)
        except Exception as e:       # surfaced by the caller
            errors.append(e)
            try:
                gate.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(n)]
    for t in threads:
        t.start()
    if before_start is not None:
        before_start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gate.wait()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if errors:
        raise errors[0]
    return t1 - t0
