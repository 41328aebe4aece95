"""The N>1 path of bench.py on CPU: two gloo ranks, one independent stream each (no data-path
collective).  Each rank runs its stream through the CPU oracle (the GPU is not available here; on
the GPU box the same sharding drives the HIP tracker) and the test checks that
  * rank r gets stream r (seed base+r) and the two streams differ,
  * a rank's result is identical to running that stream alone in a single process,
  * the control plane (barrier, max/sum over ranks, aggregate steps/s) works."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(seed, frames=6):
    from oracle import bts_oracle as bo
    synth = importlib.import_module("bot-sort-onnx-tensorrt_amd.synth")
    st = synth.SyntheticStream(48, 24, 64, seed=seed, churn=0.05)
    trk = bo.TrackerOracle()
    acc = []
    for _ in range(frames):
        fr = st.next()
        trk.update_arrays(fr.boxes, fr.scores.astype(np.float64), fr.feats)
        acc.append([t.track_id for t in trk.tracked] + [-1] + [t.track_id for t in trk.lost])
        acc.append([int(v) for v in trk.trace["m1"].ravel()])
    return repr(acc)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ms = importlib.import_module("bot-sort-onnx-tensorrt_amd.multistream")
    mine = ms.streams_of_rank(world, rank, world)
    assert mine == [rank]
    ms.barrier(dist)
    d = _digest(ms.stream_seed(1234, mine[0]))
    elapsed = 1.0 + rank                      # pretend rank 1 is the slower stream
    tmax = ms.max_over_ranks(elapsed, dist)
    total = ms.sum_over_ranks(10.0, dist)
    ms.barrier(dist)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%r\n%r\n%s" % (tmax, total, d))
    dist.destroy_process_group()


def test_two_streams_two_ranks(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, REPO)
    ms = importlib.import_module("bot-sort-onnx-tensorrt_amd.multistream")
    outs = []
    for r in range(2):
        tmax, total, d = open(os.path.join(str(tmp_path), "r%d.txt" % r)).read().split("\n", 2)
        assert float(tmax) == 2.0 and float(total) == 20.0
        assert d == _digest(ms.stream_seed(1234, r))          # same as a single-process run
        outs.append(d)
    assert outs[0] != outs[1]
    assert ms.aggregate_throughput(100, 2.0, 2) == 100.0
    assert ms.streams_of_rank(8, 3, 4) == [3, 7]
