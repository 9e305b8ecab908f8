"""world_size-2 gloo test of the N>1 bookkeeping used by bench.py (runs on CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from canvas_amd import parallel, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank "processes" its own sample: rank r needs (r+1) seconds for 100*(r+1) bins
    secs, units, rate = parallel.aggregate_throughput(1.0 + rank, 100.0 * (rank + 1))
    counts, recs = parallel.gather_boundary_records([rank * 10 + k for k in range(rank + 2)], 8)
    shards = parallel.shard_units(synth.GRCH38, world)
    q.put((rank, secs, units, rate, counts, recs, shards[rank], parallel.sample_seed(7, rank)))
    dist.destroy_process_group()


def test_two_rank_bookkeeping():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs: p.join(30)
    for rank, secs, units, rate, counts, recs, shard, seed in res:
        assert secs == 2.0 and units == 300.0 and rate == 150.0           # MAX over ranks of time, SUM of units
        assert counts == [2, 3] and recs == [[0, 1], [10, 11, 12]]
        assert seed == 7 + 1000 * rank
    s0, s1 = res[0][6], res[1][6]
    assert sorted(s0 + s1) == list(range(24)) and not set(s0) & set(s1)
    l0 = sum(synth.GRCH38[i] for i in s0); l1 = sum(synth.GRCH38[i] for i in s1)
    assert abs(l0 - l1) / (l0 + l1) < 0.02                                    # LPT balance by chromosome length


def test_single_process_paths():
    assert parallel.aggregate_throughput(2.0, 10.0) == (2.0, 10.0, 5.0)
    assert parallel.gather_boundary_records([4, 5], 4) == ([2], [[4, 5]])
    assert parallel.shard_units([5, 1, 1, 1, 1, 1], 2) == [[0], [1, 2, 3, 4, 5]]
