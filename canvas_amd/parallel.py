"""Multi-process plumbing of the hot path (one process per GPU).  The path shards by independent units (samples of a cohort /
chromosomes of a sample) with NO data-path collective; the only exchanges are the boundary all-gather (RCCL, in the library) and the
throughput bookkeeping below, which works on any torch.distributed backend (nccl on the GPU box, gloo in the CPU tests)."""
import numpy as np


def shard_units(weights, world):
    """Longest-processing-time assignment of units (e.g. chromosomes by length) to ranks: returns a list of index lists."""
    order = np.argsort(-np.asarray(weights, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += weights[i]
    return [sorted(o) for o in out]


def sample_seed(base_seed, rank):
    """cohort mode: rank r owns sample r"""
    return base_seed + 1000 * rank


def aggregate_throughput(seconds, units, device=None):
    """(max over ranks of the timed region, sum over ranks of the units processed) -> (seconds, units, units/s)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item()), float(u.item()) / float(t.item())


def gather_boundary_records(local_records, max_per_rank):
    """reference semantics of canvas_allgather_boundaries on any backend: every rank contributes [count, rec...] padded to
    1 + max_per_rank int32; returns (counts per rank, [records of rank r])"""
    import torch
    import torch.distributed as dist
    rec = torch.zeros(1 + max_per_rank, dtype=torch.int32)
    rec[0] = len(local_records)
    rec[1:1 + len(local_records)] = torch.as_tensor(local_records, dtype=torch.int32)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [len(local_records)], [list(local_records)]
    out = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    counts = [int(o[0]) for o in out]
    return counts, [o[1:1 + c].tolist() for o, c in zip(out, counts)]
