"""Multi-GPU placement of a batch of independent Newton systems.

The path shards only across independent problems: contiguous shards of ceil(B/G) problems, one process
and one device per shard, no collective on the data path.  torch.distributed (RCCL on the GPU box, gloo
in the CPU tests) is used for the barrier that brackets the timed region, the max-over-ranks time and
the gather of per-shard status vectors.
"""
import os


def shard_range(total, world, rank):
    """[start, stop) of the problems owned by `rank` (contiguous, sizes differ by at most ceil-floor)."""
    per = -(-total // world)
    start = min(rank * per, total)
    return start, min(start + per, total)


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend):
    """process group for the barrier / reductions; MASTER_ADDR defaults to 127.0.0.1"""
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def max_over_ranks(value, dist, device="cpu"):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts, dist, device="cpu"):
    """sum over ranks of a small integer vector (e.g. [problems, successes, factorisations])"""
    import torch
    t = torch.tensor(list(local_counts), dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
